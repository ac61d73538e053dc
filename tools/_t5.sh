mkdir -p gpurun_out/t5
(timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 -x 2>&1 | tail -15) > gpurun_out/t5/tests.log 2>&1; tail -6 gpurun_out/t5/tests.log
b() { echo "== $*"; env "$@" python bench.py --steps 120 --warmup 20 --no-cpu --sweeps 60 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['poisson_jacobi_sweep']['avg_us'], d['poisson_jacobi_sweep']['reads_v_like_reference']['avg_us'])"; }
b FS_X=1
b FS_X=2
python tools/kbench.py --res 1600 --bc 2 --steps 10 --warm 20 --sweeps 100 2>&1 | grep -v "^#"
