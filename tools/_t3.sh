mkdir -p gpurun_out/t3
(timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40) > gpurun_out/t3/tests.log 2>&1; tail -30 gpurun_out/t3/tests.log
for g in 1 0; do echo "== FS_LIMIT_GATE=$g"; FS_LIMIT_GATE=$g python bench.py --steps 200 --warmup 40 --no-cpu --sweeps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done
