mkdir -p gpurun_out/t4
(timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -30) > gpurun_out/t4/tests.log 2>&1; tail -12 gpurun_out/t4/tests.log
b() { echo "== $*"; env "$@" python bench.py --steps 100 --warmup 20 --no-cpu --sweeps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
b FS_FUSE_TRANSPORT=0
b FS_K34_RT=2
b FS_K34_RT=4
b FS_K34_RT=2 FS_STACK=17
