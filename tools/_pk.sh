timeout 900 python -m pytest tests/test_gpu_traj.py tests/test_gpu_random_masks.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_fast_paths_equal_reference_literal_kernels" -m gpu -q -x 2>&1 | tail -2
b() { echo "== $*"; env "$@" python bench.py --steps 120 --warmup 20 --no-cpu --sweeps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
b FS_X=1
b FS_X=2
