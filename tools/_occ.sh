b() { echo "== $*"; env "$@" python bench.py --steps 120 --warmup 20 --no-cpu --sweeps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
b FS_X=occ4
b FS_LIB=$PWD/tools/libfs_noslp.so
b FS_X=occ4
b FS_LIB=$PWD/tools/libfs_noslp.so
