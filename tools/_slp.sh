b() { echo "== $*"; env "$@" python bench.py --steps 120 --warmup 20 --no-cpu --sweeps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
b FS_X=1
b FS_LIB=$PWD/tools/libfs_noslp.so
b FS_X=1
b FS_LIB=$PWD/tools/libfs_noslp.so
echo cfg4; for l in "" "$PWD/tools/libfs_noslp.so"; do FS_LIB=$l python tools/kbench.py --bc 3 --scheme kk --vc 10 --steps 20 --warm 30 --sweeps 0 2>&1 | grep -E "mac_update|vort|rbsor"; done
