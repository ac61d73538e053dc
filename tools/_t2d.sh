(timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_rccl_overlap.py -m gpu -q 2>&1 | tail -5)
run() { echo "== $*"; env "$@" python tools/kbench.py --steps 20 --warm 30 --sweeps 60 2>&1 | grep -v "^#" | grep -v "_bc\|poisson_source"; }
run FS_TILE2D=0
run FS_TILE2D=63 FS_TILE_ROWS=32 FS_TILE_WAVES=4
run FS_TILE2D=63 FS_TILE_ROWS=64 FS_TILE_WAVES=4
run FS_TILE2D=63 FS_TILE_ROWS=32 FS_TILE_WAVES=8
run FS_TILE2D=63 FS_TILE_ROWS=64 FS_TILE_WAVES=8
run FS_TILE2D=63 FS_TILE_ROWS=128 FS_TILE_WAVES=4
run FS_TILE2D=63 FS_TILE_ROWS=16 FS_TILE_WAVES=8
