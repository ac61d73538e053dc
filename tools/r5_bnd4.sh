#!/bin/bash
# (Historical: FS_FUSE_K2 numbering of the time - 2 = two launches (today's 1), 3 = one launch (today's 2, the default).)
# Run ON THE GPU BOX: the dye's step as two launches (FS_FUSE_K2=2) against one launch over both kinds of tile (3); parity of all three forms first
set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_cip_step.py -q -m gpu -x 2>&1 | tail -3
A=$PWD/tools/ab/lib_base.so
for cfg in "--dye --steps 100 --warmup 20 --no-cpu --sweeps 0" "--res 1600 --bc 2 --dye --steps 200 --warmup 40 --no-cpu --sweeps 0" "--res 1200 --bc 2 --steps 600 --warmup 40 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh bnd4 "A1:FS_LIB=$A" "B1:FS_FUSE_K2=2" "C1:FS_FUSE_K2=3" "A2:FS_LIB=$A" "B2:FS_FUSE_K2=2" "C2:FS_FUSE_K2=3" | cut -c1-330
done 2>&1 | tee gpurun_out/r5_bnd4.txt
