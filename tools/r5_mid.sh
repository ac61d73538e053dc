#!/bin/bash
# Run ON THE GPU BOX: fs_cip_step's three-part form on mid grids.  (Historical: FS_K234_CELLS - the threshold in cells - was an env switch while this
# ran; it is the constant 8 M cells since - cip_step_three_parts in csrc/fs_transport.hip.  Numbers: DESIGN.md section 5.)
set -u
for cfg in "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 1600 --bc 2 --jacobi 50 --steps 200 --warmup 20 --no-cpu --sweeps 0" "--res 1200 --bc 2 --steps 600 --warmup 40 --no-cpu --sweeps 0" "--res 800 --bc 2 --steps 1000 --warmup 40 --no-cpu --sweeps 0" "--res 2048 --bc 5 --steps 300 --warmup 40 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh mid "A1:FS_K234_CELLS=8388608" "B1:FS_K234_CELLS=0" "A2:FS_K234_CELLS=8388608" "B2:FS_K234_CELLS=0" | cut -c1-330
done
