#!/bin/bash
# Run ON THE GPU BOX: the red-black pair on mid grids - one launch of one-wave 4-row tiles (default below 8 M cells) against the two-part forms with
# plain tiles of 16 (two stacked waves), 8 and 4 rows
set -u
mkdir -p gpurun_out
for cfg in "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 1200 --bc 2 --steps 600 --warmup 40 --no-cpu --sweeps 0" "--res 2048 --bc 5 --steps 300 --warmup 40 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh mid3 "A1:FS_RBPAIR_SPLIT=1" "B1:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=16" "C1:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=8" "D1:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=4" "A2:FS_RBPAIR_SPLIT=1" "B2:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=16" "C2:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=8" "D2:FS_RBPAIR_SPLIT=2 FS_RBPAIR_PLAIN_RT=4" | cut -c1-230
done 2>&1 | tee gpurun_out/r5_mid3.txt
