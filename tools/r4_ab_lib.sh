#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of two builds of libfs_hip.so (FS_LIB) on several configurations: tools/r4_ab_lib.sh <libA> <libB>
set -u
A=$PWD/$1; B=$PWD/$2
for cfg in "--steps 120 --warmup 20 --no-cpu --sweeps 0" "--res 1600 --bc 2 --jacobi 50 --steps 200 --warmup 20 --no-cpu --sweeps 0" "--res 400 --bc 2 --dye --steps 2000 --warmup 50 --no-cpu --sweeps 0" "--res 200 --bc 1 --scheme upwind --vc 0 --re 1000 --dt 0.0005 --steps 4000 --warmup 50 --no-cpu --sweeps 0" "--res 4096 --bc 3 --scheme kk --vc 10 --re 1e8 --steps 120 --warmup 20 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh lib "A1:FS_LIB=$A" "B1:FS_LIB=$B" "A2:FS_LIB=$A" "B2:FS_LIB=$B" "A3:FS_LIB=$A" "B3:FS_LIB=$B" | cut -c1-260
done
