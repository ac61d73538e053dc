#!/bin/bash
# (Historical: the two variant libraries were built by editing launch_k34's geometry line - `const int N = 2;`, RT = 4 from 2 M cells - and are not kept.)
# Run ON THE GPU BOX: K3+K4 on mid grids - quads x 2 rows (scalar) against pairs x 2 / x 4 rows (packed): tools/ab/lib_n2.so, lib_n2rt4.so
set -u
A=$PWD/2d-fluid-simulator_amd/csrc/libfs_hip.so; B=$PWD/tools/ab/lib_n2.so; C=$PWD/tools/ab/lib_n2rt4.so
for cfg in "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 1200 --bc 2 --steps 600 --warmup 40 --no-cpu --sweeps 0" "--res 1024 --bc 5 --steps 600 --warmup 40 --no-cpu --sweeps 0" "--res 1600 --bc 2 --dye --steps 200 --warmup 40 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh mid "A1:FS_LIB=$A" "B1:FS_LIB=$B" "C1:FS_LIB=$C" "A2:FS_LIB=$A" "B2:FS_LIB=$B" "C2:FS_LIB=$C" | cut -c1-330
done
