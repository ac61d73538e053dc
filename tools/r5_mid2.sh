#!/bin/bash
# Run ON THE GPU BOX: mid grids with the large grids' launch forms (FS_RBPAIR_SPLIT=2: fs_cip_step with K2 in registers - two launches since the
# boundary tiles evaluate it too - and the two-part red-black pair) against their own defaults; the dye's general K3 + K4 kernel at 3 waves
# per SIMD without spills (tools/ab/lib_dyew3.so) against 4 with 20 - 52 bytes of scratch
set -u
mkdir -p gpurun_out
L=$PWD/2d-fluid-simulator_amd/csrc/libfs_hip.so; W3=$PWD/tools/ab/lib_dyew3.so
for cfg in "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 1200 --bc 2 --steps 600 --warmup 40 --no-cpu --sweeps 0" "--res 800 --bc 2 --steps 1000 --warmup 40 --no-cpu --sweeps 0" "--res 2048 --bc 5 --steps 300 --warmup 40 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh mid2 "A1:FS_RBPAIR_SPLIT=1" "B1:FS_RBPAIR_SPLIT=2" "A2:FS_RBPAIR_SPLIT=1" "B2:FS_RBPAIR_SPLIT=2" | cut -c1-330
done 2>&1 | tee gpurun_out/r5_mid2.txt
for cfg in "--res 1600 --bc 2 --dye --steps 200 --warmup 40 --no-cpu --sweeps 0" "--res 400 --bc 2 --dye --steps 2000 --warmup 50 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh mid2d "A1:FS_LIB=$L" "B1:FS_LIB=$W3" "C1:FS_LIB=$L FS_RBPAIR_SPLIT=2" "A2:FS_LIB=$L" "B2:FS_LIB=$W3" "C2:FS_LIB=$L FS_RBPAIR_SPLIT=2" | cut -c1-330
done 2>&1 | tee -a gpurun_out/r5_mid2.txt
