#!/bin/bash
# Run ON THE GPU BOX: the boundary bodies after the register work (selective stores through the scalar row base, carried gradients re-read, the
# sibling's rows read from LDS where used): parity, then A/B of the committed build (tools/ab/lib_base.so) against the new one with k_cip_step_bnd at
# 4 waves per SIMD (the tree's library) and at 3 (tools/ab/lib_bndw3.so)
set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_cip_step.py tests/test_gpu_traj.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
A=$PWD/tools/ab/lib_base.so; B=$PWD/2d-fluid-simulator_amd/csrc/libfs_hip.so; C=$PWD/tools/ab/lib_bndw3.so
for cfg in "--steps 120 --warmup 20 --no-cpu --sweeps 0" "--dye --steps 100 --warmup 20 --no-cpu --sweeps 0" "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 1600 --bc 2 --dye --steps 200 --warmup 40 --no-cpu --sweeps 0" "--res 400 --bc 2 --dye --steps 2000 --warmup 50 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh bnd2 "A1:FS_LIB=$A" "B1:FS_LIB=$B" "C1:FS_LIB=$C" "A2:FS_LIB=$A" "B2:FS_LIB=$B" "C2:FS_LIB=$C" | cut -c1-330
done 2>&1 | tee gpurun_out/r5_bnd2.txt
