#!/bin/bash
# Run ON THE GPU BOX: parity of fs_cip_step (tests), then A/B of the headline step with / without K2 evaluated in registers
set -u
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_cip_step.py -q -m gpu 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg3 or dye" 2>&1 | tail -5
BENCH_ARGS="--steps 120 --warmup 20 --no-cpu --sweeps 0" bash tools/r3_ab.sh k2 "A1:FS_FUSE_K2=0" "B1:FS_FUSE_K2=1" "A2:FS_FUSE_K2=0" "B2:FS_FUSE_K2=1" | cut -c1-300 | tee gpurun_out/r5_k234_ab.txt
