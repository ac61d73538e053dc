#!/bin/bash
# (Historical: FS_FUSE_K2 numbering of the time - 1 = the three-part form, since removed; 2 = two launches, today's 1.)
# Run ON THE GPU BOX: parity of fs_cip_step with K2 in registers on the boundary tiles too (FS_FUSE_K2=2), then A/B against the three-part form
set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_cip_step.py -q -m gpu -x 2>&1 | tail -15
BENCH_ARGS="--steps 120 --warmup 20 --no-cpu --sweeps 0" bash tools/r3_ab.sh bnd "A1:FS_FUSE_K2=1" "B1:FS_FUSE_K2=2" "A2:FS_FUSE_K2=1" "B2:FS_FUSE_K2=2" | cut -c1-400 | tee gpurun_out/r5_bnd_ab.txt
