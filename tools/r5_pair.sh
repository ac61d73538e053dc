#!/bin/bash
# Run ON THE GPU BOX: parity of the stacked two-wave form of the pair pass's plain part, then A/B against the 8-row one-wave form
set -u
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_rbpair.py tests/test_gpu_cip_step.py -q -m gpu -x 2>&1 | tail -6
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg3 or cfg4 or cfg5" 2>&1 | tail -5
BENCH_ARGS="--steps 120 --warmup 20 --no-cpu --sweeps 0" bash tools/r3_ab.sh pair "A1:FS_RBPAIR_PLAIN_RT=8" "B1:FS_RBPAIR_PLAIN_RT=16" "A2:FS_RBPAIR_PLAIN_RT=8" "B2:FS_RBPAIR_PLAIN_RT=16" | cut -c1-300 | tee gpurun_out/r5_pair_ab.txt
BENCH_ARGS="--res 4096 --bc 3 --scheme kk --vc 10 --re 1e8 --steps 120 --warmup 20 --no-cpu --sweeps 0" bash tools/r3_ab.sh pair "A1:FS_RBPAIR_PLAIN_RT=8" "B1:FS_RBPAIR_PLAIN_RT=16" | cut -c1-300
