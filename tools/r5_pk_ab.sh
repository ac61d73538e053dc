#!/bin/bash
# (tools/ab/lib_pk0.so = the library built with EXTRA=-DFS_K34_PK=0, the scalar bodies; not kept in the tree.)
# Run ON THE GPU BOX: parity of the packed K3+K4 body, then interleaved A/B against the scalar build (tools/ab/lib_pk0.so)
set -u
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_traj.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5
bash tools/r4_ab_lib.sh tools/ab/lib_pk0.so 2d-fluid-simulator_amd/csrc/libfs_hip.so 2>&1 | tee gpurun_out/r5_pk_ab.txt
