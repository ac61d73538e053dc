#!/bin/bash
# Run ON THE GPU BOX, ONE lease (gpurun -- 'bash tools/r4_lease.sh'): everything profiles/r4_* is made of - the box's own read / copy rate,
# one bench.py JSON per BASELINE configuration (+ the reference's default resolution with and without dye), rocprofv3 kernel stats +
# PMC traffic + SQ wave-cycle split of the headline run.  Output: gpurun_out/r4/...
set -u
OUT=gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp
tools/membw.bin > $OUT/membw.txt 2>&1
bash tools/bench_configs.sh r4 > $OUT/bench_configs.log 2>&1
run() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_r4/$name.json 2> gpurun_out/bench_r4/$name.err; }
run res400_bc2_cip_vc      --bc 2 --res 400 --steps 6000 --warmup 100 --sweeps 0 --no-cpu
run res400_bc2_cip_vc_dye  --bc 2 --res 400 --dye --steps 6000 --warmup 100 --sweeps 0 --no-cpu
bash tools/profile.sh r4 > $OUT/profile.log 2>&1
EXTRA_PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/r3_pmc.sh r4 > $OUT/pmc.log 2>&1
tools/membw.bin > $OUT/membw_after.txt 2>&1
ls gpurun_out/bench_r4 gpurun_out/prof_r4 gpurun_out/pmc_r4 > $OUT/files.txt 2>&1
tail -5 $OUT/bench_configs.log
