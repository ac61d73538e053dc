#!/bin/bash
# Run ON THE GPU BOX, ONE lease (gpurun -- 'bash tools/r4_lease.sh'): everything profiles/r4_* is made of - the box's own read / copy rate,
# one bench.py JSON per BASELINE configuration (+ the reference's default resolution with and without dye), rocprofv3 kernel stats +
# PMC traffic + SQ wave-cycle split of the headline run, the loop-back slab step, the kernel timeline of the small grids.
# Output: gpurun_out/r4/...; tools/r4_collect.sh (run locally afterwards) assembles profiles/r4_* from it.
set -u
OUT=gpurun_out/r4; mkdir -p $OUT gpurun_out/bench_r4; export TMPDIR=/tmp
tools/membw.bin > $OUT/membw.txt 2>&1
bash tools/bench_configs.sh r4 > $OUT/bench_configs.log 2>&1
run() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_r4/$name.json 2> gpurun_out/bench_r4/$name.err; }
run res400_bc2_cip_vc      --bc 2 --res 400 --steps 6000 --warmup 100 --sweeps 0 --no-cpu
run res400_bc2_cip_vc_dye  --bc 2 --res 400 --dye --steps 6000 --warmup 100 --sweeps 0 --no-cpu
run res1600_bc2_cip_vc     --bc 2 --res 1600 --steps 1200 --warmup 60 --sweeps 0 --no-cpu
bash tools/profile.sh r4 > $OUT/profile.log 2>&1
EXTRA_PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/r3_pmc.sh r4 > $OUT/pmc.log 2>&1
{ echo "== defaults (halo 20, pair on)"; timeout 600 python3 tools/overlap_bench.py 16 20 2>&1 | tail -8; echo "== FS_RBSOR_PAIR=0"; FS_RBSOR_PAIR=0 timeout 600 python3 tools/overlap_bench.py 16 20 2>&1 | tail -8; } > $OUT/loopback.txt 2>&1
bash tools/r4_small_trace.sh > $OUT/small_trace.txt 2>&1
for e in X=1 FS_RBPAIR_RT=4 FS_SMALL_TILES=0; do echo "== $e"; env $e python3 tools/r4_chain.py 2>&1 | tail -4; done > $OUT/chain.txt 2>&1
tools/membw.bin > $OUT/membw_after.txt 2>&1
ls gpurun_out/bench_r4 gpurun_out/prof_r4 gpurun_out/pmc_r4 > $OUT/files.txt 2>&1
tail -5 $OUT/bench_configs.log
