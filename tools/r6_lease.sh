#!/bin/bash
# Run ON THE GPU BOX, ONE lease (gpurun -- 'bash tools/r6_lease.sh'): everything profiles/r6_* is made of - the box's own read / copy rate,
# one bench.py JSON per BASELINE configuration (+ the reference's default resolution with and without dye, the mid grids), rocprofv3 kernel
# stats + PMC traffic + VALU counts (stamped with the library's hash) + SQ wave-cycle split of the headline run, the loop-back slab step.
# Output: gpurun_out/r6/...; tools/r6_collect.sh (run locally afterwards) assembles profiles/r6_* from it.
set -u
OUT=gpurun_out/r6; mkdir -p $OUT gpurun_out/bench_r6; export TMPDIR=/tmp
tools/membw.bin > $OUT/membw.txt 2>&1
bash tools/bench_configs.sh r6 > $OUT/bench_configs.log 2>&1
run() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_r6/$name.json 2> gpurun_out/bench_r6/$name.err; }
run res400_bc2_cip_vc      --bc 2 --res 400 --steps 6000 --warmup 100 --sweeps 0 --no-cpu
run res400_bc2_cip_vc_dye  --bc 2 --res 400 --dye --steps 6000 --warmup 100 --sweeps 0 --no-cpu
run res800_bc2_cip_vc      --bc 2 --res 800 --steps 3000 --warmup 60 --sweeps 0 --no-cpu
run res1600_bc2_cip_vc     --bc 2 --res 1600 --steps 1200 --warmup 60 --sweeps 0 --no-cpu
run res1600_bc2_cip_vc_dye --bc 2 --res 1600 --dye --steps 600 --warmup 60 --sweeps 0 --no-cpu
bash tools/profile.sh r6 > $OUT/profile.log 2>&1
EXTRA_PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/r3_pmc.sh r6 > $OUT/pmc.log 2>&1
{ echo "== defaults (halo 20, pair on): middle slab of the 8-way cut"; timeout 600 python3 tools/overlap_bench.py 16 20 2>&1 | tail -8;
  for w in "4 1" "2 1"; do set -- $w; echo "== slab $2 of the $1-way cut (none = compute only, tape = the recorded period with its exchanges)"; OB_MODES=none,tape OB_WORLD=$1 OB_RANK=$2 timeout 300 python3 tools/overlap_bench.py 20 2>&1 | tail -2; done
  echo "== the 8- / 4- / 2-way cuts with fs_cip_step as its two calls (FS_FUSE_K2=0: K2 as a launch of its own, as before round 5 on slabs)"
  for w in "8 3" "4 1" "2 1"; do set -- $w; OB_MODES=tape OB_WORLD=$1 OB_RANK=$2 FS_FUSE_K2=0 timeout 300 python3 tools/overlap_bench.py 20 2>&1 | tail -1 | sed "s/^/$1-way: /"; done; } > $OUT/loopback.txt 2>&1
tools/membw.bin > $OUT/membw_after.txt 2>&1
ls gpurun_out/bench_r6 gpurun_out/prof_r6 gpurun_out/pmc_r6 > $OUT/files.txt 2>&1
tail -8 $OUT/bench_configs.log
