#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel-trace stats of configs[1] (bc2 res 1600, Jacobi 50), configs[4] (bc3 res 4096 KK + VC 10), the dye run (bc5 res 4096) and the mid grid bc2 res 1600
# -> gpurun_out/prof_r5_cfg1|cfg4|dye/...kernel_stats.csv; copied to profiles/r5_kernel_stats_{cfg1,cfg4,dye}.csv
set -u
export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5_$name -o t -- python3 bench.py "$@" --no-cpu --no-graph --sweeps 0 > gpurun_out/prof_r5_$name.json 2> gpurun_out/prof_r5_$name.err; }
run cfg1 --bc 2 --res 1600 --jacobi 50 --steps 60 --warmup 12
run cfg4 --bc 3 --scheme kk --vc 10 --re 1e8 --steps 40 --warmup 10
run dye --dye --steps 30 --warmup 10
run res1600 --bc 2 --res 1600 --steps 100 --warmup 20
find gpurun_out -name "t_kernel_stats.csv" | grep prof_r5_
