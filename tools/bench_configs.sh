#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/bench_configs.sh <tag>'): bench.py for every BASELINE.json configuration that fits one
# GPU (+ the dye mode the reference's main.py runs by default), one JSON line each -> gpurun_out/bench_<tag>/; copy to profiles/.
#   configs[0] bc1 res 200 upwind Re 1000 dt 5e-4 (VC off)      configs[1] bc2 res 1600 CIP, 50 Jacobi sweeps / step
#   configs[2] bc5 res 4096 CIP + VC (the headline)              configs[3] bc2 res 8192 CIP on ONE GPU
#   configs[4] bc3 res 4096 KK + VC 10 Re 1e8, f32 and f64
set -u
TAG=${1:-r2}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
run() { name=$1; shift; python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(head -c 300 $OUT/$name.json)"; }
run cfg0_bc1_res200_upwind   --bc 1 --res 200 --scheme upwind --vc 0 --re 1000 --dt 0.0005 --steps 20000 --warmup 200 --sweeps 0 --cpu-seconds 5
run cfg1_bc2_res1600_jacobi50 --bc 2 --res 1600 --jacobi 50 --steps 400 --warmup 40 --sweeps 0 --cpu-seconds 8
run cfg2_bc5_res4096_cip_vc  --steps 200 --warmup 40
run cfg3_bc2_res8192_cip_vc  --bc 2 --res 8192 --steps 60 --warmup 10 --sweeps 0 --no-cpu
run cfg4_bc3_res4096_kk_vc10 --bc 3 --scheme kk --vc 10 --re 1e8 --steps 200 --warmup 40 --sweeps 0 --no-cpu
run cfg4_bc3_res4096_kk_vc10_f64 --bc 3 --scheme kk --vc 10 --re 1e8 --dtype f64 --steps 100 --warmup 20 --sweeps 0 --no-cpu
run dye_bc5_res4096_cip_vc   --dye --steps 100 --warmup 20 --sweeps 0 --no-cpu
