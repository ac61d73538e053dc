#!/bin/bash
# Run ON THE GPU BOX: A/B of bench.py under environment switches -> gpurun_out/ab_<tag>/<name>.json   (tools/r3_ab.sh tag "NAME:ENV=1 ENV2=x" ...)
set -u
TAG=$1; shift
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
ARGS=${BENCH_ARGS:---steps 120 --warmup 20 --no-cpu --sweeps 0}
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ks = d["kernels"]
    print(f"{sys.argv[2]:28s} {d['value']:9.1f} steps/s  " + "  ".join(f"{k}={v['avg_us']:.1f}x{v['launches_per_step']:g}" for k, v in ks.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
P
done
