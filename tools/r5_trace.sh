#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace stats of the headline bench (eager launches), per kernel INSTANTIATION -> gpurun_out/r5_trace_<tag>/stats.txt
set -u
TAG=${1:-a}
OUT=gpurun_out/r5_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ARGS="bench.py --steps 20 --warmup 10 --no-cpu --no-graph --sweeps ${SWEEPS:-0} ${BENCH_EXTRA:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/bench.json 2> $OUT/trace.err
python3 - $OUT <<'P'
import csv, glob, os, re, sys
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Name"]); n = re.sub(r"\(.*", "", n).replace("fs::", "")
        rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), n))
rows.sort(reverse=True)
with open(os.path.join(out, "stats.txt"), "w") as fh:
    for tot, calls, n in rows[:40]:
        line = f"{tot / calls / 1e3:10.2f} us x {calls:6d}  {n[:150]}"
        print(line); fh.write(line + "\n")
P
