#!/bin/bash
# kernel timeline of small-grid graph replays (ON THE GPU BOX): per-kernel duration and the gap to the previous kernel
export TMPDIR=/tmp
for cfg in "res400 --bc 2 --res 400" "cfg0 --bc 1 --res 200 --scheme upwind --vc 0 --re 1000 --dt 0.0005" "res400dye --bc 2 --res 400 --dye"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/tr_$name
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 bench.py "$@" --steps 600 --warmup 60 --sweeps 0 --no-cpu > /dev/null 2>&1
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$name" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-6000:]                      # the timed replays at the end
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:60]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev_end is not None and s - prev_end < 20000: gap[n].append(s - prev_end)
    prev_end = e
print("==", sys.argv[2], "kernels", len(rows))
tot = 0
for n in dur:
    d = sorted(dur[n]); g = sorted(gap[n]) or [0]
    print(f"  {n:60s} n {len(d):5d}  dur med {d[len(d)//2]/1e3:6.2f} us   gap before med {g[len(g)//2]/1e3:6.2f} us")
P
done
