#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/overlap_bench.py: for the last N kernels print name, start offset, duration (us)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
