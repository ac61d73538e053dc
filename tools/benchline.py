#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> one short line (steps/s, launch form, per-kernel us x launches per step)"""
import json
import sys
d = json.loads(sys.stdin.read())
ks = d["kernels"]
print(f"{d['value']:10.1f} steps/s  [{d['config']['launch'][-70:]}]  " + "  ".join(f"{k}={v['avg_us']}x{v['launches_per_step']:g}" for k, v in ks.items()))
