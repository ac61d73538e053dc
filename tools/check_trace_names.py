#!/usr/bin/env python3
"""Contract check: every kernel the bench line names - roofline.gpu_kernels, poisson_jacobi_sweep.*.gpu_kernels and the
per-kernel gpu_kernels lists - appears in the rocprofv3 --kernel-trace --stats of THE SAME command, and its average duration there agrees with
the line's HIP-event average (the bench's own number) to within `tol`.

    tools/check_trace_names.py <bench line .json> <kernel_stats.csv> [tol=0.15]       exit 0 = every name found (durations are reported, not asserted,
                                                                                         beyond `tol` for the roofline kernel)"""
import csv
import json
import sys


def main():
    line, stats = sys.argv[1], sys.argv[2]
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
    d = json.loads([l for l in open(line).read().splitlines() if l.strip().startswith("{")][-1])
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}

    def find(sym):      # "fs::k_x<4, 5>" -> the trace's "void fs::k_x<4, 5>(fs::Grid, ...)"
        return [n for n in rows if n.startswith("void " + sym + "(") or n.startswith(sym + "(")]

    wanted = {}
    for name, k in d["kernels"].items():
        for sym in k.get("gpu_kernels") or []:
            wanted.setdefault(sym, []).append(f"kernels.{name}")
    for sym in d["roofline"].get("gpu_kernels") or []:
        wanted.setdefault(sym, []).append("roofline")
    jac = d.get("poisson_jacobi_sweep") or {}
    for key, leg in [("", jac)] + [(k, v) for k, v in jac.items() if isinstance(v, dict)]:
        for sym in leg.get("gpu_kernels") or []:
            wanted.setdefault(sym, []).append("poisson_jacobi_sweep" + ("." + key if key else ""))
    missing = []
    for sym, where in sorted(wanted.items()):
        hit = find(sym)
        if not hit:
            missing.append((sym, where))
            continue
        r = rows[hit[0]]
        print(f"ok   {sym:58s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.2f} us   <- {', '.join(where)}")
    for sym, where in missing:
        print(f"MISSING in the kernel trace: {sym}   <- {', '.join(where)}")
    rf = d["roofline"]
    syms = rf.get("gpu_kernels") or []
    if len(syms) == 1 and find(syms[0]):
        tr_us = float(rows[find(syms[0])[0]]["AverageNs"]) / 1e3
        rel = abs(tr_us - rf["avg_us"]) / rf["avg_us"]
        print(f"roofline kernel {syms[0]}: bench HIP-event average {rf['avg_us']} us, rocprofv3 average {tr_us:.2f} us ({rel:.1%} apart; tolerance {tol:.0%})")
        if rel > tol:
            missing.append((syms[0], ["duration disagrees"]))
    sys.exit(1 if missing else 0)


if __name__ == "__main__":
    main()
