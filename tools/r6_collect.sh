#!/bin/bash
# Run LOCALLY after `gpurun -- 'bash tools/r6_lease.sh'`: assemble profiles/r6_* from what the lease wrote into gpurun_out/.
set -u
cd "$(dirname "$0")/.."
for f in gpurun_out/bench_r6/*.json; do cp "$f" profiles/r6_bench_$(basename "$f"); done
cp gpurun_out/prof_r6/trace/t_kernel_stats.csv profiles/r6_kernel_stats.csv
cp gpurun_out/prof_r6/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/prof_r6/bench_trace.json profiles/r6_bench_under_rocprof.json
cp gpurun_out/pmc_r6/summary.txt profiles/r6_sq_wave_cycles.txt
{ echo "tools/overlap_bench.py, round 6: one slab (rank 3 of 8; rank 1 of 4 / of 2) of bc5 res 4096 on ONE MI355X, ghost-row exchanges through RCCL in loop-back (the rank is its own neighbour)."
  echo "none = exchanges removed (compute only); blocking = in line on the compute stream; tape = the recorded period replayed from C++.  One box, one call (tools/r6_lease.sh)."
  cat gpurun_out/r6/loopback.txt; } > profiles/r6_loopback_slab_step.txt
python3 - <<'PY' > profiles/r6_summary.txt
import json, glob, os, re
def first(path, pat):
    for l in open(path):
        m = re.search(pat, l)
        if m: return m.group(1)
    return "?"
r0, c0 = first("gpurun_out/r6/membw.txt", r"read  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s"), first("gpurun_out/r6/membw.txt", r"copy  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s")
r1, c1 = first("gpurun_out/r6/membw_after.txt", r"read  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s"), first("gpurun_out/r6/membw_after.txt", r"copy  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s")
head = json.load(open("gpurun_out/bench_r6/cfg2_bc5_res4096_cip_vc.json"))
print("Round 6, ONE lease (tools/r6_lease.sh, assembled by tools/r6_collect.sh): every number below comes from the same MI355X box, back to back.\n")
print(f"This box (tools/membw.hip before / after the runs): float4 read {r0} / {r1} GB/s, float4 copy {c0} / {c1} GB/s (read + write, 537 MB buffers);")
print(f"the library's own fs_box_rates (bench.py \"box\", 268 MB buffers): read {head['box']['read_GBps']:.0f}, copy {head['box']['copy_GBps']:.0f} GB/s (k_box_read / k_box_copy rows below).\n")
print("== rocprofv3 --kernel-trace --stats + PMC passes (FETCH_SIZE doubled per the gfx950 correction; Infinity-Cache hits count as fetches) of")
print("   python3 bench.py --steps 20 --warmup 10 --no-cpu --no-graph --sweeps 40   (tools/profile.sh; profiles/r6_kernel_stats.csv, profiles/pmc_traffic.json)")
print(open("gpurun_out/prof_r6/summary.txt").read().rstrip())
print("\n== SQ wave-cycle split + instruction counts per wave + traffic, per kernel instantiation (tools/r3_pmc.sh; fs_cip_step and - since round 6 - fs_rbsor_pair are ONE launch over both kinds of tile: k_cip_step_all, k_rbsor_pair_all)")
print("   wait = parked on s_waitcnt, stall = ready but not issued, act = issuing; qc/wave = quad-cycles a wave is resident")
print(open("gpurun_out/pmc_r6/summary.txt").read().rstrip())
print("\n== steps/s of every BASELINE configuration on this box (profiles/r6_bench_*.json)")
for f in sorted(glob.glob("gpurun_out/bench_r6/*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    print(f"{os.path.basename(f)[:-5]:34s} {d['value']:10.1f} steps/s  {d['ms_per_step']:.4f} ms/step  box copy {d['box']['copy_GBps']:.0f} GB/s  [{d['config']['launch'][:60]}]")
print("\n== loop-back slab step: profiles/r6_loopback_slab_step.txt")
PY
ls -la profiles/r6_* | head -30
