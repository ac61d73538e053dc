#!/bin/bash
# Run LOCALLY after `gpurun -- 'bash tools/r4_lease.sh'`: assemble profiles/r4_* from what the lease wrote into gpurun_out/.
set -u
cd "$(dirname "$0")/.."
for f in gpurun_out/bench_r4/*.json; do cp "$f" profiles/r4_bench_$(basename "$f"); done
cp gpurun_out/prof_r4/trace/t_kernel_stats.csv profiles/r4_kernel_stats.csv
cp gpurun_out/prof_r4/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/prof_r4/bench_trace.json profiles/r4_bench_under_rocprof.json
cp gpurun_out/pmc_r4/summary.txt profiles/r4_sq_wave_cycles.txt
{ echo "tools/overlap_bench.py, round 4: middle slab (rank 3 of 8) of bc5 res 4096 on ONE MI355X, ghost-row exchanges through RCCL in loop-back (the rank is its own neighbour)."
  echo "none = exchanges removed (compute only); blocking = in line on the compute stream; tape = the recorded period replayed from C++.  One box, one call (tools/r4_lease.sh)."
  cat gpurun_out/r4/loopback.txt; } > profiles/r4_loopback_slab_step.txt
{ echo "Kernel timeline of graph-replayed small grids (tools/r4_small_trace.sh: rocprofv3 --kernel-trace; a kernel's duration includes its dispatch) and the marginal"
  echo "cost of one more launch of the same kernel inside a graph (tools/r4_chain.py, res 200): what a small grid's step is made of (DESIGN.md section 6)."
  cat gpurun_out/r4/small_trace.txt; echo; cat gpurun_out/r4/chain.txt; } > profiles/r4_small_grids.txt
python3 - <<'PY' > profiles/r4_summary.txt
import json, glob, os, re
def first(path, pat):
    for l in open(path):
        m = re.search(pat, l)
        if m: return m.group(1)
    return "?"
r0, c0 = first("gpurun_out/r4/membw.txt", r"read  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s"), first("gpurun_out/r4/membw.txt", r"copy  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s")
r1, c1 = first("gpurun_out/r4/membw_after.txt", r"read  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s"), first("gpurun_out/r4/membw_after.txt", r"copy  tile U=1\s*:\s*[\d.]+ us\s+(\d+) GB/s")
head = json.load(open("gpurun_out/bench_r4/cfg2_bc5_res4096_cip_vc.json"))
print("Round 4, ONE lease (tools/r4_lease.sh, assembled by tools/r4_collect.sh): every number below comes from the same MI355X box, back to back.\n")
print(f"This box (tools/membw.hip before / after the runs): float4 read {r0} / {r1} GB/s, float4 copy {c0} / {c1} GB/s (read + write, 537 MB buffers);")
print(f"the library's own fs_box_rates (bench.py \"box\", 268 MB buffers): read {head['box']['read_GBps']:.0f}, copy {head['box']['copy_GBps']:.0f} GB/s (k_box_read / k_box_copy rows below).\n")
print("== rocprofv3 --kernel-trace --stats + PMC passes (FETCH_SIZE doubled per the gfx950 correction; Infinity-Cache hits count as fetches) of")
print("   python3 bench.py --steps 20 --warmup 10 --no-cpu --no-graph --sweeps 40   (tools/profile.sh; profiles/r4_kernel_stats.csv, profiles/pmc_traffic.json)")
print(open("gpurun_out/prof_r4/summary.txt").read().rstrip())
print("\n== SQ wave-cycle split + instruction counts per wave + traffic, per kernel instantiation (tools/r3_pmc.sh; two-part launches: <..., true, ...> / PATH 3 = plain part, false / 2 = boundary part)")
print("   wait = parked on s_waitcnt, stall = ready but not issued, act = issuing; qc/wave = quad-cycles a wave is resident")
print(open("gpurun_out/pmc_r4/summary.txt").read().rstrip())
print("\n== steps/s of every BASELINE configuration on this box (profiles/r4_bench_*.json)")
for f in sorted(glob.glob("gpurun_out/bench_r4/*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    print(f"{os.path.basename(f)[:-5]:34s} {d['value']:10.1f} steps/s  {d['ms_per_step']:.4f} ms/step  box copy {d['box']['copy_GBps']:.0f} GB/s  [{d['config']['launch'][:60]}]")
print("\n== loop-back slab step: profiles/r4_loopback_slab_step.txt; small grids: profiles/r4_small_grids.txt")
PY
ls -la profiles/r4_* | head -30
