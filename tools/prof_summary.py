#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one table per kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 B for wide (16 B/lane)
streaming reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled: hbm_read = 2 * FETCH_SIZE * 1024."""
import csv, glob, os, re, sys, collections
out = sys.argv[1]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"fs::", "", n)
    n = n.replace("k_cip_grad_advect_n<2,", "k_cip_grad_advect_rt<").replace("k_cip_grad_advect_n<3,", "k_cip_grad_advect_dye<")   # one body, two fields (fs_k34n.h)
    return re.sub(r"<.*", "", n)
stats = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        s = stats.setdefault(k, [0, 0.0])
        s[0] += int(r["Calls"]); s[1] += float(r["TotalDurationNs"])
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
full = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))     # short -> instantiation -> counter -> values
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            full[short(r["Kernel_Name"])][r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# kernels launched in two compact parts per logical launch (workgroups that see nothing but fluid / the others: two instantiations)
SPLIT = {"k_cip_grad_advect_dye", "k_jacobi_quad"}      # (round 6: the red-black pair is ONE launch over both kinds of tile - k_rbsor_pair_all) | ({"k_cip_grad_advect_rt"} if "k_cip_step_plain" not in stats else set())
import json
traffic, by_short = {}, {}
NAMES = {"k_rbsor_pair": "rbsor_pair", "k_jacobi_quad": "jacobi_quad_lazy", "k_cip_grad_advect_dye": "cip_grad_advect_dye", "k_mac_update_n": "mac_update_kk",
         "k_cip_grad_advect_rt": "cip_grad_advect_rt", "k_cip_advect_quad": "cip_advect", "k_rbsor_iter_n": "rbsor_iteration", "k_cip_nonadv_grad_quad": "cip_nonadv_grad",
         "k_cip_nonadv_n": "cip_nonadv", "k_vort_n": "vort_confine", "k_limit": "limit_field", "k_limit_quad": "limit_field",
         "k_jacobi_pair": "jacobi_pair_lazy", "k_jacobi_lazy": "jacobi_sweep_lazy",
         "k_cip_step_all": "cip_step", "k_cip_dye": "cip_step_dye", "k_rbsor_pair_all": "rbsor_pair"}      # (late round 5: fs_cip_step / fs_cip_step_dye as ONE launch over every tile)
print(f"{'kernel':28s} {'calls':>6s} {'avg_us':>9s} {'fetch_MB(x2)':>13s} {'write_MB':>9s} {'L2hit%':>7s} {'HBM GB/s':>9s}")
for k, (calls, tot) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    avg = tot / calls / 1e3
    c = pmc.get(k, {})
    mean = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else None
    fe, wr, hit, miss = mean("FETCH_SIZE"), mean("WRITE_SIZE"), mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    parts = len(full.get(k, {})) if k in SPLIT else 1
    if parts > 1:       # per LOGICAL launch: the sum over the parts (each part's mean), time likewise
        tot = lambda name: sum(sum(v[name]) / len(v[name]) for v in full[k].values() if v.get(name)) if any(v.get(name) for v in full[k].values()) else None
        fe, wr = tot("FETCH_SIZE"), tot("WRITE_SIZE")
        avg, calls = avg * parts, calls // parts
    fe_mb = None if fe is None else 2 * fe * 1024 / 1e6
    wr_mb = None if wr is None else wr * 1024 / 1e6
    hr = None if hit is None or miss is None or hit + miss == 0 else 100 * hit / (hit + miss)
    bw = None if fe_mb is None or wr_mb is None else (fe_mb + wr_mb) * 1e6 / (avg * 1e-6) / 1e9
    if fe_mb is not None and wr_mb is not None:
        by_short[k] = int((fe_mb + wr_mb) * 1e6)
        if k in NAMES:
            traffic[NAMES[k]] = by_short[k]
    f = lambda x, w, p=1: (f"{x:{w}.{p}f}" if x is not None else " " * (w - 1) + "-")
    print(f"{k:28s} {calls:6d} {avg:9.2f} {f(fe_mb,13)} {f(wr_mb,9)} {f(hr,7)} {f(bw,9,0)}")

# jacobi_tile is launched in two forms by bench.py (v-reading first, then source-pair): split by launch order
rows = []
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) in ("k_jacobi_tile", "k_jacobi_ov"):
                rows.append((r["Counter_Name"], "SRC" if "ILb1E" in r["Kernel_Name"] or "<true" in r["Kernel_Name"] else "V", float(r["Counter_Value"])))
for form, key in (("V", "jacobi_sweep"), ("SRC", "jacobi_sweep_src")):
    fe = [v for n, fo, v in rows if n == "FETCH_SIZE" and fo == form]
    wr = [v for n, fo, v in rows if n == "WRITE_SIZE" and fo == form]
    if fe and wr:
        traffic[key] = int((2 * sum(fe) / len(fe) + sum(wr) / len(wr)) * 1024)
# fs_cip_step: ONE kernel (k_cip_step_all); FS_FUSE_K2=1: two kernels per logical launch (csrc/fs_k234.h: the all-fluid tiles, the others)
parts3 = ("k_cip_step_plain", "k_cip_step_bnd")
if all(k in by_short for k in parts3):
    traffic["cip_step"] = sum(by_short[k] for k in parts3)
    traffic["cip_step_parts"] = {k: by_short[k] for k in parts3}
    traffic.pop("cip_grad_advect_rt", None); traffic.pop("cip_nonadv", None)
# the literal Jacobi sweep is its own kernel since round 5 (packed lanes of 2 cells, fs_jquad.h k_jacobi_ov2); k_jacobi_ov is then the source-pair form only
if "k_jacobi_ov2" in by_short:
    traffic["jacobi_sweep"] = by_short["k_jacobi_ov2"]
# fs_rbsor_pair: the all-fluid tiles as stacked two-wave workgroups (k_rbsor_pair_stack) + the boundary tiles (k_rbsor_pair)
parts2 = ("k_rbsor_pair_stack", "k_rbsor_pair")
if all(k in by_short for k in parts2):
    traffic["rbsor_pair"] = sum(by_short[k] for k in parts2)
    traffic["rbsor_pair_parts"] = {k: by_short[k] for k in parts2}
# VALU wave-instructions per launch (SQ_INSTS_VALU pass), per kernel and for the logical launches
valu = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_valu", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            valu[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
valu_per_launch = {k: sum(v) / len(v) for k, v in valu.items() if v}
if all(k in valu_per_launch for k in parts3):
    valu_per_launch["cip_step"] = sum(valu_per_launch[k] for k in parts3)
elif "k_cip_step_all" in valu_per_launch:
    valu_per_launch["cip_step"] = valu_per_launch["k_cip_step_all"]
if all(k in valu_per_launch for k in parts2):
    valu_per_launch["rbsor_pair"] = sum(valu_per_launch[k] for k in parts2)
elif "k_rbsor_pair_all" in valu_per_launch:      # round 6: one launch over both kinds of tile
    valu_per_launch["rbsor_pair"] = valu_per_launch["k_rbsor_pair_all"]
if "k_vort_n" in valu_per_launch:
    valu_per_launch["vort_confine"] = valu_per_launch["k_vort_n"]
# the stamp: these numbers belong to ONE build of the library - bench.py quotes them only when the library it loaded has this hash
import hashlib
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-fluid-simulator_amd", "csrc", "libfs_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
json.dump({"note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from rocprofv3 --pmc passes (gfx950 FETCH_SIZE correction x2), "
                   "workload bc5 res4096 cip+vc, see tools/profile.sh; valu_wave_insts_per_launch = SQ_INSTS_VALU of the same workload",
           "lib_sha256": sha, "bytes_per_launch": traffic, "valu_wave_insts_per_launch": {k: int(v) for k, v in valu_per_launch.items()}},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
