#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels: tools/isa_mix.py [name-substring ...]   (device-only -S of csrc/fs_transport.hip, fs_pressure.hip, fs_core.hip)"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "2d-fluid-simulator_amd", "csrc")


def main():
    lines = []
    for tu in ("fs_transport", "fs_pressure", "fs_core"):
        asm = f"/tmp/{tu}_isa.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w",
                        "-I/opt/rocm/include", "--cuda-device-only", "-S", os.path.join(CSRC, tu + ".hip"), "-o", asm], check=True)
        lines += open(asm).read().split("\n")
    subs = sys.argv[1:] or ["k_rbsor_pair", "k_cip_grad_advect_n", "k_vort_n", "k_cip_nonadv_n", "k_jacobi_ov", "k_jacobi_quad", "k_limit_quad"]
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN2fs\w+):", lines[i])
        if not m or not any(s in m.group(1) for s in subs):
            i += 1
            continue
        name = m.group(1)
        cnt = collections.Counter()
        i += 1
        while not lines[i].startswith(".Lfunc_end"):
            l = lines[i].strip()
            mm = re.match(r"^([a-z_0-9]+)\s", l)
            if mm:
                op = mm.group(1)
                kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                        "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "other")
                cnt[kind] += 1
                if op.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")):
                    cnt[op] += 1
                if "dpp" in l:
                    cnt["dpp"] += 1
                if op.startswith("v_cndmask"):
                    cnt["cndmask"] += 1
                if op.startswith("s_cbranch"):
                    cnt["branch"] += 1
            mv = re.search(r"; NumVgprs: (\d+)", l)
            i += 1
        # metadata follows the body
        meta = " ".join(lines[i:i + 60])
        vg = re.search(r"NumVgprs: (\d+)", meta)
        occ = re.search(r"Occupancy: (\d+)", meta)
        sp = re.search(r"ScratchSize: (\d+)", meta)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{demangled[:110]}\n    vgprs {vg.group(1) if vg else '?'} occ {occ.group(1) if occ else '?'} scratch {sp.group(1) if sp else '?'}  {dict(cnt)}")


if __name__ == "__main__":
    main()
