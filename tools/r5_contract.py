#!/usr/bin/env python3
"""Price of bit-exactness: the library built with FMA contraction inside the CIP polynomial (EXTRA=-DFS_CONTRACT_CIP, FS_LIB=...) against the
default build - rel-L2 of v and p after 1 .. 20 steps at bc5 res 4096, vorticity confinement off / on.  usage: r5_contract.py dump <out.npz> | cmp <a.npz> <b.npz>"""
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
STEPS = (1, 2, 5, 10, 20)


def dump(path):
    importlib.import_module("2d-fluid-simulator_amd")
    import fs
    res = 4096
    out = {}
    for vc in (None, 5.0):
        fs.runtime.init(gpu=0, dtype="f32")
        sim = fs.FluidSimulator.create(5, res, 0.05 / res, 1.0 / res, 1.0e6, vc, "cip")
        for step in range(1, max(STEPS) + 1):
            sim.step()
            if step in STEPS:
                f = sim.field_to_numpy()
                out[f"vc{vc}.step{step}.v"], out[f"vc{vc}.step{step}.p"] = f["v"], f["p"]
        sim._solver._bc.device.close()
    np.savez(path, **out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    for vc in (None, 5.0):
        parts = []
        for step in STEPS:
            r = []
            for k in ("v", "p"):
                x, y = A[f"vc{vc}.step{step}.{k}"].astype(np.float64), B[f"vc{vc}.step{step}.{k}"].astype(np.float64)
                r.append(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
            parts.append(f"step{step}: v {r[0]:.2e} p {r[1]:.2e}")
        print(f"[contract(fast) in cip_point vs bit-exact, bc5 res4096 cip vc={vc}] " + "  ".join(parts))


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
