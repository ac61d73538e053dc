#!/usr/bin/env python3
"""Differential fuzzing of the slab path: random grid / mask / scheme / updater, cut into a random number of slabs with a random
halo depth, N slab contexts on ONE GPU driven by threads (host-copied ghost rows, partial depth like the RCCL leg), against the
CPU oracle on the undivided grid - bit for bit.    python tools/fuzz_slabs.py [--cases 100] [--seed 0]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
importlib.import_module("2d-fluid-simulator_amd")
from oracle import oracle as O  # noqa: E402
import fuzz_parity  # noqa: E402
import test_gpu_slab_threads as T  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 6))
    halo = int(rng.choice([2, 3, 4, 6, 8, 16]))
    nyl_min = max(halo, 4) + int(rng.integers(0, 40 if rng.random() < 0.7 else 160))
    Y = world * nyl_min + int(rng.integers(0, world))
    X = int(rng.choice([rng.integers(4, 80), 4 * rng.integers(1, 150), rng.integers(80, 700)]))
    const, mask, dye = fuzz_parity.random_scene(rng, X, Y)
    scheme = str(rng.choice(["cip", "cip", "kk", "upwind"]))
    with_dye = bool(rng.random() < 0.3)
    f64 = bool(rng.random() < 0.2)
    dtype = np.float64 if f64 else np.float32
    vc = None if rng.random() < 0.3 else float(rng.choice([0.5, 5.0, 10.0]))
    updater = ("rbsor", float(rng.choice([1.0, 1.3])), int(rng.integers(1, 4))) if rng.random() < 0.6 else ("jacobi", int(rng.choice([1, 4, 9, 14])))
    res = float(rng.choice([2 ** rng.integers(3, 10), rng.integers(10, 900)]))
    os.environ["FS_FUSE_TRANSPORT"] = "1" if rng.random() < 0.2 else "0"
    os.environ["FS_OVERLAP"] = "1" if rng.random() < 0.7 else "0"
    steps = int(rng.integers(2, 7))
    for k in ("FS_FUSE_TRANSPORT", "FS_OVERLAP"):          # FORCE_<knob>=0/1 overrides the drawn value (diagnosis)
        if os.environ.get("FORCE_" + k):
            os.environ[k] = os.environ["FORCE_" + k]
    if os.environ.get("FORCE_HALO"):
        halo = int(os.environ["FORCE_HALO"])
        if Y // world < halo:               # (a forced depth the drawn grid cannot carry: not a case)
            return None
    cfg = dict(bc=0, res=res, dt=0.05 / res, dx=1.0 / res, re=float(rng.choice([100.0, 1e6])), vor_eps=vc, scheme=scheme, updater=updater,
               dye=with_dye, fp64=f64, snaps=[steps])
    desc = (f"seed {seed}: {X}x{Y} world={world} halo={halo} {np.dtype(dtype).name} {scheme} vc={vc} {updater} dye={with_dye} steps={steps} "
            f"fuse={os.environ['FS_FUSE_TRANSPORT']} overlap={os.environ['FS_OVERLAP']}")
    g = {"bc_const": const.astype(dtype), "bc_mask": mask, "bc_dye": dye.astype(dtype)}
    try:
        results = T._run_slabs(g, cfg, world, halo)
    except AssertionError as exc:
        msg = str(exc)
        if "boundary kernels reach" in msg:                  # chained thin walls need a deeper halo than drawn: a refusal, not a failure
            return None
        return f"ERROR {desc}: {msg[:300]}"
    ref = O.make_simulator(g["bc_const"], mask, g["bc_dye"] if with_dye else None, scheme=scheme, dt=cfg["dt"], dx=cfg["dx"], re=cfg["re"],
                           vor_eps=vc, updater=updater, dtype=dtype)
    for _ in range(steps):
        ref.update()
    for k, (name, e) in enumerate(ref.fields().items()):
        full = np.concatenate([results[r][0][steps][k] for r in range(world)], axis=1)
        if not np.array_equal(full, e, equal_nan=True):
            bad = np.argwhere(~((full == e) | (np.isnan(full) & np.isnan(e))))
            return f"MISMATCH {desc} field {name}: {len(bad)} cells, first {bad[0].tolist()}, j range {bad[:, 1].min()}..{bad[:, 1].max()}"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, nargs="*")
    a = ap.parse_args()
    O.set_threads(8)
    if a.only:
        for sd in a.only:
            print(one_case(sd) or f"ok seed {sd}", flush=True)
        return
    t0, bad = time.time(), 0
    for k in range(a.cases):
        try:
            r = one_case(a.seed + k)
        except Exception as exc:     # noqa: BLE001
            r = f"ERROR seed {a.seed + k}: {type(exc).__name__}: {str(exc)[:300]}"
        if r:
            bad += 1
            print(r, flush=True)
    print(f"{a.cases} slab cases from seed {a.seed}: {bad} failing, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
