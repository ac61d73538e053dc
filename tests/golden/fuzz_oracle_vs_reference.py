#!/usr/bin/env python3
"""Differential fuzzing of the CPU oracle against the REFERENCE'S OWN KERNEL SOURCE (build container only: needs /root/reference).

Random small scenes that no reference scene resembles - salt-and-pepper walls, one-cell-thin and diagonal walls (chained boundary
hazards, mirrors that overwrite fluid cells), inflow / outflow cells anywhere, fluid touching the domain edge - are run through the
reference's classes under the serial fake-taichi (oracle/shim) and through oracle/libfs_oracle.so; every field must agree bit for bit.
With --save N the first N cases are written as tests/golden/traj_fuzz_<k>.npz (same layout as the other trajectories), so that the GPU
suite replays them as well.   python tests/golden/fuzz_oracle_vs_reference.py [--cases 200] [--seed 0] [--save 0] [--fp64]"""
import argparse
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def case(args):
    seed, fp64, save = args
    import numpy as np
    np.seterr(all="ignore")
    sys.path.insert(0, os.path.join(REPO, "oracle", "shim"))
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import taichi as ti
    ti.OOB_POLICY = "clamp"
    if fp64:
        ti.set_default_fp(np.float64)
    from fs.advection import advect_kk_scheme, advect_upwind
    from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition
    from fs.pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater
    from fs.solver import CipMacSolver, DyeCipMacSolver, DyeMacSolver, MacSolver
    from fs.vorticity_confinement import VorticityConfinement
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_scene", os.path.join(REPO, "tools", "fuzz_parity.py"))
    src = open(os.path.join(REPO, "tools", "fuzz_parity.py")).read()
    ns = {}
    exec(src[src.index("def random_scene"):src.index("def one_case")], {"np": np}, ns)     # only the scene generator (no HIP import)
    random_scene = ns["random_scene"]
    from oracle import oracle as O

    rng = np.random.default_rng(seed)
    X, Y = int(rng.integers(6, 40)), int(rng.integers(6, 28))
    dtype = np.float64 if fp64 else np.float32
    const, mask, dye = random_scene(rng, X, Y)
    const, dye = const.astype(dtype), dye.astype(dtype)
    scheme = str(rng.choice(["cip", "kk", "upwind"]))
    with_dye = bool(rng.random() < 0.4)
    vc = None if rng.random() < 0.3 else float(rng.choice([0.5, 5.0, 10.0]))
    updater = ("rbsor", float(rng.choice([1.0, 1.3, 1.9])), int(rng.integers(1, 4))) if rng.random() < 0.6 else ("jacobi", int(rng.choice([1, 3, 6])))
    res = float(rng.choice([16, 32, 25, 100]))
    dt, dx, re = 0.05 / res, 1.0 / res, float(rng.choice([100.0, 1e6]))
    steps = 3
    bc = DyeBoundaryCondition(const, dye, mask) if with_dye else BoundaryCondition(const, mask)
    vcobj = VorticityConfinement(bc, dt, dx, vc) if vc is not None else None
    pu = RedBlackSorPressureUpdater(bc, dt, dx, updater[1], updater[2]) if updater[0] == "rbsor" else JacobiPressureUpdater(bc, dt, dx, updater[1])
    if scheme == "cip":
        solver = (DyeCipMacSolver if with_dye else CipMacSolver)(bc, pu, dt, dx, re, vcobj)
    else:
        adv = advect_upwind if scheme == "upwind" else advect_kk_scheme
        solver = (DyeMacSolver if with_dye else MacSolver)(bc, pu, adv, dt, dx, re, vcobj)
    ref = O.make_simulator(const, mask, dye if with_dye else None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=updater, dtype=dtype)
    amp = float(rng.choice([1e-3, 1.0, 20.0]))
    v0 = (rng.uniform(-1, 1, (X, Y, 2)) * amp).astype(dtype)
    p0 = rng.uniform(-1, 1, (X, Y)).astype(dtype)
    solver.v.current.from_numpy(v0); ref.v.current[...] = v0
    solver.p.current.from_numpy(p0); ref.p.current[...] = p0
    desc = f"seed {seed}: {X}x{Y} {np.dtype(dtype).name} {scheme} vc={vc} {updater} dye={with_dye} res={res:g} re={re:g} amp={amp:g}"
    out = {"params": np.array([0, res, dt, dx, re, -1.0 if vc is None else vc], dtype=np.float64), "scheme": np.array(scheme),
           "updater": np.array([str(x) for x in updater]), "dye": np.array(with_dye), "fp64": np.array(fp64), "snaps": np.array([1, 2, 3]),
           "bc_const": const, "bc_mask": mask, "bc_dye": dye, "init.v": v0, "init.p": p0}
    for step in range(1, steps + 1):
        solver.update()
        ref.update()
        got = dict(zip(("v", "p", "dye"), [f.to_numpy() for f in solver.get_fields()]))
        for k, e in ref.fields().items():
            a = got[k]
            out[f"step{step}.{k}"] = a
            if not np.array_equal(a, e, equal_nan=True):
                bad = np.argwhere(~((a == e) | (np.isnan(a) & np.isnan(e))))
                return f"MISMATCH {desc} step {step} field {k}: {len(bad)} cells, first {bad[0].tolist()}"
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):          # internal buffers too (stale-cell choreography)
        if hasattr(solver, name):
            for which in ("current", "next"):
                a = getattr(getattr(solver, name), which).to_numpy()
                e = getattr(getattr(ref, name), which)
                out[f"final.{name}.{which}"] = a
                if not np.array_equal(a, e, equal_nan=True):
                    return f"MISMATCH {desc} internal buffer {name}.{which}"
    if vcobj is not None:
        for name in ("vorticity", "vorticity_abs"):
            a = getattr(vcobj, name).to_numpy()
            out[f"final.{name}"] = a
            if not np.array_equal(a, getattr(ref.vc, name), equal_nan=True):
                return f"MISMATCH {desc} {name}"
    if save:
        np.savez_compressed(os.path.join(HERE, f"traj_fuzz_{seed}.npz"), **out)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", type=int, default=0)
    ap.add_argument("--fp64", action="store_true")
    a = ap.parse_args()
    jobs = [(a.seed + k, a.fp64, k < a.save) for k in range(a.cases)]
    bad = 0
    with mp.get_context("spawn").Pool(8) as pool:
        for r in pool.imap_unordered(case, jobs, chunksize=4):
            if r:
                bad += 1
                print(r, flush=True)
    print(f"{a.cases} cases from seed {a.seed} ({'f64' if a.fp64 else 'f32'}): {bad} failing", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
