#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own kernel source.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):
the reference package `fs` is imported from /root/reference under the serial fake-taichi in
oracle/shim/ (Taichi itself is not installable here), its kernels / classes are executed on
seeded inputs and the inputs + outputs are written as small .npz fixtures.  Nothing of the
reference's source is stored - only arrays.

    python tests/golden/make_golden.py            # everything (about 10-15 min on 8 cores)
    python tests/golden/make_golden.py scenes kernels traj   # subsets

Fixture families
  scenes.npz / scene_hashes.json : outputs of create_boundary_condition1..6 (boundary_condition.py:222-524)
  kernels_bc{n}.npz              : single-call I/O of every @ti.kernel on the step() path, res 16
  traj_*.npz                     : FluidSimulator.step() trajectories (field_to_numpy + internal buffers)
  vis_*.npz                      : a state + the RGB buffers of get_norm/pressure/vorticity/dye_field() for it
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def _setup(fp64=False, oob="clamp"):
    import numpy as np

    np.seterr(all="ignore")
    sys.path.insert(0, os.path.join(REPO, "oracle", "shim"))
    sys.path.insert(0, REFERENCE)
    import taichi as ti

    ti.OOB_POLICY = oob
    if fp64:
        ti.set_default_fp(np.float64)
    return np, ti


def _sha(a):
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def _scene_arrays(bc):
    out = {"bc_const": bc._bc_const.arr.copy(), "bc_mask": bc._bc_mask.arr.copy()}
    if hasattr(bc, "_bc_dye"):
        out["bc_dye"] = bc._bc_dye.arr.copy()
    return out


# --------------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------------
def job_scenes(_):
    np, ti = _setup()
    import PIL
    from fs.boundary_condition import get_boundary_condition

    arrays, hashes = {}, {}
    for res in (16, 32, 48):
        for n in range(1, 7):
            s = _scene_arrays(get_boundary_condition(n, res, enable_dye=True))
            for k, a in s.items():
                arrays[f"bc{n}_res{res}_{k}"] = a
    for n, res in [(1, 200), (2, 400), (3, 200), (4, 400), (5, 800), (6, 400)]:
        s = _scene_arrays(get_boundary_condition(n, res, enable_dye=True))
        m = s["bc_mask"]
        hashes[f"bc{n}_res{res}"] = {
            "shape": list(m.shape),
            "counts": [int((m == c).sum()) for c in range(4)],
            "sha_mask": _sha(m), "sha_bc_const": _sha(s["bc_const"]), "sha_bc_dye": _sha(s["bc_dye"]),
        }
    np.savez_compressed(os.path.join(HERE, "scenes.npz"), **arrays)
    return {"pillow_version_used_for_bc6": PIL.__version__, "scenes": hashes}


def job_scene_hash_big(args):
    n, res = args
    np, ti = _setup()
    from fs.boundary_condition import get_boundary_condition

    s = _scene_arrays(get_boundary_condition(n, res, enable_dye=True))
    m = s["bc_mask"]
    return f"bc{n}_res{res}", {
        "shape": list(m.shape),
        "counts": [int((m == c).sum()) for c in range(4)],
        "sha_mask": _sha(m), "sha_bc_const": _sha(s["bc_const"]), "sha_bc_dye": _sha(s["bc_dye"]),
    }


# --------------------------------------------------------------------------------------------
# per-kernel vectors
# --------------------------------------------------------------------------------------------
def job_kernels(n):
    np, ti = _setup()
    import fs.solver as S
    from fs.advection import advect_kk_scheme, advect_upwind
    from fs.boundary_condition import get_boundary_condition
    from fs.pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater
    from fs.vorticity_confinement import VorticityConfinement

    res = 16
    dt, dx, re, w, omega = 0.05 / res, 1.0 / res, 1000.0, 5.0, 1.3
    bc = get_boundary_condition(n, res, enable_dye=True)
    X, Y = bc.get_resolution()
    rng = np.random.default_rng(1000 + n)
    out = {"params": np.array([res, dt, dx, re, w, omega], dtype=np.float64)}
    out.update(_scene_arrays(bc))

    def rnd(c, lo=-1.0, hi=1.0):
        shape = (X, Y) if c == 1 else (X, Y, c)
        return rng.uniform(lo, hi, shape).astype(np.float32)

    def F(a):  # numpy -> shim field
        f = ti.field(ti.f32, a.shape) if a.ndim == 2 else ti.Vector.field(a.shape[2], ti.f32, a.shape[:2])
        f.from_numpy(a)
        return f

    def rec(name, ins, fn, outs):
        fields = {k: F(a) for k, a in ins.items()}
        fn(fields)
        for k, a in ins.items():
            out[f"{name}.in.{k}"] = a
        for k in outs:
            out[f"{name}.out.{k}"] = fields[k].to_numpy()

    jac = JacobiPressureUpdater(bc, dt, dx, 3)
    sor = RedBlackSorPressureUpdater(bc, dt, dx, omega, 2)
    mac_up = S.DyeMacSolver(bc, sor, advect_upwind, dt, dx, re, None)
    mac_kk = S.DyeMacSolver(bc, sor, advect_kk_scheme, dt, dx, re, None)
    cip = S.DyeCipMacSolver(bc, sor, dt, dx, re, None)

    # K1 / K7 / K10 boundary-condition kernels (boundary_condition.py:16-65, 94-99)
    rec("velocity_bc", {"v": rnd(2)}, lambda f: bc.set_velocity_boundary_condition(f["v"]), ["v"])
    rec("pressure_bc", {"p": rnd(1, -10, 10)}, lambda f: bc.set_pressure_boundary_condition(f["p"]), ["p"])
    rec("dye_bc", {"dye": rnd(3, 0, 1)}, lambda f: bc.set_dye_boundary_condition(f["dye"]), ["dye"])

    # K2' MacSolver._update_velocities, K11 _update_dye (solver.py:94-107, 157-161)
    for tag, slv in (("upwind", mac_up), ("kk", mac_kk)):
        rec(f"mac_update_{tag}", {"vn": rnd(2), "vc": rnd(2), "pc": rnd(1, -10, 10)},
            lambda f, slv=slv: slv._update_velocities(f["vn"], f["vc"], f["pc"]), ["vn"])
        rec(f"mac_dye_{tag}", {"dn": rnd(3, 0, 1), "dc": rnd(3, 0, 1), "vc": rnd(2)},
            lambda f, slv=slv: slv._update_dye(f["dn"], f["dc"], f["vc"]), ["dn"])

    # CIP kernels (solver.py:207-332, 378-383)
    rec("cip_set_grad", {"fx": rnd(2), "fy": rnd(2), "f": rnd(2)},
        lambda f: cip._set_grad(f["fx"], f["fy"], f["f"]), ["fx", "fy"])
    rec("cip_nonadv", {"fn": rnd(2), "fc": rnd(2), "pc": rnd(1, -10, 10)},
        lambda f: cip._non_advection_phase(f["fn"], f["fc"], f["pc"]), ["fn"])
    rec("cip_nonadv_dye", {"dn": rnd(3, 0, 1), "dc": rnd(3, 0, 1)},
        lambda f: cip._non_advection_phase_dye(f["dn"], f["dc"]), ["dn"])
    for c in (2, 3):
        rec(f"cip_nonadv_grad_c{c}",
            {"fxn": rnd(c), "fyn": rnd(c), "fxc": rnd(c), "fyc": rnd(c), "fc": rnd(c), "fn": rnd(c)},
            lambda f: cip._non_advection_phase_grad(f["fxn"], f["fyn"], f["fxc"], f["fyc"], f["fc"], f["fn"]),
            ["fxn", "fyn"])
    rec("cip_advect_c2",
        {"fn": rnd(2), "fxn": rnd(2), "fyn": rnd(2), "fc": rnd(2), "fxc": rnd(2, -4, 4), "fyc": rnd(2, -4, 4)},
        lambda f: cip._advection_phase(f["fn"], f["fxn"], f["fyn"], f["fc"], f["fxc"], f["fyc"], f["fc"]),
        ["fn", "fxn", "fyn"])
    rec("cip_advect_c3",
        {"fn": rnd(3), "fxn": rnd(3), "fyn": rnd(3), "fc": rnd(3, 0, 1), "fxc": rnd(3, -4, 4),
         "fyc": rnd(3, -4, 4), "v": rnd(2)},
        lambda f: cip._advection_phase(f["fn"], f["fxn"], f["fyn"], f["fc"], f["fxc"], f["fyc"], f["v"]),
        ["fn", "fxn", "fyn"])

    # K5 / K6 vorticity confinement (vorticity_confinement.py:27-59); second case = H4 (all-zero v)
    for tag, vin in (("rand", rnd(2)), ("zero", np.zeros((X, Y, 2), np.float32))):
        vc_ = VorticityConfinement(bc, dt, dx, w)
        fields = {"vn": F(rnd(2)), "vc": F(vin)}
        out[f"vort_{tag}.in.vn"] = fields["vn"].to_numpy()
        out[f"vort_{tag}.in.vc"] = vin
        vc_._calc_vorticity(fields["vc"])
        out[f"vort_{tag}.out.vorticity"] = vc_.vorticity.to_numpy()
        out[f"vort_{tag}.out.vorticity_abs"] = vc_.vorticity_abs.to_numpy()
        vc_._add_vorticity(fields["vn"], fields["vc"])
        out[f"vort_{tag}.out.vn"] = fields["vn"].to_numpy()

    # K8J / K8R single sweeps (pressure_updater.py:62-66, 98-114)
    rec("jacobi_sweep", {"pn": rnd(1, -10, 10), "pc": rnd(1, -10, 10), "vc": rnd(2)},
        lambda f: jac._update(f["pn"], f["pc"], f["vc"]), ["pn"])
    rec("rbsor_odd", {"pn": rnd(1, -10, 10), "pc": rnd(1, -10, 10), "vc": rnd(2)},
        lambda f: sor._update_pressures_odd(f["pn"], f["pc"], f["vc"]), ["pn"])
    rec("rbsor_even", {"pn": rnd(1, -10, 10), "vc": rnd(2)},
        lambda f: sor._update_pressures_even(f["pn"], f["pn"], f["vc"]), ["pn"])

    # whole PressureUpdater.update incl. K7 and the buffer choreography (pressure_updater.py:56-60, 86-96)
    from fs.double_buffer import DoubleBuffer
    for tag, upd in (("jacobi3", jac), ("rbsor2", sor)):
        p = DoubleBuffer((X, Y), 1)
        a, b, v = rnd(1, -10, 10), rnd(1, -10, 10), rnd(2)
        p.current.from_numpy(a)
        p.next.from_numpy(b)
        upd.update(p, F(v))
        out[f"pressure_update_{tag}.in.p_current"] = a
        out[f"pressure_update_{tag}.in.p_next"] = b
        out[f"pressure_update_{tag}.in.v"] = v
        out[f"pressure_update_{tag}.out.p_current"] = p.current.to_numpy()
        out[f"pressure_update_{tag}.out.p_next"] = p.next.to_numpy()

    # K9 / K13 (solver.py:38-49)
    rec("limit_field", {"v": rnd(2, -15, 15)}, lambda f: S.limit_field(f["v"], S.VELOCITY_LIMIT), ["v"])
    rec("clamp_field", {"dye": rnd(3, -0.5, 1.5)}, lambda f: S.clamp_field(f["dye"], 0.0, 1.0), ["dye"])

    np.savez_compressed(os.path.join(HERE, f"kernels_bc{n}.npz"), **out)
    return f"kernels_bc{n}", {str(k): v for k, v in ti.OOB_LOG.items()}


# --------------------------------------------------------------------------------------------
# trajectories
# --------------------------------------------------------------------------------------------
def _traj_jobs():
    jobs = []

    def add(name, **kw):
        d = dict(name=name, bc=1, res=32, scheme="cip", vc=None, re=1.0e6, dt=None, updater=("rbsor", 1.3, 2),
                 dye=False, snaps=(1, 2, 5, 10), fp64=False)
        d.update(kw)
        jobs.append(d)

    for bc in (1, 2, 3, 4, 5):
        for scheme in ("upwind", "kk", "cip"):
            for vc in (None, 5.0):
                add(f"bc{bc}_{scheme}_vc{0 if vc is None else int(vc)}", bc=bc, scheme=scheme, vc=vc)
    # BASELINE config 1 parameters (README.md:34): Re 1000, dt 5e-4, VC off
    add("cfg1_bc1_upwind_re1000", bc=1, scheme="upwind", re=1000.0, dt=0.0005, snaps=(1, 2, 5, 10, 20))
    # Jacobi composed by hand (BASELINE config 2 style; pressure_updater.py:41-66)
    add("bc2_cip_jacobi4_vc0", bc=2, updater=("jacobi", 4), snaps=(1, 2, 5))
    add("bc2_cip_jacobi4_vc5", bc=2, vc=5.0, updater=("jacobi", 4), snaps=(1, 2, 5))
    add("bc2_cip_jacobi50_vc0", bc=2, updater=("jacobi", 50), snaps=(1, 2, 3))
    add("bc1_upwind_jacobi4_vc0", bc=1, scheme="upwind", updater=("jacobi", 4), snaps=(1, 2, 5))
    add("bc5_kk_jacobi4_vc5", bc=5, scheme="kk", vc=5.0, updater=("jacobi", 4), snaps=(1, 2, 5))
    # dye transport (DyeFluidSimulator.create, fluid_simulator.py:129-176)
    for bc in (1, 2, 5):
        for scheme in ("upwind", "kk", "cip"):
            add(f"dye_bc{bc}_{scheme}_vc5", bc=bc, scheme=scheme, vc=5.0, dye=True, snaps=(1, 2, 5))
    # BASELINE config 5 parameters at a resolution where bc3's circles are >= 3 cells in radius
    add("cfg5_bc3_res96_kk_vc10_re1e8", bc=3, res=96, scheme="kk", vc=10.0, re=1.0e8, snaps=(1, 2, 4))
    add("bc6_cip_vc5", bc=6, vc=5.0, snaps=(1, 2, 5))
    add("bc6_res64_cip_vc5_dye", bc=6, res=64, vc=5.0, dye=True, snaps=(1, 3))
    # f64 'truth' (the reference has no f64 mode; the build's f64 instantiation is pinned by these)
    add("f64_bc1_cip_vc0", bc=1, fp64=True, snaps=(1, 5, 10))
    add("f64_bc3_kk_vc0_re1e8", bc=3, res=48, scheme="kk", re=1.0e8, fp64=True, snaps=(1, 5, 10))
    add("f32_bc3_res48_kk_vc0_re1e8", bc=3, res=48, scheme="kk", re=1.0e8, snaps=(1, 5, 10))
    return jobs


def job_traj(job):
    np, ti = _setup(fp64=job["fp64"])
    from fs.advection import advect_kk_scheme, advect_upwind
    from fs.boundary_condition import get_boundary_condition
    from fs.fluid_simulator import DyeFluidSimulator, FluidSimulator
    from fs.pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater
    from fs.solver import CipMacSolver, DyeCipMacSolver, DyeMacSolver, MacSolver
    from fs.vorticity_confinement import VorticityConfinement

    t0 = time.time()
    res, scheme, dye = job["res"], job["scheme"], job["dye"]
    dt = job["dt"] if job["dt"] is not None else 0.05 / res
    dx = 1 / res
    bc = get_boundary_condition(job["bc"], res, enable_dye=dye)
    vc = VorticityConfinement(bc, dt, dx, job["vc"]) if job["vc"] is not None else None
    u = job["updater"]
    pu = (RedBlackSorPressureUpdater(bc, dt, dx, u[1], u[2]) if u[0] == "rbsor"
          else JacobiPressureUpdater(bc, dt, dx, u[1]))
    if scheme == "cip":
        solver = (DyeCipMacSolver if dye else CipMacSolver)(bc, pu, dt, dx, job["re"], vc)
    else:
        adv = advect_upwind if scheme == "upwind" else advect_kk_scheme
        solver = (DyeMacSolver if dye else MacSolver)(bc, pu, adv, dt, dx, job["re"], vc)
    sim = (DyeFluidSimulator if dye else FluidSimulator)(solver)

    out = {"params": np.array([job["bc"], res, dt, dx, job["re"], -1.0 if job["vc"] is None else job["vc"]],
                              dtype=np.float64),
           "scheme": np.array(scheme), "updater": np.array([str(x) for x in u]),
           "dye": np.array(dye), "fp64": np.array(job["fp64"]), "snaps": np.array(job["snaps"])}
    out.update(_scene_arrays(bc))
    for step in range(1, max(job["snaps"]) + 1):
        sim.step()
        if step in job["snaps"]:
            for k, a in sim.field_to_numpy().items():
                out[f"step{step}.{k}"] = a
    # internal buffers after the last step (pins the H5 stale-cell choreography)
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
        if hasattr(solver, name):
            db = getattr(solver, name)
            out[f"final.{name}.current"] = db.current.to_numpy()
            out[f"final.{name}.next"] = db.next.to_numpy()
    if vc is not None:
        out["final.vorticity"] = vc.vorticity.to_numpy()
        out["final.vorticity_abs"] = vc.vorticity_abs.to_numpy()
    np.savez_compressed(os.path.join(HERE, f"traj_{job['name']}.npz"), **out)
    return job["name"], round(time.time() - t0, 1)


# --------------------------------------------------------------------------------------------
# visualisation buffers (fluid_simulator.py:22-58, 112-126)
# --------------------------------------------------------------------------------------------
def job_vis(job):
    """State after a few steps + the four image buffers the reference's getters produce from it."""
    np, ti = _setup()
    from fs.fluid_simulator import DyeFluidSimulator

    bcn, res, scheme, steps = job
    dt, dx = 0.05 / res, 1 / res
    sim = DyeFluidSimulator.create(bcn, res, dt, dx, 1.0e6, 5.0, scheme)
    for _ in range(steps):
        sim.step()
    out = {"params": np.array([bcn, res, dt, dx, 1.0e6, 5.0], dtype=np.float64), "scheme": np.array(scheme), "steps": np.array(steps)}
    out.update(_scene_arrays(sim._solver._bc))
    for k, a in sim.field_to_numpy().items():
        out[f"state.{k}"] = a
    out["rgb.norm"] = sim.get_norm_field().to_numpy().copy()
    out["rgb.pressure"] = sim.get_pressure_field().to_numpy().copy()
    out["rgb.vorticity"] = sim.get_vorticity_field().to_numpy().copy()
    out["rgb.dye"] = sim.get_dye_field().to_numpy().copy()
    name = f"vis_bc{bcn}_res{res}_{scheme}"
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return name


def main():
    what = set(sys.argv[1:]) or {"scenes", "kernels", "traj", "bighash", "vis"}
    ctx = mp.get_context("spawn")
    meta_path = os.path.join(HERE, "scene_hashes.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    with ctx.Pool(8, maxtasksperchild=1) as pool:
        pending = []
        if "scenes" in what:
            pending.append(("scenes", pool.apply_async(job_scenes, (0,))))
        if "bighash" in what:
            # BASELINE.json sizes; bc2@8192 needs ~6 GB and ~1 min
            for n, res in [(2, 1600), (5, 4096), (3, 4096), (2, 8192)]:
                pending.append(("big", pool.apply_async(job_scene_hash_big, ((n, res),))))
        if "kernels" in what:
            for n in range(1, 7):
                pending.append(("kernels", pool.apply_async(job_kernels, (n,))))
        if "vis" in what:
            for job in [(1, 32, "cip", 4), (2, 32, "kk", 3), (5, 32, "cip", 5), (3, 48, "upwind", 3)]:
                pending.append(("vis", pool.apply_async(job_vis, (job,))))
        if "traj" in what:
            for job in sorted(_traj_jobs(), key=lambda j: -j["res"] * (3 if j["scheme"] == "cip" else 1)):
                pending.append(("traj", pool.apply_async(job_traj, (job,))))
        for kind, r in pending:
            res = r.get()
            if kind == "scenes":
                meta.update({k: v for k, v in res.items() if k != "scenes"})
                meta.setdefault("scenes", {}).update(res["scenes"])
            elif kind == "big":
                meta.setdefault("scenes", {})[res[0]] = res[1]
            elif kind == "kernels":
                meta.setdefault("oob_log", {})[res[0]] = res[1]
            print(kind, res if kind != "scenes" else "ok", flush=True)
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
