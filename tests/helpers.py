"""Shared helpers for the parity tests: build oracle / product simulators from a golden trajectory file."""
import numpy as np


def traj_config(g):
    bcn, res, dt, dx, re, vc = [float(x) for x in g["params"]]
    upd = [str(x) for x in g["updater"]]
    updater = ("rbsor", float(upd[1]), int(upd[2])) if upd[0] == "rbsor" else ("jacobi", int(upd[1]))
    return dict(bc=int(bcn), res=int(res), dt=dt, dx=dx, re=re, vor_eps=None if vc < 0 else vc,
                scheme=str(g["scheme"]), updater=updater, dye=bool(g["dye"]), fp64=bool(g["fp64"]),
                snaps=[int(s) for s in g["snaps"]])


def make_oracle(g, cfg=None):
    from oracle import oracle as O
    cfg = cfg or traj_config(g)
    sim = O.make_simulator(g["bc_const"], g["bc_mask"], g["bc_dye"] if cfg["dye"] else None, scheme=cfg["scheme"],
                           dt=cfg["dt"], dx=cfg["dx"], re=cfg["re"], vor_eps=cfg["vor_eps"], updater=cfg["updater"],
                           dtype=np.float64 if cfg["fp64"] else np.float32)
    if "init.v" in g:        # fuzz trajectories (tests/golden/fuzz_oracle_vs_reference.py) start from a random state
        sim.v.current[...] = g["init.v"]
        sim.p.current[...] = g["init.p"]
    return sim


def make_product(g, cfg=None, precompute_source=None, vc_kwargs=None, rb_fused=True, fused_transport=None, fused_clamp=None, rb_pair=None):
    """Compose the product classes by hand from the scene arrays stored in the fixture (constructor-level API)."""
    import fs
    from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition
    cfg = cfg or traj_config(g)
    dt, dx, re = cfg["dt"], cfg["dx"], cfg["re"]
    bc = (DyeBoundaryCondition(g["bc_const"], g["bc_dye"], g["bc_mask"]) if cfg["dye"]
          else BoundaryCondition(g["bc_const"], g["bc_mask"]))
    vc = fs.VorticityConfinement(bc, dt, dx, cfg["vor_eps"], **(vc_kwargs or {})) if cfg["vor_eps"] is not None else None
    u = cfg["updater"]
    pu = (fs.RedBlackSorPressureUpdater(bc, dt, dx, u[1], u[2], precompute_source=bool(precompute_source), fused=rb_fused, pair=rb_pair) if u[0] == "rbsor"
          else fs.JacobiPressureUpdater(bc, dt, dx, u[1], precompute_source=precompute_source))
    if cfg["scheme"] == "cip":
        solver = (fs.DyeCipMacSolver if cfg["dye"] else fs.CipMacSolver)(bc, pu, dt, dx, re, vc, fused_transport=fused_transport)
        if fused_clamp is not None and cfg["dye"]:
            solver._fused_clamp = bool(fused_clamp) and solver.resolution[0] % 4 == 0
    else:
        adv = fs.advect_upwind if cfg["scheme"] == "upwind" else fs.advect_kk_scheme
        solver = (fs.DyeMacSolver if cfg["dye"] else fs.MacSolver)(bc, pu, adv, dt, dx, re, vc)
    if "init.v" in g:
        solver.v.current.from_numpy(g["init.v"])
        solver.p.current.from_numpy(g["init.p"])
    return (fs.DyeFluidSimulator if cfg["dye"] else fs.FluidSimulator)(solver)


def dead_buffers(solver):
    """Internal buffers whose content is unobservable in the run mode of `solver` and therefore not reproduced:
    with the fused gradient+advection pass the intermediate gradients (vx.next / vy.next) are never stored."""
    dead = {"vx.next", "vy.next"} if getattr(solver, "_fused_transport", False) else set()
    if getattr(solver, "_fused_dye", False):
        dead |= {"dyex.next", "dyey.next"}
    return dead


def fluid_dead_buffers(solver):
    """Internal buffers whose FLUID cells are unobservable (and not reproduced) while their other cells are: where fs_cip_step / fs_cip_step_dye run
    in their three-part form (large single-GPU f32 grids; csrc/fs_k234.h) the post-K2 / post-K12 field of the all-fluid tiles stays in registers.
    v.next is that buffer only without vorticity confinement (with it, v.next is the advected velocity, fully written)."""
    dev = solver._bc.device
    if not (getattr(solver, "_fused_k2", False) and getattr(dev, "cip_step_fused", False)):
        return set()
    out = set()
    if getattr(solver, "vorticity_confinement", None) is None:
        out.add("v.next")
    if getattr(solver, "_fused_dye", False):
        out.add("dye.next")
    return out
