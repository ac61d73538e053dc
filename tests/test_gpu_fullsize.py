"""GPU parity at BASELINE.json's full sizes, through size-independent properties:

  * the gfx950 fast paths (quad/tile kernels, fused vorticity confinement, fused red-black iteration, precomputed
    Poisson source, exact-reciprocal shortcut) must reproduce, BIT FOR BIT, the one-cell-per-lane reference-literal
    kernels (FS_MARCH=0) that the small-size tests pin against the oracle and the golden vectors;
  * a run is deterministic (same bits twice);
  * at sizes the CPU oracle finishes in seconds (res 200 / 512) the comparison is against the oracle itself;
  * f64 vs f32 (BASELINE config 5's tolerance sweep): <= 1e-5 rel-L2 with vorticity confinement off; with it on the
    curve is reported, not asserted (hazard H4 makes the force discontinuous - SURVEY.md Appendix C).
"""
import os

import numpy as np
import pytest
from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _build(bc, res, scheme, vc, re, updater, march, dtype="f32", dye=False, **kw):
    import fs
    old = os.environ.get("FS_MARCH")
    os.environ["FS_MARCH"] = "1" if march else "0"
    try:
        fs.runtime.init(gpu=0, dtype=dtype)
        cls = fs.DyeFluidSimulator if dye else fs.FluidSimulator
        return cls.create(bc, res, kw.get("dt", 0.05 / res), 1.0 / res, re, vc, scheme, pressure_updater=updater)
    finally:
        if old is None:
            os.environ.pop("FS_MARCH", None)
        else:
            os.environ["FS_MARCH"] = old


def _state(sim):
    s = sim._solver
    out = {}
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
        if hasattr(s, name):
            out[name + ".current"] = getattr(s, name).current.to_numpy()
            out[name + ".next"] = getattr(s, name).next.to_numpy()
    return out


CONFIGS = [
    # BASELINE.json configs[2]: bc5 res 4096 CIP + VC, RB-SOR(1.3, 2)
    pytest.param(dict(bc=5, res=4096, scheme="cip", vc=5.0, re=1e6, updater=None), 6, id="cfg3-bc5-res4096-cip-vc"),
    # configs[3]: bc2 res 8192 CIP (+ the CLI's default VC 5), 16384 x 8192 cells - the grid the 8-GPU slab run cuts; fits ONE GPU (10 GB)
    pytest.param(dict(bc=2, res=8192, scheme="cip", vc=5.0, re=1e6, updater=None), 3, id="cfg4-bc2-res8192-cip-vc"),
    # configs[4]: bc3 res 4096 KK + VC 10, Re 1e8
    pytest.param(dict(bc=3, res=4096, scheme="kk", vc=10.0, re=1e8, updater=None), 6, id="cfg5-bc3-res4096-kk-vc10"),
    # configs[1]: bc2 res 1600 CIP, 50 Jacobi sweeps per step (res 1600: dx is not a power of two -> division path)
    pytest.param(dict(bc=2, res=1600, scheme="cip", vc=5.0, re=1e6, updater=("jacobi", 50)), 4, id="cfg2-bc2-res1600-cip-jacobi50"),
    # dye transport at size (next-row component)
    pytest.param(dict(bc=5, res=2048, scheme="cip", vc=5.0, re=1e6, updater=None, dye=True), 4, id="dye-bc5-res2048-cip"),
    # the f64 instantiations of the fast paths (double4 quads, shuffle instead of DPP)
    pytest.param(dict(bc=3, res=1024, scheme="kk", vc=10.0, re=1e8, updater=None, dtype="f64"), 4, id="f64-bc3-res1024-kk-vc10"),
    pytest.param(dict(bc=5, res=1000, scheme="cip", vc=5.0, re=1e6, updater=("jacobi", 10), dtype="f64", dye=True), 3, id="f64-dye-bc5-res1000-cip-jacobi10"),
]


@pytest.mark.parametrize("cfg,steps", CONFIGS)
def test_fast_paths_equal_reference_literal_kernels(cfg, steps, hip_lib):
    fast = _build(march=True, **cfg)
    ref = _build(march=False, **cfg)
    try:
        for _ in range(steps):
            fast.step()
            ref.step()
        from helpers import dead_buffers, fluid_dead_buffers
        pmax = 0.0
        for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):          # one buffer at a time: res 8192 fields are 1 GB each
            for which in ("current", "next"):
                k = f"{name}.{which}"
                if not hasattr(fast._solver, name) or k in dead_buffers(fast._solver):
                    continue
                a = getattr(getattr(fast._solver, name), which).to_numpy()
                b = getattr(getattr(ref._solver, name), which).to_numpy()
                if k in fluid_dead_buffers(fast._solver):          # (the all-fluid tiles keep the intermediate field in registers: helpers.py)
                    keep = fast._solver._bc.mask != 0
                    a, b = a[keep], b[keep]
                assert np.array_equal(a, b, equal_nan=True), f"{k}: rel-L2 {rel_l2(a, b):.3e}"
                if k == "p.current":
                    pmax = float(np.abs(a).max())
        assert pmax > 0
    finally:
        fast._solver._bc.device.close()
        ref._solver._bc.device.close()


def test_run_is_deterministic_res4096(hip_lib):
    cfg = dict(bc=5, res=4096, scheme="cip", vc=5.0, re=1e6, updater=None)
    outs = []
    for _ in range(2):
        sim = _build(march=True, **cfg)
        for _ in range(5):
            sim.step()
        outs.append(sim.field_to_numpy())
        sim._solver._bc.device.close()
    assert np.array_equal(outs[0]["v"], outs[1]["v"]) and np.array_equal(outs[0]["p"], outs[1]["p"])


@pytest.mark.parametrize("bc,res,scheme,vc,re,dt,steps", [
    (1, 200, "upwind", None, 1000.0, 0.0005, 60),     # BASELINE configs[0]
    (5, 512, "cip", 5.0, 1e6, None, 12),
    (3, 384, "kk", 10.0, 1e8, None, 12),
])
def test_against_oracle_at_moderate_size(bc, res, scheme, vc, re, dt, steps, hip_lib):
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    dt = dt if dt is not None else 0.05 / res
    sim = _build(bc, res, scheme, vc, re, None, True, dt=dt)
    try:
        const, mask, _ = create_scene_arrays(bc, res)
        ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=1.0 / res, re=re, vor_eps=vc)
        for _ in range(steps):
            sim.step()
            ref.update()
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e), f"{k}: rel-L2 {rel_l2(out[k], e):.3e}"
    finally:
        sim._solver._bc.device.close()


@pytest.mark.parametrize("bc,res,scheme,vc,dye,steps", [
    (5, 256, "cip", 5.0, False, 600),      # the flow is fully developed by then: every sign / NaN branch of CIP and VC is taken
    (2, 200, "cip", 5.0, True, 300),       # the reference's default mode (dye on), dx not a power of two
    (3, 192, "kk", 10.0, False, 400),
    (5, 512, "cip", 5.0, False, 1500),
])
def test_long_run_stays_bit_identical(bc, res, scheme, vc, dye, steps, hip_lib):
    """Hundreds of steps: with vorticity confinement a one-ulp difference anywhere grows to O(1e-3) within a few steps
    (hazard H4), so equality at the end of a long run is a much stronger statement than at step 20."""
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    dt, dx = 0.05 / res, 1.0 / res
    sim = _build(bc, res, scheme, vc, 1e6, None, True, dye=dye)
    try:
        const, mask, bdye = create_scene_arrays(bc, res)
        ref = O.make_simulator(const, mask, bdye if dye else None, scheme=scheme, dt=dt, dx=dx, re=1e6, vor_eps=vc)
        for step in range(1, steps + 1):
            sim.step()
            ref.update()
            if step % 100 == 0 or step == steps:
                out = sim.field_to_numpy()
                for k, e in ref.fields().items():
                    assert np.array_equal(out[k], e, equal_nan=True), f"step {step} {k}: rel-L2 {rel_l2(out[k], e):.3e}"
        assert np.isfinite(out["v"]).all() and float(np.abs(out["v"]).max()) > 0.5
    finally:
        sim._solver._bc.device.close()


def test_against_oracle_at_baseline_size(hip_lib):
    """BASELINE configs[2] itself (bc5, res 4096, CIP + VC): two steps on the GPU against two steps of the CPU oracle, bit for bit."""
    import os
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    res = 4096
    dt, dx = 0.05 / res, 1.0 / res
    sim = _build(5, res, "cip", 5.0, 1e6, None, True)
    O.set_threads(min(64, len(os.sched_getaffinity(0))))
    try:
        const, mask, _ = create_scene_arrays(5, res)
        ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=1e6, vor_eps=5.0)
        for _ in range(2):
            sim.step()
            ref.update()
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e), f"{k}: rel-L2 {rel_l2(out[k], e):.3e}"
        assert float(np.abs(out["p"]).max()) > 0
    finally:
        O.set_threads(8)
        sim._solver._bc.device.close()


def test_f64_vs_f32_tolerance_sweep(hip_lib, capsys):
    """BASELINE configs[4] at its own size: bc3, res 4096 (8192 x 4096 cells), Kawamura-Kuwahara, Re 1e8, f32 against the build's
    f64 instantiation (the reference is f32 only) after 1, 2, 5, 10, 20 steps.  Vorticity confinement off: asserted <= 1e-5
    rel-L2 on v and p; VC = 10 (the configuration's value): reported, not asserted - the confinement force is discontinuous
    (hazard H4), a one-ulp difference flips whole cells.  The curve is printed and written to gpurun_out/ (copied to profiles/)."""
    import json
    res, snaps = 4096, (1, 2, 5, 10, 20)
    curves = {}
    for vc in (None, 10.0):
        sims = {d: _build(3, res, "kk", vc, 1e8, None, True, dtype=d) for d in ("f32", "f64")}
        curve = []
        for step in range(1, max(snaps) + 1):
            for s in sims.values():
                s.step()
            if step in snaps:
                ev = ep = None
                for name, key in (("v", 0), ("p", 1)):
                    a = sims["f32"]._solver.get_fields()[key].to_numpy()
                    b = sims["f64"]._solver.get_fields()[key].to_numpy()
                    e = rel_l2(a, b)
                    del a, b
                    if name == "v":
                        ev = e
                    else:
                        ep = e
                curve.append((step, ev, ep))
        for s in sims.values():
            s._solver._bc.device.close()
        curves[vc] = curve
    report = {"config": "BASELINE.json configs[4]: bc=3 res=4096 (8192x4096 cells) scheme=kk Re=1e8 dt=0.05/res RB-SOR(1.3, 2)",
              "metric": "rel-L2 of the f32 run against the f64 run (same kernels, double instantiation)",
              "curves": {("vc_off" if vc is None else f"vc_{vc:g}"): [{"step": s_, "v": ev, "p": ep} for s_, ev, ep in c]
                         for vc, c in curves.items()},
              "asserted": "vc_off: v, p <= 1e-5 at every snapshot; vc_10: reported only (discontinuous confinement force, SURVEY.md H4)"}
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "f64_vs_f32_bc3_res4096.json"), "w") as f:
            json.dump(report, f, indent=1)
    with capsys.disabled():
        for vc, curve in curves.items():
            print(f"\n[f64-vs-f32 bc3 res{res} kk Re1e8 vc={vc}] " + "  ".join(f"step{s}: v {ev:.2e} p {ep:.2e}" for s, ev, ep in curve))
    assert all(ev <= 1e-5 and ep <= 1e-5 for _, ev, ep in curves[None]), curves[None]
    assert all(np.isfinite(ev) and np.isfinite(ep) for _, ev, ep in curves[10.0])
