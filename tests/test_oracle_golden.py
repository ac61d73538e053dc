"""Pin the CPU oracle (oracle/fs_oracle*.{c,h}) against the golden vectors captured from the reference's own
kernel source (tests/golden/make_golden.py).  Bit-exact: the oracle evaluates the same expression trees in
IEEE f32/f64 as the reference source does under the serial shim."""
import glob
import os

import numpy as np
import pytest
from conftest import GOLDEN, golden
from helpers import make_oracle, traj_config
from oracle import oracle as O


def _eq(got, exp, what):
    assert got.dtype == exp.dtype and got.shape == exp.shape, what
    assert np.array_equal(got, exp, equal_nan=True), f"{what}: max|d| = {np.nanmax(np.abs(got - exp))}"


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_every_kernel_against_golden(n):
    g = golden(f"kernels_bc{n}.npz")
    res, dt, dx, re, w, omega = [float(x) for x in g["params"]]
    bc = O.OracleBC(g["bc_const"], g["bc_mask"], g["bc_dye"])
    X, Y, f32 = bc.X, bc.Y, np.float32

    def I(name, key):
        return g[f"{name}.in.{key}"].copy()

    def E(name, key, got):
        _eq(got, g[f"{name}.out.{key}"], f"bc{n} {name}.{key}")

    v = I("velocity_bc", "v"); bc.set_velocity_boundary_condition(v); E("velocity_bc", "v", v)
    p = I("pressure_bc", "p"); bc.set_pressure_boundary_condition(p); E("pressure_bc", "p", p)
    d = I("dye_bc", "dye"); bc.set_dye_boundary_condition(d); E("dye_bc", "dye", d)
    for tag, s in (("upwind", 0), ("kk", 1)):
        k = f"mac_update_{tag}"; vn = I(k, "vn")
        O._call("oracle_mac_update", f32, X, Y, dt, dx, re, s, bc.mask, vn, I(k, "vc"), I(k, "pc")); E(k, "vn", vn)
        k = f"mac_dye_{tag}"; dn = I(k, "dn")
        O._call("oracle_mac_dye", f32, X, Y, dt, dx, re, s, bc.mask, dn, I(k, "dc"), I(k, "vc")); E(k, "dn", dn)
    k = "cip_set_grad"; fx, fy = I(k, "fx"), I(k, "fy")
    O._call("oracle_cip_set_grad", f32, X, Y, dx, 2, fx, fy, I(k, "f")); E(k, "fx", fx); E(k, "fy", fy)
    k = "cip_nonadv"; fn = I(k, "fn")
    O._call("oracle_cip_nonadv", f32, X, Y, dt, dx, re, bc.mask, fn, I(k, "fc"), I(k, "pc")); E(k, "fn", fn)
    k = "cip_nonadv_dye"; dn = I(k, "dn")
    O._call("oracle_cip_nonadv_dye", f32, X, Y, dt, dx, re, bc.mask, dn, I(k, "dc")); E(k, "dn", dn)
    for c in (2, 3):
        k = f"cip_nonadv_grad_c{c}"; fxn, fyn = I(k, "fxn"), I(k, "fyn")
        O._call("oracle_cip_nonadv_grad", f32, X, Y, dx, c, bc.mask, fxn, fyn, I(k, "fxc"), I(k, "fyc"), I(k, "fc"), I(k, "fn"))
        E(k, "fxn", fxn); E(k, "fyn", fyn)
        k = f"cip_advect_c{c}"; fn, fxn, fyn, fc = I(k, "fn"), I(k, "fxn"), I(k, "fyn"), I(k, "fc")
        adv = fc if c == 2 else I(k, "v")
        O._call("oracle_cip_advect", f32, X, Y, dt, dx, c, bc.mask, fn, fxn, fyn, fc, I(k, "fxc"), I(k, "fyc"), adv)
        E(k, "fn", fn); E(k, "fxn", fxn); E(k, "fyn", fyn)
    for tag in ("rand", "zero"):
        k = f"vort_{tag}"; vc = O.OracleVorticity(bc, dt, dx, w); vn, vcur = I(k, "vn"), I(k, "vc")
        vc.calc(vcur); E(k, "vorticity", vc.vorticity); E(k, "vorticity_abs", vc.vorticity_abs)
        vc.add(vn, vcur); E(k, "vn", vn)
    jac, sor = O.OracleJacobi(bc, dt, dx, 3), O.OracleRedBlackSor(bc, dt, dx, omega, 2)
    k = "jacobi_sweep"; pn = I(k, "pn"); jac.sweep(pn, I(k, "pc"), I(k, "vc")); E(k, "pn", pn)
    k = "rbsor_odd"; pn = I(k, "pn"); sor.half(1, pn, I(k, "pc"), I(k, "vc")); E(k, "pn", pn)
    k = "rbsor_even"; pn = I(k, "pn"); sor.half(0, pn, pn, I(k, "vc")); E(k, "pn", pn)
    for tag, u in (("jacobi3", jac), ("rbsor2", sor)):
        k = f"pressure_update_{tag}"; pb = O.Buf2((X, Y), 1, f32)
        pb.current, pb.next = I(k, "p_current"), I(k, "p_next")
        u.update(pb, I(k, "v")); E(k, "p_current", pb.current); E(k, "p_next", pb.next)
    k = "limit_field"; v = I(k, "v"); O.limit_field(v); E(k, "v", v)
    k = "clamp_field"; d = I(k, "dye"); O.clamp_field(d, 0.0, 1.0); E(k, "dye", d)


def test_h4_zero_velocity_gives_plus_point_one():
    """Known-answer: all-zero v -> v.next = dt*w*0.1 on both components of every fluid cell."""
    g = golden("kernels_bc1.npz")
    res, dt, dx, re, w, omega = [float(x) for x in g["params"]]
    out = g["vort_zero.out.vn"]
    fluid = g["bc_mask"] == 0
    assert np.all(out[fluid] == np.float32(dt * w) * np.float32(0.1))
    assert np.array_equal(out[~fluid], g["vort_zero.in.vn"][~fluid])


TRAJ = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


@pytest.mark.parametrize("fname", TRAJ)
def test_trajectory_against_golden(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    cfg = traj_config(g)
    sim = make_oracle(g, cfg)
    for step in range(1, max(cfg["snaps"]) + 1):
        sim.update()
        if step in cfg["snaps"]:
            for k, a in sim.fields().items():
                _eq(a, g[f"step{step}.{k}"], f"{fname} step {step} {k}")
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
        if f"final.{name}.current" in g:
            for which in ("current", "next"):
                _eq(getattr(getattr(sim, name), which), g[f"final.{name}.{which}"], f"{fname} final {name}.{which}")
    if sim.vc is not None:
        _eq(sim.vc.vorticity, g["final.vorticity"], "vorticity")
        _eq(sim.vc.vorticity_abs, g["final.vorticity_abs"], "vorticity_abs")


def test_known_answers_predict_p():
    """v = 0, p = const -> const; v = 0, p linear in i -> interior fixed point (pressure_updater.py:28-38)."""
    X, Y = 12, 8
    mask = np.zeros((X, Y), np.uint8)
    bc = O.OracleBC(np.zeros((X, Y, 2), np.float32), mask)
    jac = O.OracleJacobi(bc, 0.01, 0.1, 1)
    v = np.zeros((X, Y, 2), np.float32)
    pc = np.full((X, Y), 3.25, np.float32); pn = np.zeros_like(pc)
    jac.sweep(pn, pc, v)
    assert np.all(pn == np.float32(3.25))
    pc = (np.arange(X, dtype=np.float32)[:, None] * np.ones((1, Y), np.float32)).copy()
    jac.sweep(pn, pc, v)
    assert np.array_equal(pn[1:-1, :], pc[1:-1, :])


def test_known_answers_uniform_flow_and_limit():
    X, Y = 16, 12
    mask = np.zeros((X, Y), np.uint8)
    v = np.empty((X, Y, 2), np.float32); v[..., 0] = 0.75; v[..., 1] = -0.25
    p = np.zeros((X, Y), np.float32)
    for scheme in (0, 1):   # uniform v: advection, diffusion, pressure gradient all exactly 0
        vn = np.full_like(v, 9.0)
        O._call("oracle_mac_update", np.float32, X, Y, 0.01, 0.1, 100.0, scheme, mask, vn, v, p)
        assert np.array_equal(vn, v)
    big = np.zeros((X, Y, 2), np.float32); big[..., 0] = 12.0; big[..., 1] = 16.0      # |v| = 20 -> exactly 10 * v/20
    O.limit_field(big)
    assert np.all(big[..., 0] == np.float32(10.0) * (np.float32(12.0) / np.float32(20.0)))
    small = v.copy(); O.limit_field(small); assert np.array_equal(small, v)


@pytest.mark.parametrize("seed", range(6))
def test_bc_cell_list_equals_full_scan(seed):
    """The O(perimeter) form of the velocity / pressure boundary kernels (OracleBC(cell_list=True), used for the large-grid CPU
    baseline) against the literal full scan, on the reference scenes and on random masks with thin walls / hazards."""
    rng = np.random.default_rng(seed)
    if seed < 3:
        g = golden(f"kernels_bc{(3, 5, 6)[seed]}.npz")
        const, mask = g["bc_const"], g["bc_mask"]
    else:
        X, Y = int(rng.integers(8, 60)), int(rng.integers(8, 40))
        mask = (rng.random((X, Y)) < 0.25).astype(np.uint8)
        mask[rng.integers(0, X, 6), :] = 1
        mask[:2, :] = 2; mask[-2:, :] = 3
        mask[:, [0, 1, Y - 2, Y - 1]] = 1
        const = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
    a, b = O.OracleBC(const, mask, cell_list=False), O.OracleBC(const, mask, cell_list=True)
    assert a.cells is None and b.cells is not None
    for _ in range(3):
        v = rng.uniform(-1, 1, mask.shape + (2,)).astype(np.float32)
        p = rng.uniform(-10, 10, mask.shape).astype(np.float32)
        v2, p2 = v.copy(), p.copy()
        a.set_velocity_boundary_condition(v); b.set_velocity_boundary_condition(v2)
        a.set_pressure_boundary_condition(p); b.set_pressure_boundary_condition(p2)
        assert np.array_equal(v, v2) and np.array_equal(p, p2)
