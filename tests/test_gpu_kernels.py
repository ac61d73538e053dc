"""GPU parity, kernel by kernel: every HIP kernel on the step() path against (a) the golden vectors captured
from the reference's own kernel source and (b) the CPU oracle, on the same seeded inputs (res 16, scenes 1-6).

Bar: bit-exact.  The kernels evaluate the reference's expression trees in IEEE f32 without FMA contraction,
as the oracle does, so any difference is a bug (tolerance 0; NaNs must match too)."""
import numpy as np
import pytest
from conftest import golden

pytestmark = pytest.mark.gpu

SCENES = [1, 2, 3, 4, 5, 6]


class Ctx:
    def __init__(self, n):
        import fs
        from fs.boundary_condition import DyeBoundaryCondition
        fs.runtime.init(gpu=0, dtype="f32")
        self.g = golden(f"kernels_bc{n}.npz")
        res, self.dt, self.dx, self.re, self.w, self.omega = [float(x) for x in self.g["params"]]
        self.bc = DyeBoundaryCondition(self.g["bc_const"], self.g["bc_dye"], self.g["bc_mask"])
        self.dev = self.bc.device

    def field(self, name, key):
        a = self.g[f"{name}.in.{key}"]
        f = self.dev.alloc(1 if a.ndim == 2 else a.shape[2])
        f.from_numpy(a)
        return f

    def expect(self, name, key, field):
        got = field.to_numpy()
        exp = self.g[f"{name}.out.{key}"]
        assert got.dtype == exp.dtype and got.shape == exp.shape
        assert np.array_equal(got, exp, equal_nan=True), f"{name}.{key}: max|d|={np.nanmax(np.abs(got - exp))}"


@pytest.fixture(scope="module", params=SCENES)
def cx(request, hip_lib):
    c = Ctx(request.param)
    yield c
    c.dev.close()


def test_velocity_bc(cx):
    v = cx.field("velocity_bc", "v"); cx.bc.set_velocity_boundary_condition(v); cx.expect("velocity_bc", "v", v)


def test_pressure_bc(cx):
    p = cx.field("pressure_bc", "p"); cx.bc.set_pressure_boundary_condition(p); cx.expect("pressure_bc", "p", p)


def test_dye_bc(cx):
    d = cx.field("dye_bc", "dye"); cx.bc.set_dye_boundary_condition(d); cx.expect("dye_bc", "dye", d)


@pytest.mark.parametrize("tag,code", [("upwind", 0), ("kk", 1)])
def test_mac_update_and_dye(cx, tag, code):
    n = f"mac_update_{tag}"
    vn = cx.field(n, "vn")
    cx.dev.mac_update(code, cx.dt, cx.dx, cx.re, vn, cx.field(n, "vc"), cx.field(n, "pc"))
    cx.expect(n, "vn", vn)
    n = f"mac_dye_{tag}"
    dn = cx.field(n, "dn")
    cx.dev.mac_dye(code, cx.dt, cx.dx, dn, cx.field(n, "dc"), cx.field(n, "vc"))
    cx.expect(n, "dn", dn)


def test_cip_set_grad(cx):
    n = "cip_set_grad"
    fx, fy = cx.field(n, "fx"), cx.field(n, "fy")
    cx.dev.cip_set_grad(cx.dx, fx, fy, cx.field(n, "f"))
    cx.expect(n, "fx", fx); cx.expect(n, "fy", fy)


def test_cip_nonadv(cx):
    n = "cip_nonadv"
    fn = cx.field(n, "fn")
    cx.dev.cip_nonadv(cx.dt, cx.dx, cx.re, fn, cx.field(n, "fc"), cx.field(n, "pc"))
    cx.expect(n, "fn", fn)
    n = "cip_nonadv_dye"
    dn = cx.field(n, "dn")
    cx.dev.cip_nonadv_dye(cx.dt, cx.dx, cx.re, dn, cx.field(n, "dc"))
    cx.expect(n, "dn", dn)


@pytest.mark.parametrize("c", [2, 3])
def test_cip_nonadv_grad(cx, c):
    n = f"cip_nonadv_grad_c{c}"
    fxn, fyn = cx.field(n, "fxn"), cx.field(n, "fyn")
    cx.dev.cip_nonadv_grad(cx.dx, fxn, fyn, cx.field(n, "fxc"), cx.field(n, "fyc"), cx.field(n, "fc"), cx.field(n, "fn"))
    cx.expect(n, "fxn", fxn); cx.expect(n, "fyn", fyn)


@pytest.mark.parametrize("c", [2, 3])
def test_cip_advect(cx, c):
    n = f"cip_advect_c{c}"
    fn, fxn, fyn, fc = cx.field(n, "fn"), cx.field(n, "fxn"), cx.field(n, "fyn"), cx.field(n, "fc")
    v = fc if c == 2 else cx.field(n, "v")
    cx.dev.cip_advect(cx.dt, cx.dx, fn, fxn, fyn, fc, cx.field(n, "fxc"), cx.field(n, "fyc"), v)
    cx.expect(n, "fn", fn); cx.expect(n, "fxn", fxn); cx.expect(n, "fyn", fyn)


@pytest.mark.parametrize("tag", ["rand", "zero"])
def test_vorticity_confinement(cx, tag):
    """'zero' is hazard H4: |grad|omega|| == 0 -> 0/0 = NaN -> NaN-ignoring clamp -> +0.1 on both components."""
    import fs
    n = f"vort_{tag}"
    vc = fs.VorticityConfinement(cx.bc, cx.dt, cx.dx, cx.w)
    vn, vcur = cx.field(n, "vn"), cx.field(n, "vc")
    vc._calc_vorticity(vcur)
    cx.expect(n, "vorticity", vc.vorticity); cx.expect(n, "vorticity_abs", vc.vorticity_abs)
    vc._add_vorticity(vn, vcur)
    cx.expect(n, "vn", vn)


@pytest.mark.parametrize("src", [False, True])
def test_pressure_sweeps(cx, src):
    import fs
    jac = fs.JacobiPressureUpdater(cx.bc, cx.dt, cx.dx, 3, precompute_source=src)
    sor = fs.RedBlackSorPressureUpdater(cx.bc, cx.dt, cx.dx, cx.omega, 2, precompute_source=src)
    n = "jacobi_sweep"
    pn, vc = cx.field(n, "pn"), cx.field(n, "vc")
    if src:
        cx.dev.poisson_source(cx.dt, cx.dx, jac._src, vc)
    jac._update(pn, cx.field(n, "pc"), vc)
    cx.expect(n, "pn", pn)
    n = "rbsor_odd"
    pn, vc = cx.field(n, "pn"), cx.field(n, "vc")
    if src:
        cx.dev.poisson_source(cx.dt, cx.dx, sor._src, vc)
    sor._update_pressures_odd(pn, cx.field(n, "pc"), vc)
    cx.expect(n, "pn", pn)
    n = "rbsor_even"
    pn, vc = cx.field(n, "pn"), cx.field(n, "vc")
    if src:
        cx.dev.poisson_source(cx.dt, cx.dx, sor._src, vc)
    sor._update_pressures_even(pn, pn, vc)
    cx.expect(n, "pn", pn)


@pytest.mark.parametrize("src", [False, True])
@pytest.mark.parametrize("tag", ["jacobi3", "rbsor2"])
def test_pressure_update_choreography(cx, tag, src):
    """n_iter x {K7, sweep(s), swap} incl. the red-black in-place pass on the stale buffer (hazard H5)."""
    import fs
    n = f"pressure_update_{tag}"
    upd = (fs.JacobiPressureUpdater(cx.bc, cx.dt, cx.dx, 3, precompute_source=src) if tag == "jacobi3"
           else fs.RedBlackSorPressureUpdater(cx.bc, cx.dt, cx.dx, cx.omega, 2, precompute_source=src))
    p = fs.DoubleBuffer(cx.bc.get_resolution(), 1)
    p.current.from_numpy(cx.g[f"{n}.in.p_current"]); p.next.from_numpy(cx.g[f"{n}.in.p_next"])
    upd.update(p, cx.field(n, "v"))
    cx.expect(n, "p_current", p.current); cx.expect(n, "p_next", p.next)


def test_rbsor_fused_iteration_matches_two_half_sweeps(cx):
    """odd golden input -> odd pass -> even pass, in one fused launch, against the oracle's two half-sweeps."""
    from oracle import oracle as O
    n = "rbsor_odd"
    pn0, pc0, v0 = cx.g[f"{n}.in.pn"], cx.g[f"{n}.in.pc"], cx.g[f"{n}.in.vc"]
    ob = O.OracleBC(cx.g["bc_const"], cx.g["bc_mask"])
    sor = O.OracleRedBlackSor(ob, cx.dt, cx.dx, cx.omega, 1)
    ref = pn0.copy()
    sor.half(1, ref, pc0, v0); sor.half(0, ref, ref, v0)
    pn = cx.field(n, "pn")
    cx.dev.rbsor_iteration(cx.dt, cx.dx, cx.omega, pn, cx.field(n, "pc"), cx.field(n, "vc"))
    got = pn.to_numpy()
    assert np.array_equal(got, ref), f"max|d| = {np.abs(got - ref).max()}"


def test_limit_and_clamp(cx):
    from fs.solver import VELOCITY_LIMIT, clamp_field, limit_field
    v = cx.field("limit_field", "v"); limit_field(v, VELOCITY_LIMIT); cx.expect("limit_field", "v", v)
    d = cx.field("clamp_field", "dye"); clamp_field(d, 0.0, 1.0); cx.expect("clamp_field", "dye", d)


def test_residual_matches_oracle_definition(cx):
    """New diagnostic (no reference counterpart): sum over not-wall cells of (predict_p(p) - p)^2."""
    from oracle import oracle as O
    n = "jacobi_sweep"
    pc, vc = cx.g[f"{n}.in.pc"], cx.g[f"{n}.in.vc"]
    s, cnt = cx.dev.poisson_residual(cx.dt, cx.dx, cx.field(n, "pc"), cx.field(n, "vc"))
    ob = O.OracleBC(cx.g["bc_const"], cx.g["bc_mask"])
    pn = pc.copy()
    O.OracleJacobi(ob, cx.dt, cx.dx, 1).sweep(pn, pc, vc)
    nw = cx.g["bc_mask"] != 1
    ref = float(np.sum((pn[nw] - pc[nw]).astype(np.float64) ** 2))   # residual formed in f32, squared in f64
    assert cnt == nw.sum()
    assert abs(s - ref) <= 1e-10 * max(ref, 1.0)


def test_roundtrip_upload_download(cx):
    rng = np.random.default_rng(7)
    X, Y = cx.bc.get_resolution()
    for c in (1, 2, 3):
        a = rng.standard_normal((X, Y) if c == 1 else (X, Y, c)).astype(np.float32)
        f = cx.dev.alloc(c)
        f.from_numpy(a)
        assert np.array_equal(f.to_numpy(), a)
