"""Odd resolutions (X = 2 res not a multiple of 4): the kernels on lanes of 2 cells (csrc/fs_k34n.h, fs_rbpair.h) need an even width only,
so the default solvers keep their fused passes there - bit-identical to the CPU oracle, and the launches are the fused ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bc_id,res", [(1, 51), (2, 75), (5, 125), (3, 81)])
@pytest.mark.parametrize("scheme", ["cip", "kk"])
def test_odd_resolution_runs_the_fused_passes(bc_id, res, scheme, hip_lib):
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    from oracle import oracle as O
    const, mask, _ = create_scene_arrays(bc_id, res)
    assert mask.shape[0] % 4 == 2
    dt, dx, re, vc = 0.05 / res, 1.0 / res, 1000.0, 5.0
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    dev = bc.device
    vcobj = fs.VorticityConfinement(bc, dt, dx, vc)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    solver = (fs.CipMacSolver(bc, pu, dt, dx, re, vcobj) if scheme == "cip"
              else fs.MacSolver(bc, pu, fs.advect_kk_scheme, dt, dx, re, vcobj))
    ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=("rbsor", 1.3, 2))
    try:
        dev.profile(True)
        for step in range(8):
            solver.update()
            ref.update()
            for a, e, name in zip([f.to_numpy() for f in solver.get_fields()], list(ref.fields().values()), ("v", "p")):
                assert np.array_equal(a, e, equal_nan=True), f"bc{bc_id} res {res} {scheme} step {step + 1} {name}"
        names = set(dev.profile_report())
        assert "vort_confine" in names, names
        if scheme == "cip":
            assert {"cip_nonadv", "cip_grad_advect_rt"} <= names and "cip_advect" not in names, names
        else:
            assert "mac_update_kk" in names, names
        if dev.rb_pair_ok:
            assert "rbsor_pair" in names, names
    finally:
        dev.close()


@pytest.mark.parametrize("scheme", ["cip", "upwind"])
def test_odd_resolution_with_dye(scheme, hip_lib):
    """The dye solvers at an odd resolution: fused dye passes (CIP) / the generic dye update (upwind), against the oracle."""
    import fs
    from fs.boundary_condition import DyeBoundaryCondition, create_scene_arrays
    from oracle import oracle as O
    res = 57
    const, mask, dye = create_scene_arrays(2, res)
    dt, dx, re, vc = 0.05 / res, 1.0 / res, 1000.0, 5.0
    fs.runtime.init(gpu=0, dtype="f32")
    bc = DyeBoundaryCondition(const, dye, mask)
    dev = bc.device
    vcobj = fs.VorticityConfinement(bc, dt, dx, vc)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    solver = (fs.DyeCipMacSolver(bc, pu, dt, dx, re, vcobj) if scheme == "cip"
              else fs.DyeMacSolver(bc, pu, fs.advect_upwind, dt, dx, re, vcobj))
    ref = O.make_simulator(const, mask, dye, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=("rbsor", 1.3, 2))
    try:
        dev.profile(True)
        for step in range(6):
            solver.update()
            ref.update()
            for a, e, name in zip([f.to_numpy() for f in solver.get_fields()], list(ref.fields().values()), ("v", "p", "dye")):
                assert np.array_equal(a, e, equal_nan=True), f"{scheme} step {step + 1} {name}"
        names = set(dev.profile_report())
        if scheme == "cip":
            assert {"cip_grad_advect_dye", "cip_nonadv_dye", "cip_grad_advect_rt"} <= names, names
    finally:
        dev.close()


@pytest.mark.parametrize("res", [51, 125])
def test_odd_resolution_literal_jacobi_sweeps_on_pair_lanes(res, hip_lib):
    """Round 6: at odd res the literal Jacobi sweep runs k_jacobi_ov2 (lanes of 2 cells: any even width) instead of one cell per lane,
    and limit_field rides in the velocity boundary launch as at even res (its quads stop at the row's width).  Jacobi(4) = the literal sweeps."""
    import fs
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    const, mask, _ = create_scene_arrays(2, res)
    assert mask.shape[0] % 4 == 2
    dt, dx, re, vc = 0.05 / res, 1.0 / res, 1.0e6, 5.0
    fs.runtime.init(gpu=0, dtype="f32")
    sim = fs.FluidSimulator.create(2, res, dt, dx, re, vc, "cip", pressure_updater=("jacobi", 4))
    dev = sim._solver._bc.device
    ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=re, vor_eps=vc, updater=("jacobi", 4))
    try:
        dev.profile(True)
        for step in range(6):      # (no download in between: a download makes a deferred limit pass run as its own launch first)
            sim.step()
            ref.update()
        names = set(dev.profile_report())
        ks = dev.profile_kernels("jacobi_sweep")
        assert ks and all(k.startswith("fs::k_jacobi_ov2<") for k in ks), ks
        assert "limit_field" not in names and "fs::k_velocity_bc_limit<float>" in dev.profile_kernels("velocity_bc"), (names, dev.profile_kernels("velocity_bc"))
        out = sim.field_to_numpy()
        for name, e in ref.fields().items():
            assert np.array_equal(out[name], e, equal_nan=True), f"res {res} after 6 steps: {name}"
    finally:
        dev.close()


def test_limit_pass_at_odd_width_stops_at_the_row_end(hip_lib):
    """limit_field with speeds above the limit everywhere at X = 2 * 51: the quad-wide pass as its own launch and merged into the boundary launch
    (deferred by the solver's end-of-step call) against the oracle."""
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    from oracle import oracle as O
    res = 51
    const, mask, _ = create_scene_arrays(2, res)
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    dev = bc.device
    try:
        rng = np.random.default_rng(5)
        v0 = rng.uniform(-20, 20, mask.shape + (2,)).astype(np.float32)
        v = dev.alloc(2)
        v.from_numpy(v0)
        dev.limit_field(10.0, v)
        exp = v0.copy()
        O.limit_field(exp, 10.0)
        assert np.array_equal(v.to_numpy(), exp)
        v.from_numpy(v0)
        assert dev.limit_deferral, "the merged launch must be available at odd res"
        dev.limit_field(10.0, v, defer=True)
        assert v.pending_limit is not None
        dev.velocity_bc(v)
        O.OracleBC(const, mask).set_velocity_boundary_condition(exp)
        assert np.array_equal(v.to_numpy(), exp)
    finally:
        dev.close()
