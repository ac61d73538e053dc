"""limit_field behind the "hot" flag (csrc/fs_device.h): the pass over the whole velocity field is skipped while no kernel has
stored a speed above 8 into the buffer.  The result must be the reference's for EVERY input, so these cases drive the speed over
the limit through each kind of writer - boundary values, the transport kernels, uploads, fills - and compare with the CPU
oracle bit for bit; plus gate on == gate off on a regular run."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(const, mask, scheme, vc, dt, dx, re=1e6, dtype="f32"):
    import fs
    from fs.boundary_condition import BoundaryCondition
    from oracle import oracle as O
    fs.runtime.init(gpu=0, dtype=dtype)
    bc = BoundaryCondition(const, mask)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    v = fs.VorticityConfinement(bc, dt, dx, vc) if vc else None
    if scheme == "cip":
        solver = fs.CipMacSolver(bc, pu, dt, dx, re, v)
    else:
        solver = fs.MacSolver(bc, pu, fs.advect_upwind if scheme == "upwind" else fs.advect_kk_scheme, dt, dx, re, v)
    ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc,
                           dtype=np.float32 if dtype == "f32" else np.float64)
    return fs.FluidSimulator(solver), ref


def _compare(sim, ref, steps, tag):
    for step in range(1, steps + 1):
        sim.step()
        ref.update()
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e, equal_nan=True), f"{tag}: step {step} {k}"
    return out


@pytest.mark.parametrize("scheme,vc", [("cip", 5.0), ("upwind", None), ("kk", 5.0)])
@pytest.mark.parametrize("res", [32, 130])
def test_inflow_faster_than_the_limit(scheme, vc, res, hip_lib):
    """bc_const asks for an inflow speed of 30: the velocity BC writes it every step, limit_field cuts it back to 10."""
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(2, res)
    const = const.copy()
    const[mask == 2] *= np.float32(30.0) / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    sim, ref = _pair(const, mask, scheme, vc, 0.05 / res, 1.0 / res)
    try:
        out = _compare(sim, ref, 4, f"inflow30 {scheme}")
        speed = np.sqrt((out["v"] ** 2).sum(-1))
        assert speed.max() <= 10.0001 and speed.max() > 9.99         # the limiter did act
    finally:
        sim._solver._bc.device.close()


@pytest.mark.parametrize("scheme", ["cip", "upwind"])
def test_transport_kernels_push_the_speed_over_the_limit(scheme, hip_lib):
    """A huge pressure gradient uploaded into p: the first transport kernel of the step produces speeds far above 10."""
    from fs.boundary_condition import create_scene_arrays
    res = 64
    const, mask, _ = create_scene_arrays(5, res)
    sim, ref = _pair(const, mask, scheme, 5.0, 0.05 / res, 1.0 / res)
    try:
        X, Y = mask.shape
        p0 = (np.linspace(0, 4.0e4, X, dtype=np.float32)[:, None] * np.ones((1, Y), np.float32)).astype(np.float32)
        sim._solver.p.current.from_numpy(p0)
        ref.p.current[...] = p0
        out = _compare(sim, ref, 2, f"pgrad {scheme}")          # (a third step of this abuse turns CIP into NaN - on both sides alike)
        assert np.nanmax(np.sqrt((out["v"] ** 2).sum(-1))) > 9.99
    finally:
        sim._solver._bc.device.close()


@pytest.mark.parametrize("how", ["upload", "fill"])
def test_uploaded_and_filled_fields_are_limited(how, hip_lib):
    from fs.boundary_condition import create_scene_arrays
    res = 48
    const, mask, _ = create_scene_arrays(1, res)
    sim, ref = _pair(const, mask, "cip", None, 0.05 / res, 1.0 / res)
    try:
        if how == "upload":
            rng = np.random.default_rng(5)
            v0 = rng.uniform(-14, 14, mask.shape + (2,)).astype(np.float32)
            sim._solver.v.current.from_numpy(v0)
            ref.v.current[...] = v0
        else:
            sim._solver.v.current.fill(9.0)          # speed 12.7 everywhere
            ref.v.current[...] = np.float32(9.0)
        _compare(sim, ref, 3, how)
    finally:
        sim._solver._bc.device.close()


def test_gate_on_equals_gate_off(hip_lib, monkeypatch):
    """The same 30 steps with the gate disabled (FS_LIMIT_GATE=0: full pass every step): identical state, every buffer."""
    import fs
    res = 256
    outs = []
    for gate in ("1", "0"):
        monkeypatch.setenv("FS_LIMIT_GATE", gate)
        fs.runtime.init(gpu=0)
        sim = fs.FluidSimulator.create(5, res, 0.05 / res, 1.0 / res, 1e6, 5.0, "cip")
        sim.run(30, graph=(gate == "1"))
        s = sim._solver
        outs.append({n + w: getattr(getattr(s, n), w).to_numpy() for n in ("v", "p", "vx", "vy") for w in ("current", "next")})
        s._bc.device.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert float(np.abs(outs[0]["vcurrent"]).max()) < 8.0      # a healthy run never comes near the gate's threshold


@pytest.mark.parametrize("scheme,vc,res,graph", [("cip", 5.0, 64, False), ("upwind", None, 130, False), ("kk", 5.0, 512, False), ("cip", 5.0, 512, True),
                                                 ("upwind", None, 64, True)])
def test_deferred_limit_rides_with_the_next_boundary_kernel(scheme, vc, res, graph, hip_lib, monkeypatch):
    """Round 4: the solvers' end-of-step limit_field is deferred and runs inside the NEXT step's velocity boundary launch
    (csrc/fs_march.h k_velocity_bc_limit) unless something looks at the field first.  Nothing looks here for several steps, and the inflow
    of 30 keeps the buffer's flag up: every step takes the kernel's rare path - the limit pass shared among the boundary kernel's
    workgroups (hundreds at res 512) and a grid barrier - and must still be the reference's result, eagerly and as a replayed hipGraph."""
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(2, res)
    const = const.copy()
    const[mask == 2] *= np.float32(30.0) / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    steps = 40 if graph else 7
    monkeypatch.setenv("FS_LIMIT_DEFER", "1")
    sim, ref = _pair(const, mask, scheme, vc, 0.05 / res, 1.0 / res)
    monkeypatch.setenv("FS_LIMIT_DEFER", "0")
    plain, _ = _pair(const, mask, scheme, vc, 0.05 / res, 1.0 / res)
    try:
        dev = sim._solver._bc.device
        assert dev.limit_deferral and not plain._solver._bc.device.limit_deferral
        if graph:
            sim.run(steps)                    # one eager step, the capture of a period, replays, eager remainder
        else:
            for _ in range(steps):
                sim.step()
        assert sim._solver.v.current.pending_limit is not None      # still owed: nobody has looked
        for _ in range(steps):
            plain.step()
            if res <= 130:
                ref.update()
        out, exp = sim.field_to_numpy(), plain.field_to_numpy()
        assert sim._solver.v.current.pending_limit is None
        for k in exp:
            assert np.array_equal(out[k], exp[k], equal_nan=True), f"deferred vs immediate: {k}"
            if res <= 130:
                assert np.array_equal(out[k], ref.fields()[k], equal_nan=True), f"deferred vs oracle: {k}"
        for name in ("v", "p"):
            for which in ("current", "next"):
                a, b = getattr(getattr(sim._solver, name), which).to_numpy(), getattr(getattr(plain._solver, name), which).to_numpy()
                assert np.array_equal(a, b, equal_nan=True), f"{name}.{which}"
        speed = np.sqrt((out["v"] ** 2).sum(-1))
        if np.isfinite(speed).all():          # (an inflow of 30 blows the long runs up - identically on both sides, which is all that matters here)
            assert speed.max() <= 10.0001 and speed.max() > 9.99
    finally:
        sim._solver._bc.device.close()
        plain._solver._bc.device.close()


@pytest.mark.parametrize("scheme,res,graph", [("cip", 64, False), ("cip", 130, True), ("upwind", 64, False), ("cip", 400, True)])
def test_dye_solvers_take_the_limit_along_and_drop_the_inflow_clamp(scheme, res, graph, hip_lib, monkeypatch):
    """Round 4: in the dye solvers the flow step's limit_field rides with the dye boundary kernel (csrc/fs_march.h k_dye_bc_limit), and the
    end-of-step clamp of the inflow cells is dropped when the next step's dye boundary kernel overwrites those very cells - or launched
    when something looks at the dye first.  Inflow of 30 (flag up: every step takes the limit pass + grid barrier on the dye kernel's few
    workgroups) and a dye colour of 1.7 on the inflow (the clamp matters), eagerly and as a replayed hipGraph, with a look at the fields in
    mid-run, against the immediate launches and (small grids) the CPU oracle."""
    import fs
    from fs.boundary_condition import DyeBoundaryCondition, create_scene_arrays
    from oracle import oracle as O
    const, mask, dye0 = create_scene_arrays(2, res)
    const, dye0 = const.copy(), dye0.copy()
    const[mask == 2] *= np.float32(30.0) / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    dye0[mask == 2] *= np.float32(1.7)
    dt, dx = 0.05 / res, 1.0 / res

    def build():
        fs.runtime.init(gpu=0, dtype="f32")
        bc = DyeBoundaryCondition(const, dye0, mask)
        pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
        vc = fs.VorticityConfinement(bc, dt, dx, 5.0) if scheme == "cip" else None
        solver = fs.DyeCipMacSolver(bc, pu, dt, dx, 1e6, vc) if scheme == "cip" else fs.DyeMacSolver(bc, pu, fs.advect_upwind, dt, dx, 1e6, vc)
        return fs.DyeFluidSimulator(solver)
    monkeypatch.setenv("FS_LIMIT_DEFER", "1")
    sim = build()
    monkeypatch.setenv("FS_LIMIT_DEFER", "0")
    plain = build()
    ref = O.make_simulator(const, mask, dye0, scheme=scheme, dt=dt, dx=dx, re=1e6, vor_eps=5.0 if scheme == "cip" else None) if res <= 130 else None
    try:
        dev = sim._solver._bc.device
        assert dev.limit_deferral and dev.dye_limit_merge and not plain._solver._bc.device.limit_deferral
        done = 0
        for chunk in ((40, 3, 23) if graph else (3, 2, 4)):
            if graph:
                sim.run(chunk)
            else:
                for _ in range(chunk):
                    sim.step()
            done += chunk
            s = sim._solver
            assert s.v.current.pending_limit is None                       # the dye boundary kernel has taken it along
            if scheme == "cip":
                assert s.dye.current.pending_clamp is not None             # owed until somebody looks
            for _ in range(chunk):
                plain.step()
                if ref is not None:
                    ref.update()
            out, exp = sim.field_to_numpy(), plain.field_to_numpy()
            assert s.dye.current.pending_clamp is None
            for k in exp:
                assert np.array_equal(out[k], exp[k], equal_nan=True), f"deferred vs immediate after {done} steps: {k}"
                if ref is not None:
                    assert np.array_equal(out[k], ref.fields()[k], equal_nan=True), f"deferred vs oracle after {done} steps: {k}"
            assert np.nanmax(out["dye"]) <= 1.0
        for name in ("v", "p", "dye"):
            for which in ("current", "next"):
                a, b = getattr(getattr(sim._solver, name), which).to_numpy(), getattr(getattr(plain._solver, name), which).to_numpy()
                assert np.array_equal(a, b, equal_nan=True), f"{name}.{which}"
    finally:
        sim._solver._bc.device.close()
        plain._solver._bc.device.close()


@pytest.mark.parametrize("scheme,vc,graph", [("cip", 5.0, False), ("kk", 5.0, True)])
def test_deferred_limit_in_f64_and_on_tiny_grids(scheme, vc, graph, hip_lib, monkeypatch):
    """The merged limit + boundary launch at double precision, and on a grid with fewer rows (24) than the 64 workgroups the launch may carry:
    flag up on every step (inflow 30), against the immediate launches and the f64 oracle."""
    from fs.boundary_condition import create_scene_arrays
    res = 24
    const, mask, _ = create_scene_arrays(2, res)
    const = const.copy()
    const[mask == 2] *= 30.0 / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    monkeypatch.setenv("FS_LIMIT_DEFER", "1")
    sim, ref = _pair(const, mask, scheme, vc, 0.05 / res, 1.0 / res, dtype="f64")
    monkeypatch.setenv("FS_LIMIT_DEFER", "0")
    plain, _ = _pair(const, mask, scheme, vc, 0.05 / res, 1.0 / res, dtype="f64")
    try:
        assert sim._solver._bc.device.limit_deferral
        steps = 36 if graph else 6
        if graph:
            sim.run(steps)
        else:
            for _ in range(steps):
                sim.step()
        assert sim._solver.v.current.pending_limit is not None
        for _ in range(steps):
            plain.step()
            ref.update()
        out, exp = sim.field_to_numpy(), plain.field_to_numpy()
        for k in exp:
            assert out[k].dtype == np.float64
            assert np.array_equal(out[k], exp[k], equal_nan=True), f"f64 deferred vs immediate: {k}"
            assert np.array_equal(out[k], ref.fields()[k], equal_nan=True), f"f64 deferred vs oracle: {k}"
    finally:
        sim._solver._bc.device.close()
        plain._solver._bc.device.close()


@pytest.mark.parametrize("graph", [False, True])
def test_a_run_that_went_hot_stops_deferring(graph, hip_lib, monkeypatch):
    """Deferring limit_field is for runs that never need it.  With the buffers' flag up the pass runs on every step, and inside a boundary launch it is
    shared by a few dozen workgroups (res 4096: 130 us against 47 as its own launch): FluidSimulator looks at the flag between launch sequences
    (fs_field_hot) and goes back to the separate launch.  Same fields either way."""
    from fs.boundary_condition import create_scene_arrays
    res = 64
    const, mask, _ = create_scene_arrays(2, res)
    const = const.copy()
    const[mask == 2] *= np.float32(30.0) / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    monkeypatch.setenv("FS_LIMIT_DEFER", "1")
    sim, ref = _pair(const, mask, "cip", 5.0, 0.05 / res, 1.0 / res)
    try:
        dev = sim._solver._bc.device
        assert dev.limit_deferral and not dev.field_hot(sim._solver.v.current)
        sim.run(20, graph=graph)                       # cold at the start of this call: deferred all the way
        assert dev.limit_deferral and sim._solver.v.current.pending_limit is not None
        assert dev.field_hot(sim._solver.v.current) or dev.field_hot(sim._solver.v.next)
        sim.run(20, graph=graph)                       # hot now: the owed pass is launched, the rest of the run does not defer
        assert not dev.limit_deferral and sim._solver.v.current.pending_limit is None
        for _ in range(40):
            ref.update()
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e, equal_nan=True), k
    finally:
        sim._solver._bc.device.close()


@pytest.mark.parametrize("vc", [5.0, None])
def test_one_fast_component_does_not_keep_the_limit_pass_running(vc, hip_lib, monkeypatch):
    """Round 4, flag word [3]: the fused K3+K4 pass sees one velocity component per wave and must raise at |u| > 7.04 although the limiter only acts
    above a speed of 10.  With vorticity confinement the buffer limit_field looks at is written by a kernel that sees both components: its flag
    stays down for a flow of (7.5, 0) - without it, K3+K4's output goes straight to limit_field and is judged conservatively.  Same fields as the
    oracle either way."""
    from fs.boundary_condition import create_scene_arrays
    res = 64
    const, mask, _ = create_scene_arrays(2, res)
    monkeypatch.setenv("FS_LIMIT_DEFER", "1")
    sim, ref = _pair(const, mask, "cip", vc, 0.05 / res, 1.0 / res)
    try:
        dev = sim._solver._bc.device
        # (7.5, 0) in the fluid cells at least 6 cells away from anything that is not fluid: what K3+K4 carries on walls / inflow / outflow stays slow
        # (those cells are judged per component as before: word [0])
        far = mask == 0
        for _ in range(6):
            f = far.copy()
            f[1:, :] &= far[:-1, :]; f[:-1, :] &= far[1:, :]; f[:, 1:] &= far[:, :-1]; f[:, :-1] &= far[:, 1:]
            f[0, :] = f[-1, :] = False; f[:, 0] = f[:, -1] = False
            far = f
        assert far.sum() > 100
        v0 = np.zeros(mask.shape + (2,), np.float32)
        v0[far, 0] = 7.5
        sim._solver.v.current.from_numpy(v0)
        ref.v.current[...] = v0
        assert not dev.field_hot(sim._solver.v.current)                  # speed 7.5: the upload's exact scan leaves the flag down
        for _ in range(2):
            sim.step()
            ref.update()
        hot = dev.field_hot(sim._solver.v.current)
        assert hot == (vc is None), f"flag of the buffer limit_field reads: {hot}"
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e, equal_nan=True), k
        speed = np.sqrt((out["v"].astype(np.float64) ** 2).sum(-1))
        assert 7.04 < np.nanmax(np.abs(out["v"][..., 0])) and np.nanmax(speed) < 9.9      # the case this test is about
    finally:
        sim._solver._bc.device.close()
