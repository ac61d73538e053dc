"""Host logic of the deferred passes (round 4, fs/runtime.py): the solvers' end-of-step limit_field rides with the next velocity boundary launch
(or, in the dye solvers, with the dye boundary launch), the dye's end-of-step clamp of the inflow cells is dropped when the next dye boundary
launch overwrites those cells - and both are launched at once when anything else looks at their field.  No GPU: the CPU stand-in device
(tests/oracle_device.py) provides the kernels from the oracle, with the two merged launches stated as what they stand for; what is tested is
which launches the product's Python issues, in which order, with which parity - and that the fields equal the oracle's at every look."""
import importlib

import numpy as np
import pytest

import conftest  # noqa: F401  (paths)

importlib.import_module("2d-fluid-simulator_amd")
import fs  # noqa: E402
from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition, create_scene_arrays  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle_device import OracleSlabDevice  # noqa: E402


class DeferringDevice(OracleSlabDevice):
    """The stand-in with the merged launches available: velocity_bc_limit = limit_field + velocity_bc, dye_bc_limit = limit_field(v) + dye_bc."""
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.log = []
        self.parity = {}

    def _p_limit_deferral_ok(self):
        return True

    def _p_dye_bc_limit_ok(self):
        return True

    def _p_kernel(self, name, *args):
        self.log.append(name)
        if name == "velocity_bc_limit":
            limit, v, parity, lb, le, lo, hi = args
            assert (lb, le) == (0, self.rows) and parity in (0, 1)
            assert self.parity.get(id(v), parity ^ 1) != parity, "two merged launches on one buffer with the same parity"      # include/fs_hip.h
            self.parity[id(v)] = parity
            super()._p_kernel("limit_field", limit, v, lb, le)
            super()._p_kernel("velocity_bc", v, lo, hi)
        elif name == "dye_bc_limit":
            limit, v, dye, lb, le, lo, hi = args
            super()._p_kernel("limit_field", limit, v, lb, le)
            super()._p_kernel("dye_bc", dye, lo, hi)
        else:
            super()._p_kernel(name, *args)


def _build(scheme, dye, res, hot):
    const, mask, dye0 = create_scene_arrays(2, res)
    const, dye0 = const.copy(), dye0.copy()
    if hot:          # an inflow of 30: limit_field acts on every step; a dye colour above 1 on the inflow: the clamp matters
        const[mask == 2] *= np.float32(30.0) / max(float(np.abs(const[mask == 2]).max()), 1e-6)
    dye0[mask == 2] *= np.float32(1.7)
    dt, dx = 0.05 / res, 1.0 / res
    fs.runtime.init(dtype="f32", device_cls=DeferringDevice)
    bc = DyeBoundaryCondition(const, dye0, mask) if dye else BoundaryCondition(const, mask)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    vc = fs.VorticityConfinement(bc, dt, dx, 5.0) if scheme == "cip" else None
    if scheme == "cip":
        solver = (fs.DyeCipMacSolver if dye else fs.CipMacSolver)(bc, pu, dt, dx, 1e6, vc)
    else:
        solver = (fs.DyeMacSolver if dye else fs.MacSolver)(bc, pu, fs.advect_upwind, dt, dx, 1e6, vc)
    sim = (fs.DyeFluidSimulator if dye else fs.FluidSimulator)(solver)
    ref = O.make_simulator(const, mask, dye0 if dye else None, scheme=scheme, dt=dt, dx=dx, re=1e6, vor_eps=5.0 if scheme == "cip" else None)
    return sim, ref


@pytest.mark.parametrize("hot", [False, True])
@pytest.mark.parametrize("scheme,dye", [("cip", False), ("cip", True), ("upwind", False), ("upwind", True)])
def test_deferred_passes_equal_the_reference_sequence(scheme, dye, hot):
    sim, ref = _build(scheme, dye, 24, hot)
    dev = sim._solver._bc.device
    try:
        assert dev.limit_deferral and (dev.dye_limit_merge or not dye)
        done = 0
        for chunk in (3, 1, 4, 2):
            mark = len(dev.log)
            for _ in range(chunk):
                sim.step()
                ref.update()
            done += chunk
            launched = dev.log[mark:]
            # nothing looked at the fields in between: no separate limit pass, no inflow clamp of the fused CIP dye path
            assert "limit_field" not in launched, launched
            merged = "dye_bc_limit" if dye else "velocity_bc_limit"
            # (dye solvers: the limit a step ends its flow part with is taken along by the same step's dye boundary launch; the others: by the
            #  NEXT step's velocity boundary launch - the first step after a look has nothing to take along)
            assert launched.count(merged) == (chunk if dye else chunk - 1), (merged, launched)
            if dye and scheme == "cip":
                assert "clamp_inflow" not in launched and sim._solver.dye.current.pending_clamp is not None
            if not dye:
                assert sim._solver.v.current.pending_limit is not None
            mark = len(dev.log)
            out = sim.field_to_numpy()                         # the look: what is owed is launched now, once
            flushed = dev.log[mark:]
            assert flushed.count("limit_field") == (0 if dye else 1) and flushed.count("clamp_inflow") == (1 if dye and scheme == "cip" else 0), flushed
            for k, e in ref.fields().items():
                assert np.array_equal(out[k], e, equal_nan=True), f"{scheme} dye={dye} hot={hot}: {k} after {done} steps"
            assert sim._solver.v.current.pending_limit is None
        if hot:
            speed = np.sqrt((out["v"] ** 2).sum(-1))
            assert np.nanmax(speed) > 9.99 and np.nanmax(speed) <= 10.0001      # the limiter did act
    finally:
        dev.close() if hasattr(dev, "close") else None


def test_an_upload_cancels_what_the_field_owes():
    """from_numpy / fill overwrite every cell: a deferred limit or clamp of the old content must not run on the new one."""
    sim, ref = _build("cip", True, 24, False)
    dev = sim._solver._bc.device
    for _ in range(2):
        sim.step()
        ref.update()
    s = sim._solver
    assert s.dye.current.pending_clamp is not None
    new = np.full(s.dye.current.to_numpy().shape, 1.5, np.float32)       # (to_numpy flushes; the next step re-arms)
    sim.step(); ref.update()
    assert s.dye.current.pending_clamp is not None
    s.dye.current.from_numpy(new)
    ref.dye.current[...] = new
    assert s.dye.current.pending_clamp is None
    mark = len(dev.log)
    got = s.dye.current.to_numpy()
    assert "clamp_inflow" not in dev.log[mark:]
    assert np.array_equal(got, new)                                      # 1.5 on the inflow cells too: the owed clamp is gone with the old data
    for _ in range(2):
        sim.step(); ref.update()
    out = sim.field_to_numpy()
    for k, e in ref.fields().items():
        assert np.array_equal(out[k], e, equal_nan=True), k
