"""Overlapped ghost-row exchange through the REAL RCCL path on one GPU.

A 1-rank communicator in loop-back mode (fs_comm_loopback: the rank is its own lower and upper neighbour) carries the
exchanges of a slab that the tracker believes to be the middle one of three.  The ghost rows then mirror the slab's own
edge rows - physically meaningless, but fully deterministic - so the run with the exchange hidden behind the interior rows
of the kernel that needed it (communication stream + events, kernel split into interior and two strips) must equal the run
with blocking exchanges bit for bit; a missing stream dependency shows up as a difference."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_cls():
    from fs import _lib
    from fs.runtime import Device, DeviceBase

    class LoopbackSlab(Device):
        def __init__(self, nx, ny, halo, overlap):
            DeviceBase.__init__(self, nx, ny, np.float32, 0, 1, 3, halo, None, None)      # tracker: middle slab of three
            self.overlap = overlap
            self._lib = _lib.load()
            ctx = ctypes.c_void_p()
            _lib.call("fs_create", ctypes.byref(ctx), 0, self.nx, self.ny, 0, self.y0, self.nyl, self.halo)
            self._ctx = ctx
            self._graphs = []
            uid = ctypes.create_string_buffer(128)
            _lib.call("fs_comm_unique_id", uid)
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                _lib.call("fs_comm_init", ctx, 0, 1, ctypes.c_char_p(uid.raw))
            finally:
                ctypes.CDLL(None).fflush(None)
                os.dup2(saved, 1)
                os.close(saved)
            _lib.call("fs_comm_loopback", ctx, 1)
            self._has_comm = True

        def _p_max_over_ranks(self, values):
            return list(values)

    return LoopbackSlab


def _run(res, halo, overlap, scheme, updater, steps, tape=False):
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    const, mask, _ = create_scene_arrays(5, res)
    os.environ["FS_OVERLAP"] = "1" if overlap else "0"      # read by fs_comm_init: exchanges on the communication stream / in line
    dev = _device_cls()(mask.shape[0], mask.shape[1], halo, overlap)
    os.environ.pop("FS_OVERLAP")
    dt, dx = 0.05 / res, 1.0 / res
    bc = BoundaryCondition(const, mask, device=dev)
    vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2) if updater == "rbsor" else fs.JacobiPressureUpdater(bc, dt, dx, 12)
    if scheme == "cip":
        solver = fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc)
    else:
        solver = fs.MacSolver(bc, pu, fs.advect_kk_scheme, dt, dx, 1e6, vc)
    if tape is not False:
        # the N > 1 timed loop of bench.py: log the period of the step, compile it into a C++ tape (fs_tape_*), replay it
        done = [0]

        def counted():
            solver.update()
            done[0] += 1
        t = dev.tape_period(counted, nsteps=2)
        assert t is not None and t["id"] is not None, "no steady period / no native tape"
        dev.replay_tape(t, 3)
        steps = done[0] + 3 * t["nsteps"] + 1
        solver.update()                   # the eager path carries on from the replayed state
        tape.append((steps, len(t["ops"]), t["nsteps"]))
    else:
        for _ in range(steps):
            solver.update()
    out = {n: getattr(solver, n).current.local_window() for n in ("v", "p", "vx", "vy") if hasattr(solver, n)}
    out.update({n + ".next": getattr(solver, n).next.local_window() for n in ("v", "p") if hasattr(solver, n)})
    stats = (dev.n_exchanges, dev.n_overlapped)
    dev.close()
    return out, stats


@pytest.mark.parametrize("res,halo,scheme,updater,steps", [
    (256, 8, "cip", "rbsor", 12), (256, 2, "cip", "rbsor", 8), (1024, 8, "cip", "rbsor", 10), (1024, 4, "kk", "jacobi", 6),
    (4096, 8, "cip", "rbsor", 6),
])
def test_overlapped_equals_blocking(res, halo, scheme, updater, steps, hip_lib):
    blocking, (nb, ob) = _run(res, halo, False, scheme, updater, steps)
    for rep in range(2):        # twice: races are timing dependent
        hidden, (nh, oh) = _run(res, halo, True, scheme, updater, steps)
        assert ob == 0 and oh > 0 and nh == nb
        for k in blocking:
            assert np.array_equal(hidden[k], blocking[k], equal_nan=True), (k, rep)
    assert float(np.nanmax(np.abs(blocking["p"]))) > 0


@pytest.mark.parametrize("res,halo,scheme,updater", [(256, 8, "cip", "rbsor"), (256, 2, "cip", "rbsor"), (1024, 16, "cip", "rbsor"),
                                                     (1024, 4, "kk", "jacobi"), (4096, 16, "cip", "rbsor")])
def test_tape_replay_equals_eager(res, halo, scheme, updater, hip_lib):
    """fs_tape_*: the recorded period (kernel launches + RCCL exchange begin / wait as C++ closures) replayed without Python
    advances the slab exactly like eager stepping through the same RCCL path."""
    info = []
    taped, (nt, _) = _run(res, halo, False, scheme, updater, 0, tape=info)
    steps, nops, period = info[0]
    eager, (ne, _) = _run(res, halo, False, scheme, updater, steps)
    assert nt == ne, "the replay must account for the exchanges it issues"
    for k in eager:
        assert np.array_equal(taped[k], eager[k], equal_nan=True), (k, steps, nops, period)
    assert float(np.nanmax(np.abs(eager["p"]))) > 0


def test_the_run_chooses_its_exchange_mode(hip_lib):
    """bench.py's N > 1 start (DeviceBase.choose_exchange_mode): the period is recorded and replayed in line and on the communication stream,
    one of the two modes stays (the timing decides; either is correct) and the state afterwards is the state of eager stepping through the same number of steps."""
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    res, halo = 1024, 16
    const, mask, _ = create_scene_arrays(5, res)
    dt, dx = 0.05 / res, 1.0 / res

    def build():
        dev = _device_cls()(mask.shape[0], mask.shape[1], halo, False)
        bc = BoundaryCondition(const, mask, device=dev)
        solver = fs.CipMacSolver(bc, fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2), dt, dx, 1e6, fs.VorticityConfinement(bc, dt, dx, 5.0))
        return dev, solver
    dev, solver = build()
    done = [0]

    def counted():
        solver.update()
        done[0] += 1
    tape, rep = dev.choose_exchange_mode(counted, record_tries=20, trial_steps=48)
    assert tape is not None and rep["in_line_us_per_step"] and rep["overlapped_us_per_step"], rep
    # WHICH mode wins is a measurement (loop-back: 119 against 139 us per step at the headline size), not a property of the code: the test
    # asserts only that one of the two was kept and that the device's settings are those of the mode the report names
    mode = rep["chosen"].split()[0]
    assert mode in ("in", "overlapped"), rep
    assert bool(dev.overlap) == bool(dev.overlap_stream) == (mode == "overlapped"), (rep, dev.overlap, dev.overlap_stream)
    dev.replay_tape(tape, 2)
    steps = done[0] + rep["replayed_steps"] + 2 * tape["nsteps"]
    got = {n: getattr(solver, n).current.local_window() for n in ("v", "p", "vx", "vy")}
    dev.close()
    dev2, solver2 = build()
    for _ in range(steps):
        solver2.update()
    for n, a in got.items():
        assert np.array_equal(a, getattr(solver2, n).current.local_window(), equal_nan=True), n
    dev2.close()
