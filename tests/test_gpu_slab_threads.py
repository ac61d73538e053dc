"""The slab code path of the HIP library on ONE GPU: N slab contexts (y0 > 0, ghost rows, extended row ranges,
per-slab boundary op lists, red-black parity offset) driven by N host threads, with the ghost rows carried between
the contexts by host copies instead of RCCL (a single-GPU box cannot form an RCCL communicator of N ranks).
Everything else - kernels, slab geometry, the validity tracker, the solver orchestration - is the product path.
Results must be bit-identical to the single-domain golden trajectories."""
import os
import threading

import numpy as np
import pytest
from conftest import GOLDEN
from helpers import traj_config

pytestmark = pytest.mark.gpu


def _make_device_cls(world, shared):
    from fs.runtime import Device

    class ThreadSlabDevice(Device):
        """Device whose ghost-row exchange goes through host memory shared by the threads of one process."""

        def __init__(self, nx, ny, dtype, rank, halo):
            super().__init__(nx, ny, dtype, gpu=0, rank=0, nranks=1, halo=0)   # plain context first, replaced below
            self.close()
            # re-run DeviceBase geometry as a slab, then create the context by hand (no RCCL communicator)
            from fs.runtime import DeviceBase
            import ctypes
            from fs import _lib
            DeviceBase.__init__(self, nx, ny, dtype, 0, rank, world, halo, None, None)
            ctx = ctypes.c_void_p()
            _lib.call("fs_create", ctypes.byref(ctx), 0, self.nx, self.ny, 0 if self.dtype == np.float32 else 1,
                      self.y0, self.nyl, self.halo)
            self._ctx = ctx

        def _p_exchange(self, h, nchan, depth):
            self._p_exchange_many([(h, nchan, 0)], depth)

        def _p_exchange_many(self, handles, depth):
            # depth offsets [v, depth) of every field travel (v = rows the tracker still trusts), as in fs_halo_exchange_begin_partial
            H, n = self.halo, self.nyl
            mine = [(self._p_download(h, c, H + v, depth - v), self._p_download(h, c, H + n - depth, depth - v)) if v < depth else None
                    for h, c, v in handles]
            shared["box"][self.rank] = mine
            shared["barrier"].wait()
            for k, (h, c, v) in enumerate(handles):
                if v >= depth:
                    continue
                if self.rank > 0:
                    self._p_upload(h, c, np.ascontiguousarray(shared["box"][self.rank - 1][k][1]), H - depth, depth - v)
                if self.rank < world - 1:
                    self._p_upload(h, c, np.ascontiguousarray(shared["box"][self.rank + 1][k][0]), H + n + v, depth - v)
            shared["barrier"].wait()

        def _p_exchange_begin(self, handles, depth):   # host copies are synchronous: the kernel split is still exercised
            self._p_exchange_many(handles, depth)

        def _p_exchange_wait(self):
            pass

        def _p_exchange_mark(self):
            pass

        def _p_max_over_ranks(self, values):
            shared["radii"][self.rank] = list(values)
            shared["barrier"].wait()
            out = [max(col) for col in zip(*shared["radii"])]
            shared["barrier"].wait()
            return out

    return ThreadSlabDevice


def _worker(rank, world, halo, g, cfg, Dev, results, errors):
    try:
        import fs
        from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition
        dt, dx, re = cfg["dt"], cfg["dx"], cfg["re"]
        X, Y = g["bc_mask"].shape
        dev = Dev(X, Y, np.float64 if cfg["fp64"] else np.float32, rank, halo)
        bc = (DyeBoundaryCondition(g["bc_const"], g["bc_dye"], g["bc_mask"], device=dev) if cfg["dye"]
              else BoundaryCondition(g["bc_const"], g["bc_mask"], device=dev))
        vc = fs.VorticityConfinement(bc, dt, dx, cfg["vor_eps"]) if cfg["vor_eps"] is not None else None
        u = cfg["updater"]
        pu = (fs.RedBlackSorPressureUpdater(bc, dt, dx, u[1], u[2]) if u[0] == "rbsor" else fs.JacobiPressureUpdater(bc, dt, dx, u[1]))
        if cfg["scheme"] == "cip":
            solver = (fs.DyeCipMacSolver if cfg["dye"] else fs.CipMacSolver)(bc, pu, dt, dx, re, vc)
        else:
            adv = fs.advect_upwind if cfg["scheme"] == "upwind" else fs.advect_kk_scheme
            solver = (fs.DyeMacSolver if cfg["dye"] else fs.MacSolver)(bc, pu, adv, dt, dx, re, vc)
        snaps = {}
        for step in range(1, max(cfg["snaps"]) + 1):
            solver.update()
            if step in cfg["snaps"]:
                snaps[step] = [f.to_numpy(local=True) for f in solver.get_fields()]
        results[rank] = (snaps, dev.n_exchanges / max(cfg["snaps"]))
        dev.close()
    except BaseException as e:   # noqa: BLE001 - surface in the main thread
        errors.append((rank, repr(e)))
        try:
            shared_abort = threading.current_thread()._fs_shared
            shared_abort["barrier"].abort()
        except Exception:
            pass


CASES = [
    ("traj_bc5_cip_vc5.npz", 2, 2), ("traj_bc5_cip_vc5.npz", 2, 8), ("traj_bc5_cip_vc5.npz", 3, 4),
    ("traj_cfg5_bc3_res96_kk_vc10_re1e8.npz", 3, 8), ("traj_bc2_cip_jacobi50_vc0.npz", 2, 4),
    ("traj_dye_bc2_cip_vc5.npz", 2, 8), ("traj_bc1_upwind_vc0.npz", 4, 2), ("traj_bc6_res64_cip_vc5_dye.npz", 2, 8),
    ("traj_f64_bc1_cip_vc0.npz", 2, 4),
    # the 8-rank shape of the driver's scaling run: 4 owned rows per slab (halo 4), 12 owned rows (halo 8)
    ("traj_bc5_cip_vc5.npz", 8, 4), ("traj_cfg5_bc3_res96_kk_vc10_re1e8.npz", 8, 8),
]


def _run_slabs(g, cfg, world, halo):
    shared = {"barrier": threading.Barrier(world), "box": [None] * world, "radii": [None] * world}
    Dev = _make_device_cls(world, shared)
    results, errors = [None] * world, []
    threads = []
    for r in range(world):
        t = threading.Thread(target=_worker, args=(r, world, halo, g, cfg, Dev, results, errors))
        t._fs_shared = shared
        threads.append(t)
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    return results


@pytest.mark.parametrize("fname,world,halo", CASES)
def test_slabs_on_one_gpu_are_bit_identical(fname, world, halo, hip_lib):
    g = np.load(os.path.join(GOLDEN, fname))
    cfg = traj_config(g)
    results = _run_slabs(g, cfg, world, halo)
    names = ["v", "p", "dye"]
    for step in cfg["snaps"]:
        for k in range(len(results[0][0][step])):
            full = np.concatenate([results[r][0][step][k] for r in range(world)], axis=1)
            assert np.array_equal(full, g[f"step{step}.{names[k]}"]), f"{fname} step {step} {names[k]}"
    if halo >= 8 and cfg["updater"][0] == "rbsor" and not cfg["dye"]:
        assert results[0][1] <= 6.0


@pytest.mark.parametrize("fname,world,halo", [c for c in CASES if "_cip_" in c[0] and "f64" not in c[0]])
def test_slabs_with_k2_in_registers(fname, world, halo, hip_lib, monkeypatch):
    """fs_cip_step / fs_cip_step_dye on slabs evaluate K2 in registers as on the single-GPU grid (csrc/fs_k234.h; halo >= 3: the call reads 3 rows
    beyond its range) - forced onto the small golden scenes, where a slab's row ranges are a few rows and most tiles are boundary tiles."""
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    g = np.load(os.path.join(GOLDEN, fname))
    cfg = traj_config(g)
    results = _run_slabs(g, cfg, world, halo)
    names = ["v", "p", "dye"]
    for step in cfg["snaps"]:
        for k in range(len(results[0][0][step])):
            full = np.concatenate([results[r][0][step][k] for r in range(world)], axis=1)
            assert np.array_equal(full, g[f"step{step}.{names[k]}"]), f"{fname} step {step} {names[k]}"


@pytest.mark.parametrize("bc,res,scheme,vc,updater,world,halo", [
    (5, 1024, "cip", 5.0, ("rbsor", 1.3, 2), 8, 8),      # configs[2]'s physics, the 8-slab cut of the scaling run
    (5, 4096, "cip", 5.0, ("rbsor", 1.3, 2), 8, 8),      # configs[2] itself, as the driver's --gpus 8 run cuts it
    (2, 1024, "cip", None, ("jacobi", 12), 8, 8),        # configs[3]'s scene (bc2 CIP) with Jacobi
    (3, 1000, "kk", 10.0, ("rbsor", 1.3, 2), 4, 16),     # configs[4]: 4 slabs, 250 rows each (not a multiple of 4)
    (2, 8192, "cip", 5.0, ("rbsor", 1.3, 2), 8, 16),     # configs[3] ITSELF: bc2 res 8192 (16384 x 8192 cells), 8 slabs of 1024 rows, default halo
])
def test_slabs_at_size_equal_single_domain(bc, res, scheme, vc, updater, world, halo, hip_lib):
    """Grids wide enough for many waves per row and many tile rows per slab (the XCD band mapping, the overlapped-wave
    column mapping and the red-black parity offset all engage): N slabs == one domain, bit for bit."""
    import fs
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(bc, res)
    g = {"bc_const": const, "bc_mask": mask}
    steps = 4 if res < 8192 else 3
    cfg = dict(bc=bc, res=res, dt=0.05 / res, dx=1.0 / res, re=1e6, vor_eps=vc, scheme=scheme, updater=updater,
               dye=False, fp64=False, snaps=[steps])
    results = _run_slabs(g, cfg, world, halo)
    fs.runtime.init(gpu=0, dtype="f32")
    one = fs.FluidSimulator.create(bc, res, cfg["dt"], cfg["dx"], cfg["re"], vc, scheme, pressure_updater=updater)
    for _ in range(steps):
        one.step()
    ref = one.field_to_numpy()
    for k, name in enumerate(("v", "p")):
        full = np.concatenate([results[r][0][steps][k] for r in range(world)], axis=1)
        assert np.array_equal(full, ref[name]), name
    assert float(np.abs(ref["p"]).max()) > 0
    one._solver._bc.device.close()


@pytest.mark.parametrize("pairs", ["0", "1", "2"])
@pytest.mark.parametrize("world,halo", [(2, 4), (3, 8)])
def test_long_jacobi_runs_in_slabs(pairs, world, halo, hip_lib, monkeypatch):
    """50 Jacobi sweeps per step cut into slabs, in the three forms the updater can issue (single lazily-bounded sweeps, two-sweep passes,
    two-sweep passes with the vertical recipes in the tiles): the row windows, the general-row list and the 4-row reach of a pass all
    act on slab-local rows here.  Bit-identical to the single-domain golden trajectory."""
    monkeypatch.setenv("FS_JACOBI_PAIRS", pairs)
    fname = "traj_bc2_cip_jacobi50_vc0.npz"
    g = np.load(os.path.join(GOLDEN, fname))
    cfg = traj_config(g)
    results = _run_slabs(g, cfg, world, halo)
    for step in cfg["snaps"]:
        for k, name in enumerate(("v", "p")):
            full = np.concatenate([results[r][0][step][k] for r in range(world)], axis=1)
            assert np.array_equal(full, g[f"step{step}.{name}"]), f"{fname} step {step} {name} pairs={pairs}"


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_stacked_pair_on_slab_ranges_that_are_not_whole_tiles(overlap, hip_lib, monkeypatch):
    """the stacked plain part of the two-part red-black pass (16-row tiles, no mask, stores every row of its tile) ran on slab
    row ranges whose last tile is cut short by row_end - rows past the range were overwritten mask-free (ghost rows, and in overlap mode
    interior rows already computed).  Since round 6 a tile the range cuts short is never listed as plain (csrc/fs_core.hip tile_list).
    3 slabs of 167 / 167 / 166 rows, halo 20, FS_RBPAIR_SPLIT=2: the interior range [40, 167) and the edge strips are no multiples of 16; wall
    blocks and an inflow / outflow patch sit within reach of every range boundary.  Compared with the CPU oracle, bit for bit."""
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    monkeypatch.setenv("FS_OVERLAP", overlap)
    res, world, halo, steps = 500, 3, 20, 4
    const, mask, _ = create_scene_arrays(2, res)
    const, mask = const.copy(), mask.copy()
    X, Y = mask.shape
    from fs.runtime import slab_rows
    for r in range(world):
        y0, nyl = slab_rows(Y, r, world)
        for j in (y0 + 2 * halo - 3, y0 + 2 * halo + 9, y0 + nyl - 2 * halo + 5, y0 + nyl - 7, y0 + 3):
            if 4 <= j < Y - 8:
                i = 150 + 97 * r + (j % 5) * 40
                mask[i:i + 30, j:j + 4] = 1                      # a wall block across the range boundary's reach
                mask[i + 300:i + 306, j - 1:j + 5] = 1           # a narrow one
    g = {"bc_const": const, "bc_mask": mask}
    cfg = dict(bc=2, res=res, dt=0.05 / res, dx=1.0 / res, re=1e6, vor_eps=5.0, scheme="cip", updater=("rbsor", 1.3, 2),
               dye=False, fp64=False, snaps=[steps])
    import fs
    from fs.boundary_condition import BoundaryCondition
    fs.runtime.init(gpu=0, dtype="f32")
    probe = BoundaryCondition(const, mask)
    assert probe.device.rb_pair_ok, "the test's mask must admit the two-iteration pass, or it tests nothing"
    probe.device.close()
    results = _run_slabs(g, cfg, world, halo)
    ref = O.make_simulator(const, mask, None, scheme="cip", dt=cfg["dt"], dx=cfg["dx"], re=cfg["re"], vor_eps=5.0)
    for _ in range(steps):
        ref.update()
    exp = ref.fields()
    for k, name in enumerate(("v", "p")):
        full = np.concatenate([results[r][0][steps][k] for r in range(world)], axis=1)
        assert np.array_equal(full, exp[name], equal_nan=True), (name, overlap)
    assert float(np.abs(exp["p"]).max()) > 0
