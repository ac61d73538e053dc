"""GPU parity of fs_cip_step (K2 + K3 + K4 of the CIP velocity step as one call; csrc/fs_k234.h): on large single-GPU grids the tiles that
see nothing but fluid evaluate K2 in registers and hand the advecting component to the sibling wave through LDS.

Bar: bit-exact.  (a) against the two calls it replaces (fs_cip_nonadv + fs_cip_grad_advect, themselves pinned to the golden vectors and
the oracle) on seeded random fields; (b) whole trajectories against the CPU oracle; in both cases with the three-part launch forced onto
grids small enough for the oracle (FS_RBPAIR_SPLIT=2).  What is allowed to differ is the intermediate buffer: its FLUID cells outside
the band the boundary tiles read keep their old content (the reference overwrites them before reading them) - checked as such."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(bc, res):
    from fs.boundary_condition import create_scene_arrays
    return create_scene_arrays(bc, res)


def _random_mask(X, Y, seed):
    """wall ring of 2 cells, inflow columns 0-1, outflow columns X-2, X-1, random rectangles of wall at least 2 cells thick"""
    rng = np.random.default_rng(seed)
    m = np.zeros((X, Y), np.uint8)
    m[:, :2] = 1; m[:, -2:] = 1
    m[:2, 2:-2] = 2; m[-2:, 2:-2] = 3
    for _ in range(rng.integers(2, 6)):
        w, h = rng.integers(2, max(3, X // 12)), rng.integers(2, max(3, Y // 6))
        i, j = rng.integers(4, X - w - 4), rng.integers(4, Y - h - 4)
        m[i:i + w, j:j + h] = 1
    const = np.zeros((X, Y, 2), np.float32)
    const[m == 2, 0] = 1.0
    return const, m


# (scene, res / random mask, needs_register_tiles): the last flag asserts that some all-fluid tile lies a whole tile away from every boundary
# tile, i.e. that the case really runs K2 in registers without a stored copy next to it
CASES = [
    pytest.param(("scene", 2, 512, True), id="bc2-res512"),
    pytest.param(("scene", 1, 400, True), id="bc1-res400-division-path"),      # dx = 1/400 is not a power of two: f64-multiply divisions
    pytest.param(("scene", 5, 512, False), id="bc5-res512"),
    pytest.param(("scene", 3, 512, False), id="bc3-res512"),
    pytest.param(("random", 1200, 96, 7, True), id="random-1200x96"),
    pytest.param(("random", 840, 250, 11, True), id="random-840x250"),
]


def _build(case):
    if case[0] == "scene":
        const, mask, _ = _scene(case[1], case[2])
        res = case[2]
    else:
        const, mask = _random_mask(case[1], case[2], case[3])
        res = case[2]
    return const, mask, res, case[-1]


# FS_FUSE_K2: 2 (default) - K2 in registers on every tile, ONE launch over both kinds of tile; 1 - one launch per kind
@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("case", CASES)
def test_one_call_equals_the_two_calls(case, mode, hip_lib, monkeypatch):
    import fs
    from fs.boundary_condition import BoundaryCondition
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")          # the multi-part launch on a grid of any size
    monkeypatch.setenv("FS_FUSE_K2", str(mode))
    const, mask, res, needs_register_tiles = _build(case)
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    dev = bc.device
    try:
        assert dev.cip_step_fused
        X, Y = mask.shape
        dt, dx, re = 0.05 / res, 1.0 / res, 1.0e6
        rng = np.random.default_rng(1234)
        arr = lambda c, s: rng.uniform(-s, s, (X, Y, c) if c > 1 else (X, Y)).astype(np.float32)
        fc_h, pc_h, gx_h, gy_h, stale = arr(2, 1.0), arr(1, 10.0), arr(2, 50.0), arr(2, 50.0), arr(2, 1.0)
        names = ("out", "gxo", "gyo", "fn")
        res_ = {}
        for form in ("two", "one"):
            fc, pc, gx, gy = dev.alloc(2), dev.alloc(1), dev.alloc(2), dev.alloc(2)
            out, gxo, gyo, fn = dev.alloc(2), dev.alloc(2), dev.alloc(2), dev.alloc(2)
            for f, h in ((fc, fc_h), (pc, pc_h), (gx, gx_h), (gy, gy_h), (fn, stale), (out, fc_h), (gxo, gx_h), (gyo, gy_h)):
                f.from_numpy(h)
            dev.profile(True); dev.profile_reset()
            if form == "two":
                dev.cip_nonadv(dt, dx, re, fn, fc, pc)
                dev.cip_grad_advect(dt, dx, out, gxo, gyo, fn, fc, gx, gy)
            else:
                dev.cip_step(dt, dx, re, out, gxo, gyo, fn, fc, pc, gx, gy)
            kernels = set(dev.profile_report())
            dev.profile(False)
            if form == "one":
                assert "cip_step" in kernels and ("cip_step_bnd" in kernels) == (mode == 1) and "cip_step_band" not in kernels, kernels
            res_[form] = {n: f.to_numpy() for n, f in zip(names, (out, gxo, gyo, fn))}
        for n in ("out", "gxo", "gyo"):
            assert np.array_equal(res_["one"][n], res_["two"][n], equal_nan=True), n
        # the intermediate buffer: equal wherever it was written, and written at least on every cell that is not fluid
        a, b = res_["one"]["fn"], res_["two"]["fn"]
        written = ~np.all(a == stale, axis=2)
        assert np.array_equal(a[written], b[written])
        assert np.array_equal(a[mask != 0], b[mask != 0])
        if needs_register_tiles:
            assert (~written & (mask == 0)).any(), "every fluid cell was stored: no tile of this case runs without a stored copy of K2 next to it"
    finally:
        dev.close()


@pytest.mark.parametrize("bc,res,vc,steps,mode", [(2, 512, 5.0, 8, 2), (2, 512, None, 8, 2), (1, 400, 5.0, 8, 2), (5, 512, 5.0, 6, 2), (3, 512, None, 6, 2), (4, 512, 5.0, 6, 2),
                                                  (2, 512, 5.0, 8, 1), (1, 400, 5.0, 8, 1), (5, 512, 5.0, 6, 1), (3, 512, None, 6, 1)])
def test_trajectory_against_the_oracle(bc, res, vc, steps, mode, hip_lib, monkeypatch):
    import fs
    from oracle import oracle as O
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    monkeypatch.setenv("FS_FUSE_K2", str(mode))
    dt, dx, re = 0.05 / res, 1.0 / res, 1.0e6
    fs.runtime.init(gpu=0, dtype="f32")
    sim = fs.FluidSimulator.create(bc, res, dt, dx, re, vc, "cip")
    try:
        dev = sim._solver._bc.device
        assert sim._solver._fused_k2 and dev.cip_step_fused
        const, mask, _ = _scene(bc, res)
        ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=re, vor_eps=vc)
        dev.profile(True)
        for _ in range(steps):
            sim.step()
            ref.update()
        assert "cip_step" in dev.profile_report()
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e), k
        s = sim._solver
        for name in ("vx", "vy", "p"):
            assert np.array_equal(getattr(s, name).current.to_numpy(), getattr(ref, name).current), name
        # v.next: with vorticity confinement it is the advected velocity (fully reproduced); without, the post-K2 buffer - reproduced on
        # every cell that is not fluid
        a, e = s.v.next.to_numpy(), ref.v.next
        if vc is not None:
            assert np.array_equal(a, e)
        else:
            assert np.array_equal(a[mask != 0], e[mask != 0])
    finally:
        sim._solver._bc.device.close()


def test_hipgraph_replay_of_the_three_part_step(hip_lib, monkeypatch):
    """The period captured as a hipGraph replays the three launches with their lists: same bits as eager stepping."""
    import fs
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    res = 256
    fs.runtime.init(gpu=0, dtype="f32")
    mk = lambda: fs.FluidSimulator.create(5, res, 0.05 / res, 1.0 / res, 1.0e6, 5.0, "cip")
    a, b = mk(), mk()
    try:
        for _ in range(3):
            a.step(); b.step()
        a.run(24)
        for _ in range(24):
            b.step()
        fa, fb = a.field_to_numpy(), b.field_to_numpy()
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
    finally:
        a._solver._bc.device.close()
        b._solver._bc.device.close()


@pytest.mark.parametrize("bc,res,steps,mode", [(2, 512, 6, 2), (5, 512, 5, 2), (1, 400, 6, 2), (3, 512, 5, 2), (4, 512, 5, 2), (2, 512, 6, 1), (5, 512, 5, 1), (1, 400, 6, 1)])
def test_dye_trajectory_against_the_oracle(bc, res, steps, mode, hip_lib, monkeypatch):
    """fs_cip_step_dye: K12 in registers on the all-fluid tiles of the dye's step (k_cip_dye_plain), one wave per tile and channel."""
    import fs
    from oracle import oracle as O
    monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    monkeypatch.setenv("FS_FUSE_K2", str(mode))
    dt, dx, re = 0.05 / res, 1.0 / res, 1.0e6
    fs.runtime.init(gpu=0, dtype="f32")
    sim = fs.DyeFluidSimulator.create(bc, res, dt, dx, re, 5.0, "cip")
    try:
        dev = sim._solver._bc.device
        assert sim._solver._fused_k2 and sim._solver._fused_dye and dev.cip_step_fused
        const, mask, dye = _scene(bc, res)
        ref = O.make_simulator(const, mask, dye, scheme="cip", dt=dt, dx=dx, re=re, vor_eps=5.0)
        dev.profile(True)
        for _ in range(steps):
            sim.step()
            ref.update()
        rep = dev.profile_report()
        assert "cip_step_dye" in rep and ("cip_step_dye_bnd" in rep) == (mode == 1) and "cip_nonadv_dye" not in rep and "cip_step_dye_band" not in rep, sorted(rep)
        out = sim.field_to_numpy()
        for k, e in ref.fields().items():
            assert np.array_equal(out[k], e), k
        s = sim._solver
        for name in ("dyex", "dyey"):
            assert np.array_equal(getattr(s, name).current.to_numpy(), getattr(ref, name).current), name
        a, e = s.dye.next.to_numpy(), ref.dye.next
        assert np.array_equal(a[mask != 0], e[mask != 0])
    finally:
        sim._solver._bc.device.close()
