"""The two-iteration red-black pass as a row-marching pipeline (csrc/fs_rbmarch.h, the default form of fs_rbsor_pair) against the
register-tile form (csrc/fs_rbpair.h, FS_RBMARCH=0) and against the launch-by-launch iterations: every pressure buffer, wall cells
included, bit for bit - for every strip height / prefetch distance the kernel is built for, grids shorter and taller than a strip,
ragged last strips, odd iteration counts, uploads in mid-run, the reference's scenes and the CPU oracle."""
import numpy as np
import pytest

from test_gpu_rbpair import build, same_pressure, thick_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,PF", [(14, 1), (14, 3), (26, 3), (38, 1), (38, 3), (50, 3)])
@pytest.mark.parametrize("X,Y", [(64, 12), (124, 61), (252, 100), (500, 131), (1000, 40)])
def test_marching_equals_tiles_and_single_iterations(L, PF, X, Y, hip_lib, monkeypatch):
    rng = np.random.default_rng(L * 1000 + PF * 100 + X + Y)
    const, mask = thick_scene(rng, X, Y, boxes=10, outflow=(X + L) % 2 == 0)
    n_iter = 2 + (X + Y + L) % 3
    monkeypatch.setenv("FS_RBM_L", str(L))
    monkeypatch.setenv("FS_RBM_PF", str(PF))
    monkeypatch.setenv("FS_RBMARCH", "1")
    a = build(const, mask, n_iter, True)
    monkeypatch.setenv("FS_RBMARCH", "0")
    b = build(const, mask, n_iter, True)
    c = build(const, mask, n_iter, False)
    try:
        assert a._dev.rb_pair_ok and a.pressure_updater._pair and b.pressure_updater._pair and not c.pressure_updater._pair
        v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
        p0 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
        for s in (a, b, c):
            s.v.current.from_numpy(v0)
            s.p.current.from_numpy(p0)
        for step in range(5):
            if step == 3:
                p1 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
                for s in (a, b, c):
                    s.p.next.from_numpy(p1)
            for s in (a, b, c):
                s.update()
            same_pressure(a, b, f"marching vs tiles, L {L} PF {PF} {X}x{Y} n_iter {n_iter} step {step + 1}")
            same_pressure(a, c, f"marching vs single iterations, L {L} PF {PF} {X}x{Y} n_iter {n_iter} step {step + 1}")
    finally:
        for s in (a, b, c):
            s._dev.close()


@pytest.mark.parametrize("bc,res", [(1, 100), (2, 256), (4, 128), (5, 256), (5, 333)])
def test_marching_on_the_reference_scenes_against_the_oracle(bc, res, hip_lib, monkeypatch):
    import fs
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    monkeypatch.setenv("FS_RBMARCH", "1")
    dt, dx = 0.05 / res, 1.0 / res
    fs.runtime.init(gpu=0, dtype="f32")
    sim = fs.FluidSimulator.create(bc, res, dt, dx, 1e6, 5.0, "cip")
    const, mask, _ = create_scene_arrays(bc, res)
    ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=1e6, vor_eps=5.0)
    try:
        assert sim._solver.pressure_updater._pair
        for _ in range(8):
            sim.step()
            ref.update()
        s = sim._solver
        assert np.array_equal(s.p.current.to_numpy(), ref.p.current, equal_nan=True)
        assert np.array_equal(s.p.next.to_numpy(), ref.p.next, equal_nan=True)
        assert np.array_equal(s.v.current.to_numpy(), ref.v.current, equal_nan=True)
    finally:
        sim._solver._dev.close()
