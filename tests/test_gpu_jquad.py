"""Four Jacobi sweeps per pass (csrc/fs_jquad.h, fs_jacobi_quad_lazy) against the CPU oracle, bit for bit: reference scenes, random
channels with thick obstacles, sweep counts that leave 0 - 3 single sweeps over, uploads (different wall histories in the two
buffers: the updater must fall back to the two-sweep passes), and masks the pass must refuse."""
import numpy as np
import pytest
from test_gpu_rbpair import thick_scene

pytestmark = pytest.mark.gpu


def build(const, mask, n_iter, res=64, scheme="cip", vc=5.0):
    import fs
    from fs.boundary_condition import BoundaryCondition
    from oracle import oracle as O
    dt, dx, re = 0.05 / res, 1.0 / res, 1.0e4
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    pu = fs.JacobiPressureUpdater(bc, dt, dx, n_iter)
    v = fs.VorticityConfinement(bc, dt, dx, vc) if vc else None
    solver = (fs.CipMacSolver(bc, pu, dt, dx, re, v) if scheme == "cip" else
              fs.MacSolver(bc, pu, fs.advect_upwind if scheme == "upwind" else fs.advect_kk_scheme, dt, dx, re, v))
    ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=("jacobi", n_iter))
    return solver, ref, pu


def compare(solver, ref, steps, tag):
    for step in range(1, steps + 1):
        solver.update()
        ref.update()
        for name, a, e in (("v", solver.v.current.to_numpy(), ref.v.current), ("p", solver.p.current.to_numpy(), ref.p.current),
                           ("p.next", solver.p.next.to_numpy(), ref.p.next)):
            assert np.array_equal(a, e, equal_nan=True), f"{tag}: step {step} {name}: {int((a != e).sum())} cells differ"


@pytest.mark.parametrize("X,Y", [(64, 24), (248, 20), (252, 37), (500, 18), (1000, 12), (128, 64)])
@pytest.mark.parametrize("n_iter", [10, 11, 12, 13, 22])
def test_quad_pass_against_the_oracle(X, Y, n_iter, hip_lib, monkeypatch):
    if (X + n_iter) % 2 == 0:
        monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")      # plain / boundary workgroups as two compact launches (large grids do that by themselves)
    rng = np.random.default_rng(X * 7 + Y + n_iter)
    const, mask = thick_scene(rng, X, Y, outflow=(X + n_iter) % 2 == 0)
    solver, ref, pu = build(const, mask, n_iter, scheme=["cip", "upwind", "kk"][n_iter % 3])
    try:
        assert pu._quads and pu.form == "four sweeps per pass", pu.form
        v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
        solver.v.current.from_numpy(v0); ref.v.current[...] = v0
        compare(solver, ref, 4, f"{X}x{Y} n_iter {n_iter}")
        assert float(np.abs(ref.p.current).max()) > 0
    finally:
        solver._dev.close()


@pytest.mark.parametrize("bc,res", [(1, 64), (2, 64), (2, 200), (4, 100), (5, 128), (5, 256), (1, 256)])
def test_reference_scenes(bc, res, hip_lib, monkeypatch):
    from fs.boundary_condition import create_scene_arrays
    if res >= 200:
        monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    const, mask, _ = create_scene_arrays(bc, res)
    solver, ref, pu = build(const, mask, 14, res=res)
    try:
        assert pu._quads, f"scene {bc} at res {res} should admit the four-sweep pass"
        compare(solver, ref, 5, f"bc{bc} res {res}")
    finally:
        solver._dev.close()


def test_uploaded_pressure_falls_back(hip_lib):
    """p.current uploaded (wall cells included): its never-written wall cells now differ from p.next's - the four-sweep pass is skipped
    until ... forever (nothing carries them over), the two-sweep passes run, same bits as the oracle."""
    rng = np.random.default_rng(11)
    X, Y = 128, 40
    const, mask = thick_scene(rng, X, Y)
    solver, ref, pu = build(const, mask, 12)
    try:
        p0 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
        solver.p.current.from_numpy(p0); ref.p.current[...] = p0
        assert pu._quads and solver.p.current.static_id != solver.p.next.static_id
        compare(solver, ref, 4, "uploaded p")
    finally:
        solver._dev.close()


def test_thin_walls_are_refused(hip_lib):
    rng = np.random.default_rng(3)
    const, mask = thick_scene(rng, 64, 32, boxes=0)
    mask[30, 6:20] = 1
    solver, ref, pu = build(const, mask, 12)
    try:
        assert not pu._quads
        compare(solver, ref, 2, "thin wall")
    finally:
        solver._dev.close()
