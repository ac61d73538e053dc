"""S lazily-bounded Jacobi sweeps per pass as a row-marching pipeline (csrc/fs_jmarch.h, fs_jacobi_march) against the CPU oracle, bit for
bit: 4 / 6 / 8 sweeps per pass, every strip height the launcher may pick (shorter and taller than the grid, ragged last strips), both
prefetch distances, sweep counts that leave single sweeps over, the reference's scenes, uploads in mid-run (the updater must fall back)."""
import numpy as np
import pytest
from test_gpu_jquad import build, compare
from test_gpu_rbpair import thick_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_form():
    """(overrides the autouse fixture imported with test_gpu_jquad's helpers: the marching form is what runs here)"""
    yield


@pytest.mark.parametrize("S,L,PF", [(4, 4, 1), (4, 16, 1), (4, 28, 3), (6, 12, 1), (6, 24, 3), (8, 8, 1), (8, 20, 1), (4, 0, 1), (8, 0, 1)])
@pytest.mark.parametrize("X,Y,n_iter", [(64, 24, 10), (248, 61, 13), (252, 100, 22), (500, 37, 19), (1000, 131, 12)])
def test_marching_passes_against_the_oracle(S, L, PF, X, Y, n_iter, hip_lib, monkeypatch):
    monkeypatch.setenv("FS_JACOBI_MARCH", str(S))
    if L:
        monkeypatch.setenv("FS_JM_L", str(L))
    monkeypatch.setenv("FS_JM_PF", str(PF))
    rng = np.random.default_rng(S * 1000 + L * 10 + X + Y)
    const, mask = thick_scene(rng, X, Y, boxes=10, outflow=(X + S) % 2 == 0)
    solver, ref, pu = build(const, mask, n_iter, scheme=["cip", "upwind", "kk"][(n_iter + S) % 3])
    try:
        assert pu._march == S and "row-marching" in pu.form, pu.form
        v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
        solver.v.current.from_numpy(v0); ref.v.current[...] = v0
        compare(solver, ref, 3, f"S {S} L {L} PF {PF} {X}x{Y} n_iter {n_iter}")
        assert float(np.abs(ref.p.current).max()) > 0
    finally:
        solver._dev.close()


@pytest.mark.parametrize("S", [4, 6, 8])
@pytest.mark.parametrize("bc,res", [(1, 64), (2, 200), (4, 100), (5, 256), (1, 333)])
def test_reference_scenes(S, bc, res, hip_lib, monkeypatch):
    from fs.boundary_condition import create_scene_arrays
    monkeypatch.setenv("FS_JACOBI_MARCH", str(S))
    const, mask, _ = create_scene_arrays(bc, res)
    solver, ref, pu = build(const, mask, 20, res=res)
    try:
        if res % 2:       # (X = 2 res not a multiple of 4: the multi-sweep passes are not admitted - the updater keeps its other forms, same bits)
            assert pu._march == 0
        else:
            assert pu._march == S
        compare(solver, ref, 4, f"S {S} bc{bc} res {res}")
    finally:
        solver._dev.close()


def test_uploaded_pressure_falls_back(hip_lib, monkeypatch):
    monkeypatch.setenv("FS_JACOBI_MARCH", "8")
    rng = np.random.default_rng(5)
    X, Y = 128, 70
    const, mask = thick_scene(rng, X, Y)
    solver, ref, pu = build(const, mask, 20)
    try:
        compare(solver, ref, 2, "before the upload")
        p0 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
        solver.p.current.from_numpy(p0); ref.p.current[...] = p0
        assert pu._march == 8 and solver.p.current.static_id != solver.p.next.static_id
        compare(solver, ref, 3, "after the upload")
    finally:
        solver._dev.close()
