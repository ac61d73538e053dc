"""Stress the kernels where no reference scene goes: random masks with fluid touching the domain edge (clamped
stencils in x and y, the EDGE instantiations of the fused kernels), one-cell-thin walls (chained boundary-condition
hazards, multiple writers), inflow / outflow cells anywhere.  GPU (all fast paths) vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_scene(rng, X, Y, wall_p, io_p):
    mask = (rng.random((X, Y)) < wall_p).astype(np.uint8)
    # a few thicker blobs so that mirror targets (2nd wall layer) exist
    for _ in range(6):
        i, j, w, h = rng.integers(0, X - 4), rng.integers(0, Y - 4), rng.integers(2, 6), rng.integers(2, 6)
        mask[i:i + w, j:j + h] = 1
    io = rng.random((X, Y))
    mask[(io < io_p) & (mask == 0)] = 2
    mask[(io > 1 - io_p) & (mask == 0)] = 3
    const = np.zeros((X, Y, 2), np.float32)
    const[mask == 2] = rng.uniform(-1, 1, (int((mask == 2).sum()), 2)).astype(np.float32)
    dye = rng.uniform(0, 1, (X, Y, 3)).astype(np.float32)
    return const, mask, dye


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("scheme,vc,updater", [("cip", 5.0, ("rbsor", 1.3, 2)), ("kk", 10.0, ("rbsor", 1.3, 2)),
                                               ("upwind", None, ("jacobi", 6)), ("cip", 5.0, ("jacobi", 10))])
def test_random_mask_trajectory(seed, scheme, vc, updater, hip_lib, monkeypatch):
    import fs
    from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition
    from oracle import oracle as O
    rng = np.random.default_rng(100 + seed)
    X, Y = [(64, 32), (128, 16), (32, 64), (260, 24), (64, 8), (72, 40)][seed]
    const, mask, dye = _random_scene(rng, X, Y, wall_p=[0.05, 0.15, 0.3, 0.1, 0.02, 0.5][seed], io_p=0.03)
    res = 32 if seed % 2 == 0 else 30                      # power-of-two dx (exact-reciprocal path) and not
    dt, dx, re = 0.05 / res, 1.0 / res, 1000.0
    with_dye = seed % 3 == 0
    if seed % 2 == 1:
        monkeypatch.setenv("FS_FUSE_TRANSPORT", "0")        # the reference's two launches for K3 / K4 on the odd seeds
    fs.runtime.init(gpu=0, dtype="f32")
    bc = DyeBoundaryCondition(const, dye, mask) if with_dye else BoundaryCondition(const, mask)
    vcobj = fs.VorticityConfinement(bc, dt, dx, vc) if vc is not None else None
    pu = (fs.RedBlackSorPressureUpdater(bc, dt, dx, updater[1], updater[2]) if updater[0] == "rbsor"
          else fs.JacobiPressureUpdater(bc, dt, dx, updater[1]))
    if scheme == "cip":
        solver = (fs.DyeCipMacSolver if with_dye else fs.CipMacSolver)(bc, pu, dt, dx, re, vcobj)
    else:
        adv = fs.advect_upwind if scheme == "upwind" else fs.advect_kk_scheme
        solver = (fs.DyeMacSolver if with_dye else fs.MacSolver)(bc, pu, adv, dt, dx, re, vcobj)
    ref = O.make_simulator(const, mask, dye if with_dye else None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=updater)
    # random initial state on both sides (fluid touching the edges makes the clamped reads matter immediately)
    v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
    p0 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
    solver.v.current.from_numpy(v0); ref.v.current[...] = v0
    solver.p.current.from_numpy(p0); ref.p.current[...] = p0
    try:
        for step in range(6):
            solver.update()
            ref.update()
            got = [f.to_numpy() for f in solver.get_fields()]
            exp = list(ref.fields().values())
            for a, e, name in zip(got, exp, ("v", "p", "dye")):
                assert np.array_equal(a, e, equal_nan=True), f"seed {seed} {scheme} step {step + 1} {name}: {np.nanmax(np.abs(a - e))}"
    finally:
        bc.device.close()


# widths around the mapping boundaries of the fast paths: X % 4 != 0 (generic one-cell-per-lane kernels), exactly one
# overlapped wave (62 quads = 248 cells), one quad more / less, two waves, a 4-wave block (992) and one quad beyond it
RAGGED = [(6, 6), (10, 12), (50, 20), (130, 10), (244, 9), (248, 12), (252, 12), (496, 10), (500, 7), (992, 6), (996, 6), (1240, 5)]


@pytest.mark.parametrize("X,Y", RAGGED)
@pytest.mark.parametrize("scheme,vc,updater,with_dye", [("cip", 5.0, ("rbsor", 1.3, 2), True), ("kk", 10.0, ("jacobi", 10), False)])
def test_ragged_widths(X, Y, scheme, vc, updater, with_dye, hip_lib):
    import fs
    from fs.boundary_condition import BoundaryCondition, DyeBoundaryCondition
    from oracle import oracle as O
    rng = np.random.default_rng(X * 1000 + Y)
    const, mask, dye = _random_scene(rng, X, Y, wall_p=0.08, io_p=0.03) if min(X, Y) > 8 else (
        np.zeros((X, Y, 2), np.float32), (rng.random((X, Y)) < 0.2).astype(np.uint8), rng.uniform(0, 1, (X, Y, 3)).astype(np.float32))
    res = 64
    dt, dx, re = 0.05 / res, 1.0 / res, 1000.0
    fs.runtime.init(gpu=0, dtype="f32")
    bc = DyeBoundaryCondition(const, dye, mask) if with_dye else BoundaryCondition(const, mask)
    vcobj = fs.VorticityConfinement(bc, dt, dx, vc)
    pu = (fs.RedBlackSorPressureUpdater(bc, dt, dx, updater[1], updater[2]) if updater[0] == "rbsor"
          else fs.JacobiPressureUpdater(bc, dt, dx, updater[1]))
    if scheme == "cip":
        solver = (fs.DyeCipMacSolver if with_dye else fs.CipMacSolver)(bc, pu, dt, dx, re, vcobj)
    else:
        solver = (fs.DyeMacSolver if with_dye else fs.MacSolver)(bc, pu, fs.advect_kk_scheme, dt, dx, re, vcobj)
    ref = O.make_simulator(const, mask, dye if with_dye else None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=updater)
    v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
    p0 = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
    solver.v.current.from_numpy(v0); ref.v.current[...] = v0
    solver.p.current.from_numpy(p0); ref.p.current[...] = p0
    try:
        for step in range(3):
            solver.update()
            ref.update()
            for a, e, name in zip([f.to_numpy() for f in solver.get_fields()], list(ref.fields().values()), ("v", "p", "dye")):
                assert np.array_equal(a, e, equal_nan=True), f"{X}x{Y} {scheme} step {step + 1} {name}"
    finally:
        bc.device.close()
