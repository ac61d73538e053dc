"""The "lazy" pressure boundary condition of long Jacobi runs (csrc/fs_march.h k_jacobi_lazy): all but the last two sweeps of
JacobiPressureUpdater.update evaluate K7 (fs/boundary_condition.py:41-65) on the fly from the raw output of the previous sweep
instead of launching the boundary kernel.  Bit for bit against the CPU oracle, which runs the reference's n x (BC, sweep, swap) -
including the internal p.next buffer, whose non-fluid cells only the last two real boundary passes define."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "2"], ids=["pairs", "pairs+vertical"])
def _force_pairs(request, monkeypatch):
    """Two sweeps per launch wherever the updater may use them, in both kernel variants (by default it times the forms on the mask and
    keeps the fastest)."""
    monkeypatch.setenv("FS_JACOBI_PAIRS", request.param)
    return request.param


def _pair(const, mask, scheme, n_iter, res, dtype="f32", lazy=True, vc=5.0):
    import fs
    from fs.boundary_condition import BoundaryCondition
    from oracle import oracle as O
    dt, dx, re = 0.05 / res, 1.0 / res, 1.0e4
    fs.runtime.init(gpu=0, dtype=dtype)
    bc = BoundaryCondition(const, mask)
    pu = fs.JacobiPressureUpdater(bc, dt, dx, n_iter, precompute_source=True, lazy_bc=lazy)
    v = fs.VorticityConfinement(bc, dt, dx, vc) if vc else None
    if scheme == "cip":
        solver = fs.CipMacSolver(bc, pu, dt, dx, re, v)
    else:
        solver = fs.MacSolver(bc, pu, fs.advect_upwind if scheme == "upwind" else fs.advect_kk_scheme, dt, dx, re, v)
    ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=("jacobi", n_iter),
                           dtype=np.float32 if dtype == "f32" else np.float64)
    return solver, ref, pu


def _run(solver, ref, steps, tag, rng=None):
    if rng is not None:
        X, Y = solver.resolution
        t = solver.v.current.dev.dtype
        v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(t)
        p0 = rng.uniform(-3, 3, (X, Y)).astype(t)
        p1 = rng.uniform(-3, 3, (X, Y)).astype(t)              # the stale buffer matters too (never-written wall cells)
        solver.v.current.from_numpy(v0); ref.v.current[...] = v0
        solver.p.current.from_numpy(p0); ref.p.current[...] = p0
        solver.p.next.from_numpy(p1); ref.p.next[...] = p1
    for step in range(1, steps + 1):
        solver.update()
        ref.update()
        for name, a, e in (("v", solver.v.current.to_numpy(), ref.v.current), ("p", solver.p.current.to_numpy(), ref.p.current),
                           ("p.next", solver.p.next.to_numpy(), ref.p.next)):
            assert np.array_equal(a, e, equal_nan=True), f"{tag}: step {step} {name}: max|d| {np.nanmax(np.abs(a - e))}"


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("res,scheme,n_iter", [(64, "cip", 7), (128, "upwind", 6), (32, "kk", 3)])
def test_reference_scenes(n, res, scheme, n_iter, hip_lib):
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(n, res)
    solver, ref, pu = _pair(const, mask, scheme, n_iter, res)
    try:
        assert solver._dev.lazy_bc_ok and pu._lazy, "every reference scene admits the lazy boundary condition"
        _run(solver, ref, 4, f"bc{n} res{res} {scheme}", np.random.default_rng(n * 100 + res))
    finally:
        solver._dev.close()


def _framed_scene(rng, X, Y, wall_p, io_inside):
    """Walls on rows 0, 1, Y-2, Y-1; inflow columns 0-1, outflow column(s) at the right edge; random interior walls (thin ones
    included), optionally stray inflow / outflow cells inside (an inflow cell left of a wall makes the mask NOT admit laziness)."""
    mask = (rng.random((X, Y)) < wall_p).astype(np.uint8)
    for _ in range(5):
        i, j, w, h = rng.integers(2, X - 6), rng.integers(2, Y - 6), rng.integers(1, 7), rng.integers(1, 7)
        mask[i:i + w, j:j + h] = 1
    if io_inside:
        io = rng.random((X, Y))
        mask[(io < 0.02) & (mask == 0)] = 2
        mask[(io > 0.98) & (mask == 0)] = 3
    mask[:2, :] = 2
    if not io_inside:
        mask[2, :] = 0                       # an inflow cell left of a wall makes the mask refuse laziness
    mask[-(1 + int(rng.integers(0, 2))):, :] = 3
    mask[:, [0, 1, Y - 2, Y - 1]] = 1
    const = np.zeros((X, Y, 2), np.float32)
    const[mask == 2] = rng.uniform(0.2, 1, (int((mask == 2).sum()), 2)).astype(np.float32) * np.float32([1, 0.1])
    return const, mask


@pytest.mark.parametrize("seed", range(10))
def test_random_framed_masks(seed, hip_lib):
    rng = np.random.default_rng(7000 + seed)
    X, Y = [(64, 32), (248, 20), (252, 24), (496, 12), (1000, 10), (128, 64), (72, 40), (244, 16), (992, 9), (1240, 12)][seed]
    const, mask = _framed_scene(rng, X, Y, wall_p=[0.0, 0.03, 0.1, 0.05, 0.02, 0.2, 0.3, 0.08, 0.04, 0.06][seed], io_inside=seed % 3 == 2)
    solver, ref, pu = _pair(const, mask, ["cip", "upwind", "kk"][seed % 3], [3, 6, 7, 8, 10, 11, 6, 9, 4, 14][seed], 32 if seed % 2 else 30)
    try:
        if seed % 3 != 2:
            assert pu._lazy, "a framed mask whose only inflow cells are columns 0-1 admits the lazy boundary condition"
        _run(solver, ref, 3, f"seed {seed} lazy={pu._lazy}", rng)
    finally:
        solver._dev.close()


def test_f64_and_long_run(hip_lib):
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(2, 100)
    solver, ref, pu = _pair(const, mask, "cip", 50, 100, dtype="f64")
    try:
        assert pu._lazy
        _run(solver, ref, 3, "f64 bc2 res100 jacobi50")
    finally:
        solver._dev.close()


def test_inflow_next_to_a_wall_is_refused(hip_lib):
    """An inflow cell whose right neighbour is a wall would make K7 read that buffer's history: fs_lazy_bc_ok says no, the updater
    keeps launching the boundary kernel, results stay those of the oracle."""
    from fs.boundary_condition import create_scene_arrays
    const, mask, _ = create_scene_arrays(1, 32)
    mask = mask.copy()
    mask[2, 10:14] = 1                       # a wall stub right of the inflow column 1
    solver, ref, pu = _pair(const, mask, "cip", 6, 32)
    try:
        assert not solver._dev.lazy_bc_ok and not pu._lazy
        _run(solver, ref, 3, "refused", np.random.default_rng(3))
    finally:
        solver._dev.close()


@pytest.mark.parametrize("seed", [1, 3, 5, 6, 7])
def test_pair_equals_two_sweeps(seed, hip_lib, monkeypatch, _force_pairs):
    """k_jacobi_pair (two sweeps per pass; 3-row tiles, 2 rows with vertical recipes in the tile path) == two k_jacobi_lazy passes == two (K7, sweep) rounds, bit for bit, from random
    iterates - including a random INTERMEDIATE buffer, whose never-written wall cells the second sweep reads."""
    import fs
    from fs.boundary_condition import BoundaryCondition
    rng = np.random.default_rng(9100 + seed)
    X, Y = [(64, 32), (248, 20), (252, 24), (496, 12), (1000, 10), (128, 64), (72, 40), (244, 16)][seed]
    const, mask = _framed_scene(rng, X, Y, wall_p=[0.0, 0.03, 0.1, 0.05, 0.02, 0.2, 0.3, 0.08][seed], io_inside=False)
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    dev = bc.device
    try:
        assert dev.lazy_bc_ok
        a0 = rng.uniform(-3, 3, (X, Y)).astype(np.float32)
        b0 = rng.uniform(-3, 3, (X, Y)).astype(np.float32)
        s0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
        src = dev.alloc(2); src.from_numpy(s0)
        out = []
        for mode in ("pair", "lazy", "real"):
            a, b = dev.alloc(1), dev.alloc(1)
            a.from_numpy(a0); b.from_numpy(b0)
            if mode == "pair":                    # a -> b -> a: the second pass has the buffers the other way round
                dev.jacobi_pair_lazy(b, a, src, swapped=False, vertical=_force_pairs == "2")
                two = b.to_numpy()
                dev.jacobi_pair_lazy(a, b, src, swapped=True, vertical=_force_pairs == "2")
            elif mode == "lazy":
                for k in range(4):
                    dev.jacobi_sweep_lazy(b, a, src)
                    a, b = b, a
                    if k == 1:
                        two = a.to_numpy()
            else:
                for k in range(4):
                    bc.set_pressure_boundary_condition(a)
                    dev.jacobi_sweep_src(b, a, src)
                    a, b = b, a
                    if k == 1:
                        two = a.to_numpy()
            out.append((two[mask != 1], a.to_numpy()[mask != 1]))     # the three differ in which wall cells they touch, by design
        for k in (0, 1):
            assert np.array_equal(out[0][k], out[2][k]) and np.array_equal(out[1][k], out[2][k]), ("after 2 sweeps", "after 4 sweeps")[k]
    finally:
        dev.close()


def test_form_is_decided_from_the_mask(hip_lib, monkeypatch):
    """FS_JACOBI_PAIRS unset (four-sweep passes off - tests/test_gpu_jquad.py covers them): two sweeps per pass; the variant with vertical recipes in the tiles only where the plain one would send more
    than 5 % of the rows down its general path (scene 3's cylinders) - decided from the mask, the same on every run; the oracle's bits."""
    from fs.boundary_condition import create_scene_arrays
    monkeypatch.delenv("FS_JACOBI_PAIRS")
    monkeypatch.setenv("FS_JACOBI_QUADS", "0")
    for bc, res, vertical in ((2, 512, False), (3, 256, None)):
        const, mask, _ = create_scene_arrays(bc, res)
        solver, ref, pu = _pair(const, mask, "cip", 12, res, lazy=None)
        try:
            assert pu._lazy and pu._pairs and vertical in (None, pu._vertical) and pu.form.startswith("two sweeps per pass"), (bc, pu.form)
            _run(solver, ref, 3, f"bc{bc}: {pu.form}")
        finally:
            solver._dev.close()


def _flags_numpy(mask):
    """Restatement of k_lazy_flags / k_pair_list (csrc/fs_march.h) for a single-domain mask: bits 1-4 per (wave column of 248 cells, row)."""
    X, Y = mask.shape
    M = np.pad(mask, 1, constant_values=1).astype(np.int16)
    W, E, S, N = M[:-2, 1:-1], M[2:, 1:-1], M[1:-1, :-2], M[1:-1, 2:]
    wall = mask == 1
    fW = (W == 0) & (S == 1) & (N == 1) & wall
    fE = ~fW & (E == 0) & (S == 1) & (N == 1) & wall
    fS = ~fW & ~fE & (S == 0) & (W == 1) & (E == 1) & wall
    fN = ~fW & ~fE & ~fS & (N == 0) & (W == 1) & (E == 1) & wall
    rest = wall & ~(fW | fE | fS | fN)
    cWN = rest & (W == 0) & (N == 0)
    cEN = rest & ~cWN & (E == 0) & (N == 0)
    cWS = rest & ~cWN & ~cEN & (W == 0) & (S == 0)
    cES = rest & ~cWN & ~cEN & ~cWS & (E == 0) & (S == 0)
    inflow = mask == 2
    inflow_self = np.zeros_like(inflow); inflow_self[-1, :] = inflow[-1, :]          # clamped onto itself: p[t] = p[t], no recipe
    target = fW | fE | fS | fN | cWN | cEN | cWS | cES | (inflow & ~inflow_self) | (mask == 3)
    # the vertical source (if any) and the cell on the other side of it; in-domain neighbours only (sample() clamps, rows 0 / Y-1 are walls)
    src_below = fS | cWS | cES
    src_above = fN | cWN | cEN
    Mc = np.pad(mask, 1, mode="edge").astype(np.int16)                                # the kernel clamps rows / columns onto the edge
    Wc, Ec, Sc, Nc = Mc[:-2, 1:-1], Mc[2:, 1:-1], Mc[1:-1, :-2], Mc[1:-1, 2:]
    vert = src_below | src_above
    hard = (src_below & (Nc != 1)) | (src_above & (Sc != 1)) | (wall & ~target & ((Wc != 1) | (Ec != 1) | (Sc != 1) | (Nc != 1)))
    dirty = wall | target
    nw = (X // 4 + 61) // 62
    out = np.zeros((nw, Y), np.uint8)
    for w in range(nw):
        lo, hi = w * 248, min(X, w * 248 + 248)
        out[w] |= np.where(dirty[max(lo - 2, 0):hi + 2].any(axis=0), 2, 0).astype(np.uint8)
        out[w] |= np.where(hard[max(lo - 4, 0):hi + 4].any(axis=0), 4, 0).astype(np.uint8)
        out[w] |= np.where(vert[max(lo - 4, 0):hi + 4].any(axis=0), 8, 0).astype(np.uint8)
        for bits, flag in ((12, 16), (4, 32)):          # rows the general path owns without / with the vertical tile path
            g = (out[w] & bits) != 0
            G = g.copy()
            for d in (1, 2):
                G[d:] |= g[:-d]; G[:-d] |= g[d:]
            out[w] |= np.where(G, flag, 0).astype(np.uint8)
    computed = np.stack([(~wall[w * 248:min(X, w * 248 + 248)]).any(axis=0) for w in range(nw)])
    return out, (int((((out & 16) != 0) & computed).sum()), int((((out & 32) != 0) & computed).sum()))


@pytest.mark.parametrize("n,res", [(1, 128), (2, 200), (3, 128), (4, 128), (5, 256), (2, 1600)])
def test_tile_classification_matches_its_definition(n, res, hip_lib):
    """The flags that route rows between the plain, the row-local and the general path of the two-sweep kernel (include/fs_hip.h
    fs_lazy_flags) against a numpy restatement - a mask that sends most rows down the general path would still be bit-exact, only slow."""
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    const, mask, _ = create_scene_arrays(n, res)
    fs.runtime.init(gpu=0, dtype="f32")
    bc = BoundaryCondition(const, mask)
    try:
        got, n_general = bc.device.lazy_flags()
        want, want_general = _flags_numpy(mask)
        for bit in (2, 4, 8, 16, 32):
            bad = np.argwhere((got & bit) != (want & bit))
            assert len(bad) == 0, f"bit {bit}: {len(bad)} (wave column, row) entries differ, first {bad[:5].tolist()}, got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"
        assert n_general == want_general
    finally:
        bc.device.close()
