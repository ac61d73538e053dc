"""Two red-black SOR iterations per pass (csrc/fs_rbpair.h, fs_rbsor_pair) against the launch-by-launch form
(pressure BC, fused iteration, swap - itself pinned against the oracle and the golden vectors): every pressure buffer,
wall cells included, bit for bit; with uploads (the carrying pass), odd iteration counts (pair passes + a single iteration),
ragged sizes, the reference's scenes, and directly against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def thick_scene(rng, X, Y, boxes=8, outflow=True):
    """Channel with floor / ceiling, an inflow strip on the left, outflow (or a wall) on the right and random boxes at least
    two cells thick in both directions: a union of such boxes has no one-cell-thin wall, so the pair pass is admitted."""
    mask = np.zeros((X, Y), np.uint8)
    mask[:, :2] = 1
    mask[:, -2:] = 1
    mask[:2, 2:-2] = 2
    mask[-2:, 2:-2] = 3 if outflow else 1
    for _ in range(boxes):
        w, h = int(rng.integers(2, max(3, X // 5))), int(rng.integers(2, max(3, Y // 3)))
        i, j = int(rng.integers(4, max(5, X - 2 - w))), int(rng.integers(0, max(1, Y - h)))      # (not against the inflow strip: an inflow
        # cell whose right neighbour is a wall reads that wall cell's history, which no lazily evaluated boundary pass admits)
        mask[i:i + w, j:j + h] = 1
    const = np.zeros((X, Y, 2), np.float32)
    const[mask == 2] = (1.0, 0.0)
    return const, mask


def build(const, mask, n_iter, pair, res=64, vc=5.0, scheme="cip", omega=1.3, dtype="f32"):
    import fs
    from fs.boundary_condition import BoundaryCondition
    dt, dx, re = 0.05 / res, 1.0 / res, 1000.0
    fs.runtime.init(gpu=0, dtype=dtype)
    bc = BoundaryCondition(const, mask)
    vcobj = fs.VorticityConfinement(bc, dt, dx, vc) if vc else None
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, omega, n_iter, pair=pair)
    if scheme == "cip":
        return fs.CipMacSolver(bc, pu, dt, dx, re, vcobj)
    return fs.MacSolver(bc, pu, fs.advect_kk_scheme if scheme == "kk" else fs.advect_upwind, dt, dx, re, vcobj)


def same_pressure(a, b, what):
    for which in ("current", "next"):
        x, y = getattr(a.p, which).to_numpy(), getattr(b.p, which).to_numpy()
        assert np.array_equal(x, y, equal_nan=True), f"{what}: p.{which} differs in {int((x != y).sum())} cells"
    assert np.array_equal(a.v.current.to_numpy(), b.v.current.to_numpy(), equal_nan=True), f"{what}: v differs"


@pytest.mark.parametrize("X,Y", [(64, 16), (64, 37), (248, 24), (252, 41), (500, 18), (1000, 12), (32, 8)])
@pytest.mark.parametrize("n_iter", [2, 3, 4, 5])
def test_pair_pass_equals_single_iterations(X, Y, n_iter, hip_lib, monkeypatch):
    if (X + n_iter) % 3 == 0:
        monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")      # plain / boundary workgroups as two compact launches (large grids do that by themselves)
    rng = np.random.default_rng(X * 100 + Y + n_iter)
    const, mask = thick_scene(rng, X, Y, outflow=(X + n_iter) % 2 == 0)
    a, b = build(const, mask, n_iter, True), build(const, mask, n_iter, False)
    try:
        assert a._dev.rb_pair_ok and a.pressure_updater._pair and not b.pressure_updater._pair
        for step in range(5):
            a.update()
            b.update()
            same_pressure(a, b, f"{X}x{Y} n_iter {n_iter} step {step + 1}")
    finally:
        a._dev.close()
        b._dev.close()


@pytest.mark.parametrize("seed", range(6))
def test_uploads_are_carried(seed, hip_lib):
    """Random data uploaded into p.current / p.next (wall cells included) at the start and again in mid-run: the cells no kernel
    writes must follow the reference's two-buffer rotation although the pair pass rotates through four buffers."""
    rng = np.random.default_rng(seed)
    X, Y = [(64, 24), (128, 16), (252, 20), (64, 33), (500, 10), (96, 48)][seed]
    n_iter = [2, 3, 2, 4, 5, 2][seed]
    const, mask = thick_scene(rng, X, Y)
    a, b = build(const, mask, n_iter, True), build(const, mask, n_iter, False)
    try:
        def upload(which):
            arr = rng.uniform(-1, 1, (X, Y)).astype(np.float32)
            for s in (a, b):
                getattr(s.p, which).from_numpy(arr)
        v0 = rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32)
        for s in (a, b):
            s.v.current.from_numpy(v0)
        upload("current")
        if seed % 2:
            upload("next")
        for step in range(7):
            if step == 3:
                upload("next" if seed % 3 else "current")
            if step == 5 and seed % 2 == 0:
                for s in (a, b):
                    s.p.next.fill(0.25)
            a.update()
            b.update()
            same_pressure(a, b, f"seed {seed} step {step + 1}")
    finally:
        a._dev.close()
        b._dev.close()


@pytest.mark.parametrize("bc", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("res", [64, 100, 256])
def test_reference_scenes_admit_the_pair_pass(bc, res, hip_lib, monkeypatch):
    import fs
    if res == 256:
        monkeypatch.setenv("FS_RBPAIR_SPLIT", "2")
    sims = []
    try:
        for pair in ("1", "0"):
            import os
            os.environ["FS_RBSOR_PAIR"] = pair
            fs.runtime.init(gpu=0, dtype="f32")
            sims.append(fs.FluidSimulator.create(bc, res, 0.05 / res, 1.0 / res, 1e6, 5.0, "cip"))
        os.environ.pop("FS_RBSOR_PAIR")
        a, b = sims[0]._solver, sims[1]._solver
        if bc == 3 and not a._dev.rb_pair_ok:
            # scene 3's discs (radius 16 res / 500 cells) have single cells sticking out at their extremes up to res ~1000: a boundary
            # cell whose recipe reads the side opposite to a fluid reader - the mask is refused and the single iterations run
            assert not a.pressure_updater._pair
            pytest.skip("discs with one-cell protrusions: the mask does not admit the pair pass")
        assert a._dev.rb_pair_ok, f"scene {bc} at res {res} should admit the pair pass"
        assert a.pressure_updater._pair and not b.pressure_updater._pair
        for step in range(6):
            a.update()
            b.update()
            same_pressure(a, b, f"bc{bc} res {res} step {step + 1}")
    finally:
        import os
        os.environ.pop("FS_RBSOR_PAIR", None)
        for s in sims:
            s._solver._dev.close()


@pytest.mark.parametrize("scheme,vc", [("cip", 5.0), ("kk", 10.0), ("upwind", None)])
def test_pair_pass_against_the_oracle(scheme, vc, hip_lib):
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    X, Y, res = 256, 96, 128
    const, mask = thick_scene(rng, X, Y, boxes=14)
    a = build(const, mask, 2, True, res=res, vc=vc, scheme=scheme)
    ref = O.make_simulator(const, mask, None, scheme=scheme, dt=0.05 / res, dx=1.0 / res, re=1000.0, vor_eps=vc, updater=("rbsor", 1.3, 2))
    try:
        assert a.pressure_updater._pair
        for step in range(20):
            a.update()
            ref.update()
        assert np.array_equal(a.p.current.to_numpy(), ref.p.current, equal_nan=True)
        assert np.array_equal(a.p.next.to_numpy(), ref.p.next, equal_nan=True)
        assert np.array_equal(a.v.current.to_numpy(), ref.v.current, equal_nan=True)
        assert float(np.abs(ref.p.current).max()) > 0
    finally:
        a._dev.close()


def test_thin_walls_are_refused(hip_lib):
    """A wall one cell thick between two fluid regions: the mask is not admitted and the updater keeps the single iterations."""
    import fs
    rng = np.random.default_rng(3)
    const, mask = thick_scene(rng, 64, 32, boxes=0)
    mask[30, 6:20] = 1
    s = build(const, mask, 2, True)
    try:
        assert not s._dev.rb_pair_ok and not s.pressure_updater._pair
        s.update()
    finally:
        s._dev.close()


@pytest.mark.parametrize("X,Y,n_iter,split", [(64, 16, 2, "0"), (252, 41, 3, "2"), (500, 37, 4, "2"), (1000, 12, 2, "0"), (124, 130, 5, "2")])
def test_pair_pass_in_f64(X, Y, n_iter, split, hip_lib, monkeypatch):
    """Round 4: the pass on double2 lanes (2-row tiles; the plain part on 4-row tiles where the two-part launch is on) - BASELINE
    configs[4]'s fp64 truth leg runs it.  Against the f64 single iterations and the f64 oracle, every pressure buffer bit for bit."""
    from oracle import oracle as O
    monkeypatch.setenv("FS_RBPAIR_SPLIT", split)
    rng = np.random.default_rng(X + Y * 7 + n_iter)
    const, mask = thick_scene(rng, X, Y, boxes=9, outflow=n_iter % 2 == 0)
    a, b = build(const, mask, n_iter, True, dtype="f64"), build(const, mask, n_iter, False, dtype="f64")
    res = 64
    ref = O.make_simulator(const.astype(np.float64), mask, None, scheme="cip", dt=0.05 / res, dx=1.0 / res, re=1000.0, vor_eps=5.0,
                           updater=("rbsor", 1.3, n_iter), dtype=np.float64)
    try:
        assert a._dev.rb_pair_ok and a.pressure_updater._pair and not b.pressure_updater._pair
        v0 = rng.uniform(-1, 1, (X, Y, 2))
        p0 = rng.uniform(-1, 1, (X, Y))
        for s in (a, b):
            s.v.current.from_numpy(v0)
            s.p.current.from_numpy(p0)
        ref.v.current[...] = v0
        ref.p.current[...] = p0
        for step in range(4):
            a.update(); b.update(); ref.update()
            same_pressure(a, b, f"f64 {X}x{Y} n_iter {n_iter} step {step + 1}")
            assert np.array_equal(a.p.current.to_numpy(), ref.p.current, equal_nan=True), f"f64 vs oracle p, step {step + 1}"
            assert np.array_equal(a.p.next.to_numpy(), ref.p.next, equal_nan=True), f"f64 vs oracle p.next, step {step + 1}"
    finally:
        a._dev.close()
        b._dev.close()


@pytest.mark.parametrize("res,one_launch", [(800, True), (400, False)])
def test_launch_form_of_the_pair_by_grid_size(res, one_launch, hip_lib, monkeypatch):
    """Round 6: from 1 M cells the two iterations run as ONE launch over the all-fluid (stacked) and the masked tiles (k_rbsor_pair_all); below, the 2-row
    tiles of small grids in one kernel with per-wave hints.  The kernel that ran is read back from the library (fs_prof_kernels); values against the oracle."""
    import fs
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    for k in ("FS_RBPAIR_SPLIT", "FS_SMALL_CELLS", "FS_TILE_LIST"):
        monkeypatch.delenv(k, raising=False)
    dt, dx, re, vc = 0.05 / res, 1.0 / res, 1.0e6, 5.0
    fs.runtime.init(gpu=0, dtype="f32")
    sim = fs.FluidSimulator.create(5, res, dt, dx, re, vc, "cip")
    dev = sim._solver._bc.device
    const, mask, _ = create_scene_arrays(5, res)
    ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=re, vor_eps=vc)
    try:
        dev.profile(True)
        for _ in range(3):
            sim.step()
            ref.update()
        ks = dev.profile_kernels("rbsor_pair")
        assert len(ks) == 1 and ks[0].startswith("fs::k_rbsor_pair_all<" if one_launch else "fs::k_rbsor_pair<2, 2, "), ks
        out = sim.field_to_numpy()
        for name, e in ref.fields().items():
            assert np.array_equal(out[name], e, equal_nan=True), (res, name)
    finally:
        dev.close()
