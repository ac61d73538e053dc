"""Every selectable launch variant of the fast paths (env knobs read at fs_create) produces the same bits as the
one-cell-per-lane kernels (FS_MARCH=0), which the small-size tests pin against the oracle and the golden vectors:
  FS_JACOBI   22 / 24 / 21 = overlapped-wave register tiles of 2 / 4 / 1 rows, 30 = LDS halo tile
  FS_XCD      bit mask of the kernels launched in XCD-grouped block order (0 = all row-major, 63 = all grouped, default);
  FS_XCD_GROUP = tile rows per XCD group (1..128, default 8)
  FS_STACK    bit mask of the kernels whose workgroups are 4 stacked tile rows of one wave column instead of 4 wave columns of one row
Grid 2*res x res with res = 520: several waves per row, a ragged last wave, row count not a multiple of any tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES = 520


def _run(monkeypatch, env, what):
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    from fs.runtime import Device
    for k in ("FS_MARCH", "FS_JACOBI", "FS_XCD", "FS_XCD_GROUP", "FS_STACK", "FS_SMALL_TILES", "FS_SMALL_CELLS", "FS_RBPAIR_RT", "FS_RBPAIR_SPLIT",
              "FS_RBPAIR_PLAIN_RT", "FS_BC_NOPAIRS", "FS_K34_RT", "FS_TILE_LIST", "FS_SPLIT_WGW"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    const, mask, _ = create_scene_arrays(5, RES)
    dev = Device(mask.shape[0], mask.shape[1], np.float32, gpu=0)
    bc = BoundaryCondition(const, mask, device=dev)
    rng = np.random.default_rng(7)
    X, Y = mask.shape
    dt, dx = 0.05 / RES, 1.0 / RES
    v, pa, pb, src = dev.alloc(2), dev.alloc(1), dev.alloc(1), dev.alloc(2)
    v.from_numpy(rng.uniform(-1, 1, (X, Y, 2)).astype(np.float32))
    pa.from_numpy(rng.uniform(-10, 10, (X, Y)).astype(np.float32))
    pb.from_numpy(rng.uniform(-10, 10, (X, Y)).astype(np.float32))
    out = {}
    if what == "jacobi":
        dev.jacobi_sweep(dt, dx, pb, pa, v)
        out["v-form"] = pb.to_numpy()
        dev.poisson_source(dt, dx, src, v)
        dev.jacobi_sweep_src(pa, pb, src)
        out["src-form"] = pa.to_numpy()
    else:
        vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
        pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
        solver = fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc)
        for _ in range(3):
            solver.update()
        for name in ("v", "p", "vx", "vy"):
            out[name] = getattr(solver, name).current.to_numpy()
    dev.close()
    return out


@pytest.mark.parametrize("variant", ["22", "24", "21", "30"])
def test_jacobi_tile_variants(variant, hip_lib, monkeypatch):
    ref = _run(monkeypatch, {"FS_MARCH": "0"}, "jacobi")
    got = _run(monkeypatch, {"FS_JACOBI": variant}, "jacobi")
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (variant, k)


@pytest.mark.parametrize("env", [{"FS_XCD": "0"}, {"FS_XCD": "21"}, {"FS_XCD_GROUP": "1"}, {"FS_XCD_GROUP": "3"},
                                 {"FS_XCD_GROUP": "16"}, {"FS_XCD_GROUP": "128"}, {"FS_STACK": "63"}, {"FS_STACK": "63", "FS_XCD": "0"},
                                 {"FS_STACK": "21", "FS_XCD_GROUP": "3"}],
                         ids=lambda e: "-".join(f"{k[3:]}{v}" for k, v in e.items()))
def test_block_order_variants(env, hip_lib, monkeypatch):
    ref = _run(monkeypatch, {"FS_MARCH": "0"}, "step")
    got = _run(monkeypatch, env, "step")
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (env, k)
    assert float(np.abs(ref["p"]).max()) > 0


@pytest.mark.parametrize("env", [{"FS_SMALL_TILES": "0"}, {"FS_SMALL_CELLS": "0"}, {"FS_RBPAIR_RT": "4"}, {"FS_RBPAIR_RT": "6"}, {"FS_BC_NOPAIRS": "1"},
                                 {"FS_K34_RT": "1"}, {"FS_K34_RT": "4"}, {"FS_TILE_LIST": "0"},
                                 {"FS_RBPAIR_SPLIT": "2"}, {"FS_RBPAIR_SPLIT": "2", "FS_SMALL_TILES": "0"}, {"FS_RBPAIR_SPLIT": "2", "FS_RBPAIR_PLAIN_RT": "4"},
                                 {"FS_RBPAIR_SPLIT": "2", "FS_SPLIT_WGW": "4"}],
                         ids=lambda e: "-".join(f"{k[3:]}{v}" for k, v in e.items()))
def test_round4_tile_height_and_list_variants(env, hip_lib, monkeypatch):
    """Round 4: tile heights by grid size (this grid, 0.54 M cells, takes the small-grid heights by default), flat op-list pairs, the per-wave
    plain hints of the launch lists (FS_TILE_LIST=0: dense launches without them), the two-part launches with the mirrored 8-row plain tiles
    of the pair pass forced onto a small grid: the same bits as the one-cell-per-lane kernels."""
    ref = _run(monkeypatch, {"FS_MARCH": "0"}, "step")
    got = _run(monkeypatch, env, "step")
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (env, k)
