"""Every selectable launch variant of the fast paths (env knobs read at fs_create) produces the same bits as the
one-cell-per-lane kernels (FS_MARCH=0), which the small-size tests pin against the oracle and the golden vectors:
  FS_SMALL_CELLS, FS_TILE_LIST, FS_RBPAIR_SPLIT, FS_FUSE_K2: tile heights by grid size, compact launch lists, two- / three-part launches
(the block-order and tile-height A/B switches of rounds 2 - 4 are constants since round 5, the Jacobi tile variants and the pair pass's
 plain-tile heights since round 6: DESIGN.md section 9)
Grid 2*res x res with res = 520: several waves per row, a ragged last wave, row count not a multiple of any tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES = 520


def _run(monkeypatch, env, what):
    import fs
    from fs.boundary_condition import BoundaryCondition, create_scene_arrays
    from fs.runtime import Device
    for k in ("FS_MARCH", "FS_SMALL_CELLS", "FS_RBPAIR_SPLIT", "FS_TILE_LIST", "FS_FUSE_K2"):
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


@pytest.mark.parametrize("env", [{}, {"FS_TILE_LIST": "0"}], ids=["lists", "dense"])
def test_jacobi_sweeps_equal_the_one_cell_per_lane_kernels(env, hip_lib, monkeypatch):
    """the literal sweep on packed lanes of 2 cells with per-wave plain hints (k_jacobi_ov2; dense: without the hints) and the source-pair sweep on quads"""
    ref = _run(monkeypatch, {"FS_MARCH": "0"}, "jacobi")
    got = _run(monkeypatch, env, "jacobi")
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (env, k)


@pytest.mark.parametrize("env", [{"FS_SMALL_CELLS": "0"}, {"FS_TILE_LIST": "0"},
                                 {"FS_RBPAIR_SPLIT": "2"}, {"FS_RBPAIR_SPLIT": "2", "FS_SMALL_CELLS": "0"},
                                 {"FS_RBPAIR_SPLIT": "2", "FS_FUSE_K2": "0"}, {"FS_RBPAIR_SPLIT": "2", "FS_FUSE_K2": "1"}],
                         ids=lambda e: "-".join(f"{k[3:]}{v}" for k, v in e.items()))
def test_round4_tile_height_and_list_variants(env, hip_lib, monkeypatch):
    """Tile heights by grid size (this grid, 0.54 M cells, takes the small-grid heights by default), the per-wave plain hints of the launch
    lists (FS_TILE_LIST=0: dense launches without them), the two- and three-part launches forced onto a small grid - the pair pass's plain part
    as two stacked waves per 16-row tile; fs_cip_step with K2 in registers (one launch: default;
    one launch per kind of tile) and as its own launch: the same bits as the one-cell-per-lane kernels."""
    ref = _run(monkeypatch, {"FS_MARCH": "0"}, "step")
    got = _run(monkeypatch, env, "step")
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (env, k)
