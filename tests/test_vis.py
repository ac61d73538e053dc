"""Visualisation buffers: `get_norm_field / get_pressure_field / get_vorticity_field / get_dye_field`
against RGB buffers produced by the reference's own `_to_norm / _to_pressure / _to_vorticity / _to_dye` kernels
(fs/fluid_simulator.py:38-58, 121-126, colour maps fs/visualization.py:8-22; fixtures tests/golden/vis_*.npz written by
tests/golden/make_golden.py `vis`).

Tolerance: 0 - value equality of every f32 (np.array_equal; the sign of a zero produced by max(-0.0, 0.0) is not defined
by IEEE maxNum and is not compared).  The scales 0.2 / 0.002, 0.04, 0.005, the wall colour and the blue/red split of the
signed maps are all pinned by these buffers."""
import glob
import os

import numpy as np
import pytest
from conftest import GOLDEN

FILES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "vis_*.npz")))


def test_fixture_inventory():
    assert len(FILES) >= 4


@pytest.mark.parametrize("fname", FILES)
def test_oracle_maps_equal_the_reference_buffers(fname):
    from oracle import oracle as O
    g = np.load(os.path.join(GOLDEN, fname))
    dx = float(g["params"][3])
    v, p, dye, mask = g["state.v"], g["state.p"], g["state.dye"], g["bc_mask"]
    got = {"norm": O.vis_norm(v, p, mask), "pressure": O.vis_pressure(p, mask), "vorticity": O.vis_vorticity(v, dx, mask),
           "dye": O.vis_dye(dye, mask)}
    for k, a in got.items():
        e = g[f"rgb.{k}"]
        assert a.dtype == e.dtype and a.shape == e.shape, k
        assert np.array_equal(a, e), f"{fname} {k}: max|d| = {np.abs(a - e).max()}"
    wall = mask == 1
    assert np.array_equal(g["rgb.norm"][wall], np.broadcast_to(np.float32([0.5, 0.7, 0.5]), g["rgb.norm"][wall].shape))
    assert float(g["rgb.vorticity"][~wall].max()) > 0 and float(g["rgb.pressure"][~wall].max()) > 0     # non-trivial images


def _product_with_state(g, dtype="f32"):
    import fs
    from fs.boundary_condition import DyeBoundaryCondition
    bcn, res, dt, dx, re, vc = [float(x) for x in g["params"]]
    fs.runtime.init(gpu=0, dtype=dtype)
    bc = DyeBoundaryCondition(g["bc_const"], g["bc_dye"], g["bc_mask"])
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    scheme = str(g["scheme"])
    if scheme == "cip":
        solver = fs.DyeCipMacSolver(bc, pu, dt, dx, re, fs.VorticityConfinement(bc, dt, dx, vc))
    else:
        adv = fs.advect_upwind if scheme == "upwind" else fs.advect_kk_scheme
        solver = fs.DyeMacSolver(bc, pu, adv, dt, dx, re, fs.VorticityConfinement(bc, dt, dx, vc))
    solver.v.current.from_numpy(g["state.v"])
    solver.p.current.from_numpy(g["state.p"])
    solver.dye.current.from_numpy(g["state.dye"])
    return fs.DyeFluidSimulator(solver)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_device_kernels_equal_the_reference_buffers(fname, hip_lib):
    g = np.load(os.path.join(GOLDEN, fname))
    sim = _product_with_state(g)
    try:
        for k, getter in (("norm", sim.get_norm_field), ("pressure", sim.get_pressure_field),
                          ("vorticity", sim.get_vorticity_field), ("dye", sim.get_dye_field)):
            field = getter()
            assert field is sim.rgb_buf                      # like the reference: the getters return the image field
            a, e = field.to_numpy(), g[f"rgb.{k}"]
            assert a.dtype == e.dtype and a.shape == e.shape, k
            assert np.array_equal(a, e), f"{fname} {k}: max|d| = {np.abs(a - e).max()}"
    finally:
        sim._solver._bc.device.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bc,res,scheme,dtype,steps", [(5, 512, "cip", "f32", 40), (2, 200, "cip", "f32", 30),
                                                       (3, 256, "kk", "f64", 20), (1, 130, "upwind", "f32", 25)])
def test_device_kernels_equal_the_oracle_on_a_developed_flow(bc, res, scheme, dtype, steps, hip_lib):
    """Larger grids (many waves per row, non-power-of-two dx, X % 4 != 0, f64) against the oracle restatement."""
    import fs
    from oracle import oracle as O
    dt, dx = 0.05 / res, 1.0 / res
    fs.runtime.init(gpu=0, dtype=dtype)
    sim = fs.DyeFluidSimulator.create(bc, res, dt, dx, 1e6, 5.0, scheme)
    try:
        sim.run(steps, graph=False)
        f = sim.field_to_numpy()
        mask = sim._solver._bc.mask
        exp = {"norm": O.vis_norm(f["v"], f["p"], mask), "pressure": O.vis_pressure(f["p"], mask),
               "vorticity": O.vis_vorticity(f["v"], dx, mask), "dye": O.vis_dye(f["dye"], mask)}
        got = {"norm": sim.get_norm_field().to_numpy(), "pressure": sim.get_pressure_field().to_numpy(),
               "vorticity": sim.get_vorticity_field().to_numpy(), "dye": sim.get_dye_field().to_numpy()}
        for k in exp:
            assert got[k].dtype == exp[k].dtype and np.array_equal(got[k], exp[k]), k
        assert float(exp["vorticity"][mask != 1].max()) > 0
    finally:
        sim._solver._bc.device.close()
