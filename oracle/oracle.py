"""ctypes front-end of the CPU oracle (oracle/libfs_oracle.so) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  Arrays use the reference's own layout ((X, Y[, C]) with y contiguous,
AoS channels), dtype float32 or float64.

The classes restate the reference's host-side orchestration (which kernel runs on which physical
buffer, and when the DoubleBuffers swap - SURVEY.md hazard H5):
  OraclePressure*      fs/pressure_updater.py:41-114
  OracleVorticity      fs/vorticity_confinement.py:9-59
  OracleMacSolver      fs/solver.py:53-161   (MacSolver / DyeMacSolver)
  OracleCipSolver      fs/solver.py:165-401  (CipMacSolver / DyeCipMacSolver)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VELOCITY_LIMIT = 10.0  # fs/solver.py:12


def build(force=False):
    so = os.path.join(_HERE, "libfs_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("fs_oracle.c", "fs_oracle_impl.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfs_oracle.so"])
    return so


def set_threads(n):
    """OpenMP team size of the oracle's race-free kernels (tests keep it small for tiny grids, large for BASELINE sizes)."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        assert _LIB.oracle_abi_version() == 1
    return _LIB


_I, _D, _P = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _call(name, dtype, *args):
    fn = getattr(lib(), f"{name}_{_suf(dtype)}")
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(_ptr(a))
        elif isinstance(a, (int, np.integer)):
            conv.append(_I(int(a)))
        else:
            conv.append(_D(float(a)))
    fn.restype = None
    fn(*conv)


class Buf2:
    """fs/double_buffer.py:4-18 - two physical arrays and a reference swap."""

    def __init__(self, shape, n_channel, dtype):
        full = tuple(shape) if n_channel == 1 else tuple(shape) + (n_channel,)
        self.current = np.zeros(full, dtype)
        self.next = np.zeros(full, dtype)

    def swap(self):
        self.current, self.next = self.next, self.current


class OracleBC:
    """fs/boundary_condition.py:12-112 (BoundaryCondition / DyeBoundaryCondition).

    `cell_list`: run the velocity / pressure kernels over the precomputed list of cells that can act at all (non-fluid cells
    that are inflow / outflow or touch a fluid cell), in the same serial (i, j) order, instead of walking every cell of the
    grid serially - identical results (tests/test_oracle_golden.py), O(perimeter) work.  Default: on for grids above 1 M cells
    (the CPU-baseline sizes); the small parity grids keep the literal full scan."""

    def __init__(self, bc_const, bc_mask, bc_dye=None, dtype=np.float32, cell_list=None):
        self.dtype = np.dtype(dtype)
        self.bc_const = np.ascontiguousarray(bc_const, dtype=self.dtype)
        self.mask = np.ascontiguousarray(bc_mask, dtype=np.uint8)
        self.bc_dye = None if bc_dye is None else np.ascontiguousarray(bc_dye, dtype=self.dtype)
        self.X, self.Y = self.mask.shape
        self.cells = None
        if cell_list if cell_list is not None else self.mask.size > (1 << 20):
            m = self.mask
            fluid = m == 0
            near = np.zeros_like(fluid)
            near[1:, :] |= fluid[:-1, :]; near[:-1, :] |= fluid[1:, :]
            near[:, 1:] |= fluid[:, :-1]; near[:, :-1] |= fluid[:, 1:]
            self.cells = np.ascontiguousarray(np.flatnonzero((m >= 2) | ((m == 1) & near)), dtype=np.int64)   # ascending = (i, j) order

    def get_resolution(self):
        return (self.X, self.Y)

    def set_velocity_boundary_condition(self, v):
        if self.cells is not None:
            _call("oracle_velocity_bc_list", self.dtype, self.X, self.Y, self.mask, self.bc_const, v, self.cells, len(self.cells))
        else:
            _call("oracle_velocity_bc", self.dtype, self.X, self.Y, self.mask, self.bc_const, v)

    def set_pressure_boundary_condition(self, p):
        if self.cells is not None:
            _call("oracle_pressure_bc_list", self.dtype, self.X, self.Y, self.mask, p, self.cells, len(self.cells))
        else:
            _call("oracle_pressure_bc", self.dtype, self.X, self.Y, self.mask, p)

    def set_dye_boundary_condition(self, dye):
        _call("oracle_dye_bc", self.dtype, self.X, self.Y, self.mask, self.bc_dye, dye)


class OracleJacobi:
    def __init__(self, bc, dt, dx, n_iter):
        self.bc, self.dt, self.dx, self.n_iter = bc, dt, dx, n_iter

    def sweep(self, pn, pc, vc):
        b = self.bc
        _call("oracle_jacobi_sweep", b.dtype, b.X, b.Y, self.dt, self.dx, b.mask, pn, pc, vc)

    def update(self, p, v_current):
        for _ in range(self.n_iter):
            self.bc.set_pressure_boundary_condition(p.current)
            self.sweep(p.next, p.current, v_current)
            p.swap()


class OracleRedBlackSor:
    def __init__(self, bc, dt, dx, relaxation_factor, n_iter):
        self.bc, self.dt, self.dx, self.omega, self.n_iter = bc, dt, dx, relaxation_factor, n_iter

    def half(self, parity, pn, pc, vc):
        b = self.bc
        _call("oracle_rbsor_half", b.dtype, b.X, b.Y, self.dt, self.dx, self.omega, parity, b.mask, pn, pc, vc)

    def update(self, p, v_current):
        for _ in range(self.n_iter):
            self.bc.set_pressure_boundary_condition(p.current)
            self.half(1, p.next, p.current, v_current)   # odd cells: next <- f(current)
            self.half(0, p.next, p.next, v_current)      # even cells: in place on next
            p.swap()


class OracleVorticity:
    def __init__(self, bc, dt, dx, weight):
        self.bc, self.dt, self.dx, self.weight = bc, dt, dx, weight
        self.vorticity = np.zeros((bc.X, bc.Y), bc.dtype)
        self.vorticity_abs = np.zeros((bc.X, bc.Y), bc.dtype)

    def calc(self, vc):
        b = self.bc
        _call("oracle_vort_calc", b.dtype, b.X, b.Y, self.dx, b.mask, self.vorticity, self.vorticity_abs, vc)

    def add(self, vn, vc):
        b = self.bc
        _call("oracle_vort_add", b.dtype, b.X, b.Y, self.dt, self.dx, self.weight, b.mask, vn, vc,
              self.vorticity, self.vorticity_abs)

    def apply(self, v):
        self.calc(v.current)
        self.add(v.next, v.current)   # writes v.next only; the caller swaps


def limit_field(v, limit=VELOCITY_LIMIT):
    _call("oracle_limit_field", v.dtype, v.shape[0], v.shape[1], limit, v)


def clamp_field(f, low, high):
    _call("oracle_clamp_field", f.dtype, f.shape[0], f.shape[1], f.shape[2], low, high, f)


class OracleMacSolver:
    """scheme: 'upwind' | 'kk'."""

    def __init__(self, bc, pressure_updater, scheme, dt, dx, re, vorticity_confinement=None, dye=False):
        self.bc, self.pu, self.vc = bc, pressure_updater, vorticity_confinement
        self.scheme = {"upwind": 0, "kk": 1}[scheme]
        self.dt, self.dx, self.re = dt, dx, re
        shape = bc.get_resolution()
        self.v = Buf2(shape, 2, bc.dtype)
        self.p = Buf2(shape, 1, bc.dtype)
        self.dye = Buf2(shape, 3, bc.dtype) if dye else None

    def update(self):
        b = self.bc
        b.set_velocity_boundary_condition(self.v.current)
        _call("oracle_mac_update", b.dtype, b.X, b.Y, self.dt, self.dx, self.re, self.scheme, b.mask,
              self.v.next, self.v.current, self.p.current)
        self.v.swap()
        if self.vc is not None:
            self.vc.apply(self.v)
            self.v.swap()
        self.pu.update(self.p, self.v.current)
        limit_field(self.v.current)
        if self.dye is not None:
            b.set_dye_boundary_condition(self.dye.current)
            _call("oracle_mac_dye", b.dtype, b.X, b.Y, self.dt, self.dx, self.re, self.scheme, b.mask,
                  self.dye.next, self.dye.current, self.v.current)
            self.dye.swap()
            clamp_field(self.dye.current, 0.0, 1.0)

    def fields(self):
        out = {"v": self.v.current, "p": self.p.current}
        if self.dye is not None:
            out["dye"] = self.dye.current
        return out


class OracleCipSolver:
    def __init__(self, bc, pressure_updater, dt, dx, re, vorticity_confinement=None, dye=False):
        self.bc, self.pu, self.vc = bc, pressure_updater, vorticity_confinement
        self.dt, self.dx, self.re = dt, dx, re
        shape = bc.get_resolution()
        dt_ = bc.dtype
        self.v, self.vx, self.vy = Buf2(shape, 2, dt_), Buf2(shape, 2, dt_), Buf2(shape, 2, dt_)
        self.p = Buf2(shape, 1, dt_)
        self._set_grad(self.vx.current, self.vy.current, self.v.current, 2)
        self.dye = None
        if dye:
            self.dye, self.dyex, self.dyey = Buf2(shape, 3, dt_), Buf2(shape, 3, dt_), Buf2(shape, 3, dt_)
            self._set_grad(self.dyex.current, self.dyey.current, self.dye.current, 3)

    def _set_grad(self, fx, fy, f, c):
        b = self.bc
        _call("oracle_cip_set_grad", b.dtype, b.X, b.Y, self.dx, c, fx, fy, f)

    def _transport(self, f, fx, fy, c, advecting_v, nonadv):
        """non-advection phase + gradient update, swap, CIP advection, swap (solver.py:213-227, 385-401)."""
        b = self.bc
        nonadv(f.next, f.current)
        _call("oracle_cip_nonadv_grad", b.dtype, b.X, b.Y, self.dx, c, b.mask,
              fx.next, fy.next, fx.current, fy.current, f.current, f.next)
        f.swap(), fx.swap(), fy.swap()
        adv = f.current if advecting_v is None else advecting_v.current
        _call("oracle_cip_advect", b.dtype, b.X, b.Y, self.dt, self.dx, c, b.mask,
              f.next, fx.next, fy.next, f.current, fx.current, fy.current, adv)
        f.swap(), fx.swap(), fy.swap()

    def update(self):
        b = self.bc
        b.set_velocity_boundary_condition(self.v.current)
        self._transport(
            self.v, self.vx, self.vy, 2, None,
            lambda fn, fc: _call("oracle_cip_nonadv", b.dtype, b.X, b.Y, self.dt, self.dx, self.re, b.mask,
                                 fn, fc, self.p.current))
        if self.vc is not None:
            self.vc.apply(self.v)
            self.v.swap()
        self.pu.update(self.p, self.v.current)
        limit_field(self.v.current)
        if self.dye is not None:
            b.set_dye_boundary_condition(self.dye.current)
            self._transport(
                self.dye, self.dyex, self.dyey, 3, self.v,
                lambda dn, dc: _call("oracle_cip_nonadv_dye", b.dtype, b.X, b.Y, self.dt, self.dx, self.re,
                                     b.mask, dn, dc))
            clamp_field(self.dye.current, 0.0, 1.0)

    def fields(self):
        out = {"v": self.v.current, "p": self.p.current}
        if self.dye is not None:
            out["dye"] = self.dye.current
        return out


def make_simulator(bc_const, bc_mask, bc_dye, *, scheme, dt, dx, re, vor_eps, updater=("rbsor", 1.3, 2),
                   dtype=np.float32):
    """Same composition as FluidSimulator.create / DyeFluidSimulator.create (fluid_simulator.py:60-108,
    129-176) with the pressure updater exposed: ('rbsor', omega, n_iter) or ('jacobi', n_iter)."""
    bc = OracleBC(bc_const, bc_mask, bc_dye, dtype)
    vc = OracleVorticity(bc, dt, dx, vor_eps) if vor_eps is not None else None
    pu = (OracleRedBlackSor(bc, dt, dx, float(updater[1]), int(updater[2])) if updater[0] == "rbsor"
          else OracleJacobi(bc, dt, dx, int(updater[1])))
    dye = bc_dye is not None
    if scheme == "cip":
        return OracleCipSolver(bc, pu, dt, dx, re, vc, dye=dye)
    if scheme in ("upwind", "kk"):
        return OracleMacSolver(bc, pu, scheme, dt, dx, re, vc, dye=dye)
    raise ValueError(f"Unknown scheme: {scheme}")


# ------------------------------------------------------------------------------------------------
# Visualisation maps (fluid_simulator.py:38-58, 121-126; visualization.py:8-22): element-wise IEEE
# arithmetic in the field dtype, so plain NumPy is an exact restatement (no C needed).
# ------------------------------------------------------------------------------------------------
WALL_COLOR = (0.5, 0.7, 0.5)          # fluid_simulator.py:17


def _visualize_pressure(val):
    """visualization.py:14-16: (max(val, 0), 0, max(-val, 0)) with NaN-ignoring max (shim: np.fmax)."""
    z = np.zeros_like(val)
    return np.stack([np.fmax(val, z), z, np.fmax(-val, z)], axis=-1)


def _central(f, axis, dx):
    """differentiation.py:41-50 through sample() (clamp-to-edge): 0.5 * (f[+1] - f[-1]) / dx."""
    n = f.shape[axis]
    up = np.take(f, np.clip(np.arange(n) + 1, 0, n - 1), axis=axis)
    dn = np.take(f, np.clip(np.arange(n) - 1, 0, n - 1), axis=axis)
    return (f.dtype.type(0.5) * (up - dn)) / f.dtype.type(dx)


def _wall(rgb, mask):
    rgb[mask == 1] = np.asarray(WALL_COLOR, rgb.dtype)
    return rgb


def vis_norm(v, p, mask):
    """_to_norm, fluid_simulator.py:38-44: 0.2 * visualize_norm(v), then += 0.002 * visualize_pressure(p)."""
    t = v.dtype.type
    c = np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1])      # Vector.norm(): sqrt of the left-to-right sum of squares
    rgb = t(0.2) * np.stack([c, c, c], axis=-1)
    rgb = rgb + t(0.002) * _visualize_pressure(p)
    return _wall(rgb, mask)


def vis_pressure(p, mask):
    """_to_pressure, fluid_simulator.py:46-51."""
    return _wall(p.dtype.type(0.04) * _visualize_pressure(p), mask)


def vis_vorticity(v, dx, mask):
    """_to_vorticity, fluid_simulator.py:53-58; visualize_vorticity visualization.py:19-22."""
    w = _central(v[..., 1], 0, dx) - _central(v[..., 0], 1, dx)
    return _wall(v.dtype.type(0.005) * _visualize_pressure(w), mask)


def vis_dye(dye, mask):
    """_to_dye, fluid_simulator.py:121-126."""
    return _wall(dye.copy(), mask)
