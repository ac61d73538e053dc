"""Facade (reference: fs/fluid_simulator.py): FluidSimulator / DyeFluidSimulator.

`create()` composes boundary condition + vorticity confinement + red-black SOR(omega 1.3, 2 iterations)
+ solver exactly like the reference (fs/fluid_simulator.py:60-108, 129-176); `step()` is the hot path.
`pressure_updater=` is an extension (the reference hard-codes RB-SOR): ("jacobi", n_iter) or
("rbsor", omega, n_iter) or a ready PressureUpdater factory.
"""
import numpy as np

from . import visualization as _vis
from .advection import advect_kk_scheme, advect_upwind
from .boundary_condition import get_boundary_condition
from .pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater
from .solver import CipMacSolver, DyeCipMacSolver, DyeMacSolver, MacSolver
from .vorticity_confinement import VorticityConfinement

_WALL_COLOR = np.array([0.5, 0.7, 0.5], np.float32)   # fs/fluid_simulator.py:17


def _make_updater(spec, bc, dt, dx):
    if spec is None:
        return RedBlackSorPressureUpdater(bc, dt, dx, relaxation_factor=1.3, n_iter=2)
    if callable(spec):
        return spec(bc, dt, dx)
    kind = spec[0]
    if kind == "rbsor":
        return RedBlackSorPressureUpdater(bc, dt, dx, relaxation_factor=float(spec[1]), n_iter=int(spec[2]))
    if kind == "jacobi":
        return JacobiPressureUpdater(bc, dt, dx, int(spec[1]))
    raise ValueError(f"Unknown pressure updater: {spec!r}")


def _compose(num, resolution, dt, dx, re, vor_eps, scheme, enable_dye, pressure_updater):
    if scheme not in ("cip", "upwind", "kk"):
        msg = f"Unknown scheme: {scheme}"
        raise ValueError(msg)
    bc = get_boundary_condition(num, resolution, enable_dye=enable_dye)
    vc = VorticityConfinement(bc, dt, dx, vor_eps) if vor_eps is not None else None
    pu = _make_updater(pressure_updater, bc, dt, dx)
    if scheme == "cip":
        cls = DyeCipMacSolver if enable_dye else CipMacSolver
        return cls(bc, pu, dt, dx, re, vc)
    cls = DyeMacSolver if enable_dye else MacSolver
    adv = advect_upwind if scheme == "upwind" else advect_kk_scheme
    return cls(bc, pu, adv, dt, dx, re, vc)


class FluidSimulator:
    def __init__(self, solver):
        self._solver = solver
        self._wall_color = _WALL_COLOR

    def step(self):
        self._solver.update()

    def field_to_numpy(self):
        fields = self._solver.get_fields()
        return {"v": fields[0].to_numpy(), "p": fields[1].to_numpy()}

    # -- visualisation (GUI side of the reference, fs/fluid_simulator.py:22-58): host-side NumPy on a download --
    def _wall(self, rgb):
        rgb[self._solver._bc.mask == 1] = self._wall_color
        return rgb

    def get_norm_field(self):
        f = self.field_to_numpy()
        return self._wall(0.2 * _vis.visualize_norm(f["v"]) + 0.002 * _vis.visualize_pressure(f["p"]))

    def get_pressure_field(self):
        return self._wall(0.04 * _vis.visualize_pressure(self.field_to_numpy()["p"]))

    def get_vorticity_field(self):
        return self._wall(0.005 * _vis.visualize_vorticity(self.field_to_numpy()["v"], self._solver.dx))

    @staticmethod
    def create(num, resolution, dt, dx, re, vor_eps, scheme, pressure_updater=None):
        return FluidSimulator(_compose(num, resolution, dt, dx, re, vor_eps, scheme, False, pressure_updater))


class DyeFluidSimulator(FluidSimulator):
    def field_to_numpy(self):
        fields = self._solver.get_fields()
        return {"v": fields[0].to_numpy(), "p": fields[1].to_numpy(), "dye": fields[2].to_numpy()}

    def get_dye_field(self):
        return self._wall(self.field_to_numpy()["dye"].copy())

    @staticmethod
    def create(num, resolution, dt, dx, re, vor_eps, scheme, pressure_updater=None):
        return DyeFluidSimulator(_compose(num, resolution, dt, dx, re, vor_eps, scheme, True, pressure_updater))
