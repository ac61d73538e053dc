"""MI355X-native drop-in for the `fs` package of takah29/2d-fluid-simulator (step() hot path).

Same class / function names as the reference package; the Taichi kernels are replaced by hand-written
HIP kernels in ../csrc (libfs_hip.so, C-ABI in include/fs_hip.h).  `fs.runtime.init(...)` plays the
role of `ti.init(...)`.
"""
from . import runtime  # noqa: F401
from .advection import advect_kk_scheme, advect_upwind  # noqa: F401
from .boundary_condition import (BoundaryCondition, DyeBoundaryCondition,  # noqa: F401
                                 get_boundary_condition)
from .double_buffer import DoubleBuffer  # noqa: F401
from .fluid_simulator import DyeFluidSimulator, FluidSimulator  # noqa: F401
from .pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater  # noqa: F401
from .solver import CipMacSolver, DyeCipMacSolver, DyeMacSolver, MacSolver  # noqa: F401
from .vorticity_confinement import VorticityConfinement  # noqa: F401
