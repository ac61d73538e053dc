"""Colour maps of the reference's viewer (fs/visualization.py:8-22).

In the reference these are `@ti.func`s inlined into the `_to_*` kernels of FluidSimulator; here they live in the
device kernel `k_visualize` (csrc/fs_kernels.h) behind `fs_vis_norm / fs_vis_pressure / fs_vis_vorticity /
fs_vis_dye` (include/fs_hip.h).  This module only names the maps and their scale factors so that callers can
introspect what `FluidSimulator.get_*_field()` renders; there is no host-side implementation.
"""

# (C-ABI entry point, scale factors) per reference function - fs/fluid_simulator.py:38-58, 121-126
RENDERERS = {
    "norm": ("fs_vis_norm", {"visualize_norm": 0.2, "visualize_pressure": 0.002}),
    "pressure": ("fs_vis_pressure", {"visualize_pressure": 0.04}),
    "vorticity": ("fs_vis_vorticity", {"visualize_vorticity": 0.005}),
    "dye": ("fs_vis_dye", {}),
}
WALL_COLOR = (0.5, 0.7, 0.5)   # fs/fluid_simulator.py:17
