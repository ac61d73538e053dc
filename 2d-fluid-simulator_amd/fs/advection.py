"""Advection-scheme tokens (reference: fs/advection.py).

In the reference `advect_upwind` / `advect_kk_scheme` are @ti.func objects handed to MacSolver and
inlined into its kernel at JIT time.  Here the per-cell formulas live in the HIP kernels
(csrc/fs_kernels.h: adv_upwind / adv_kk, citing fs/advection.py:12-24 and :27-60); the Python
objects only select which instantiation MacSolver launches.
"""


class AdvectionScheme:
    def __init__(self, name, code, radius):
        self.name, self.code, self.radius = name, code, radius

    def __repr__(self):
        return f"<advection scheme {self.name}>"

    def __call__(self, *args, **kwargs):
        raise TypeError(f"{self.name} is evaluated on the GPU inside MacSolver; it cannot be called from Python")


advect_upwind = AdvectionScheme("upwind", 0, 1)       # 1st-order upwind, +-1 stencil
advect_kk_scheme = AdvectionScheme("kk", 1, 2)        # Kawamura-Kuwahara 3rd-order upwind, +-2 stencil
