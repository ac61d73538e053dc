"""Vorticity confinement (reference: fs/vorticity_confinement.py:9-59)."""


class VorticityConfinement:
    def __init__(self, boundary_condition, dt, dx, weight):
        self._bc = boundary_condition
        self._dev = boundary_condition.device
        self.dt = dt
        self.dx = dx
        self.weight = weight
        self._resolution = boundary_condition.get_resolution()
        self.vorticity = self._dev.alloc(1)
        self.vorticity_abs = self._dev.alloc(1)

    def _calc_vorticity(self, vc):
        self._dev.vort_calc(self.dx, self.vorticity, self.vorticity_abs, vc)

    def _add_vorticity(self, vn, vc):
        self._dev.vort_add(self.dt, self.dx, self.weight, vn, vc, self.vorticity, self.vorticity_abs)

    def apply(self, v):
        """Writes v.next only; the caller swaps (fs/vorticity_confinement.py:57-59, fs/solver.py:84-86)."""
        self._calc_vorticity(v.current)
        self._add_vorticity(v.next, v.current)
