"""Per-step orchestration (reference: fs/solver.py).

Each `update()` issues the same kernel sequence, on the same physical buffers, with the same
DoubleBuffer swaps as the reference - that choreography is part of the algorithm.
The kernels themselves are HIP (csrc/fs_kernels.h); launches are asynchronous on the device stream.
"""
import os

import numpy as np
from abc import ABCMeta, abstractmethod

from .double_buffer import DoubleBuffer

VELOCITY_LIMIT = 10.0


class Solver(metaclass=ABCMeta):
    def __init__(self, boundary_condition):
        self._bc = boundary_condition
        self._dev = boundary_condition.device
        self.resolution = boundary_condition.get_resolution()

    @abstractmethod
    def update(self):
        pass

    @abstractmethod
    def get_fields(self):
        pass

    def is_wall(self, i, j):
        return self._bc.is_wall(i, j)

    def is_fluid_domain(self, i, j):
        return self._bc.is_fluid_domain(i, j)


def limit_field(field, limit, defer=False):
    """Cap the velocity magnitude (fs/solver.py:38-43).  defer: the end-of-step call of the solvers - the launch may ride with the next
    step's velocity boundary kernel (runtime.DeviceBase.limit_field; same results)."""
    field.dev.limit_field(limit, field, defer=defer)


def clamp_field(field, low, high):
    """Clamp every channel (fs/solver.py:46-49)."""
    field.dev.clamp_field(low, high, field)


class MacSolver(Solver):
    """Explicit-Euler MAC-style solver with upwind or Kawamura-Kuwahara advection (fs/solver.py:53-107)."""

    def __init__(self, boundary_condition, pressure_updater, advect_function, dt, dx, re, vorticity_confinement=None):
        super().__init__(boundary_condition)
        self._advect = advect_function
        self.dt = dt
        self.dx = dx
        self.re = re
        self.pressure_updater = pressure_updater
        self.vorticity_confinement = vorticity_confinement
        self.v = DoubleBuffer(self.resolution, 2, self._dev)
        self.p = DoubleBuffer(self.resolution, 1, self._dev)

    def _flow_step(self):
        self._bc.set_velocity_boundary_condition(self.v.current)
        self._update_velocities(self.v.next, self.v.current, self.p.current)
        self.v.swap()
        if self.vorticity_confinement is not None:
            self.vorticity_confinement.apply(self.v)
            self.v.swap()
        self.pressure_updater.update(self.p, self.v.current)
        limit_field(self.v.current, VELOCITY_LIMIT, defer=True)

    def update(self):
        self._flow_step()

    def get_fields(self):
        return self.v.current, self.p.current

    def _update_velocities(self, vn, vc, pc):
        self._dev.mac_update(self._advect.code, self.dt, self.dx, self.re, vn, vc, pc)


class DyeMacSolver(MacSolver):
    """MacSolver + passive dye transport (fs/solver.py:110-161)."""

    def __init__(self, boundary_condition, pressure_updater, advect_function, dt, dx, re, vorticity_confinement=None):
        super().__init__(boundary_condition, pressure_updater, advect_function, dt, dx, re, vorticity_confinement)
        self.dye = DoubleBuffer(self.resolution, 3, self._dev)

    def update(self):
        self._flow_step()
        self._bc.set_dye_boundary_condition(self.dye.current, self.v.current)
        self._update_dye(self.dye.next, self.dye.current, self.v.current)
        self.dye.swap()
        clamp_field(self.dye.current, 0.0, 1.0)

    def get_fields(self):
        return self.v.current, self.p.current, self.dye.current

    def _update_dye(self, dn, dc, vc):
        self._dev.mac_dye(self._advect.code, self.dt, self.dx, dn, dc, vc)


class CipMacSolver(Solver):
    """Two-phase CIP solver: non-advection phase (pressure gradient + diffusion, with the gradient fields
    updated alongside), then CIP advection of value and gradients (fs/solver.py:165-332)."""

    def __init__(self, boundary_condition, pressure_updater, dt, dx, re, vorticity_confinement=None, fused_transport=None, fused_k2=None):
        super().__init__(boundary_condition)
        self.dt = dt
        self.dx = dx
        self.re = re
        self.pressure_updater = pressure_updater
        self.vorticity_confinement = vorticity_confinement
        self.v = DoubleBuffer(self.resolution, 2, self._dev)
        self.vx = DoubleBuffer(self.resolution, 2, self._dev)
        self.vy = DoubleBuffer(self.resolution, 2, self._dev)
        self.p = DoubleBuffer(self.resolution, 1, self._dev)
        self._set_grad(self.vx.current, self.vy.current, self.v.current)
        # Fused gradient-update + advection pass (same observable bits, 64 instead of 98 B/cell of HBM traffic; needs a third
        # velocity buffer because the reference's in-place result buffer is still an input of neighbouring tiles).  The first,
        # one-row form was issue-bound (544 us against 231 + 287 us for the two kernels at res 4096: three gradient rows
        # recomputed per output row); on 2-row register tiles, every row in one launch, it takes 373-394 us -> ON by default since round 2
        # (fused_transport=False or FS_FUSE_TRANSPORT=0 gives the reference's two launches and its intermediate buffers).
        if fused_transport is None:
            fused_transport = os.environ.get("FS_FUSE_TRANSPORT", "1") == "1"
        self._fused_transport = (bool(fused_transport) and self.resolution[0] % 2 == 0 and self._dev.dtype == np.float32
                                 and os.environ.get("FS_MARCH", "1") != "0")      # (f64: 256 VGPRs per tile - the two-kernel form)
        self._v_spare = self._dev.alloc(2) if self._fused_transport else None
        # ... and K2 in the same call (fs_cip_step; FS_FUSE_K2=0 / fused_k2=False: K2 as its own launch everywhere)
        self._fused_k2 = self._fused_transport and (os.environ.get("FS_FUSE_K2", "2") != "0" if fused_k2 is None else bool(fused_k2)) and hasattr(self._dev, "cip_step")

    def _flow_step(self):
        self._bc.set_velocity_boundary_condition(self.v.current)
        self._update_velocities(self.v, self.vx, self.vy, self.p)
        if self.vorticity_confinement is not None:
            self.vorticity_confinement.apply(self.v)
            self.v.swap()
        self.pressure_updater.update(self.p, self.v.current)
        spare_too = self._v_spare is not None and self._v_spare.static_id != 0
        limit_field(self.v.current, VELOCITY_LIMIT, defer=not spare_too)
        if spare_too:
            # limit_field is the one kernel that rewrites wall cells nothing else touches (uploaded data above the limit).  The fused
            # transport pass relies on those cells being EQUAL in v.current and the spare buffer it writes next (it carries only what
            # some kernel writes): limit the spare's copy of them the same way.  Only in runs that uploaded a velocity field.
            limit_field(self._v_spare, VELOCITY_LIMIT)

    def update(self):
        self._flow_step()

    def get_fields(self):
        return self.v.current, self.p.current

    def _set_grad(self, fx, fy, f):
        self._dev.cip_set_grad(self.dx, fx, fy, f)

    def _update_velocities(self, v, vx, vy, p):
        grads = (vx.current, vx.next, vy.current, vy.next)
        fused34 = self._fused_transport and not any(f.user_data for f in grads)
        if fused34 and self._fused_k2:
            # K2 + K3 + K4 as one call (include/fs_hip.h fs_cip_step): on large single-GPU grids the tiles that see nothing but fluid evaluate
            # K2 in registers on the way (csrc/fs_k234.h) - v.next then holds the post-K2 velocity only where something reads it before the
            # reference's own sequence overwrites it (every fluid cell: this step's vorticity confinement, or the next step's K2)
            full = self._v_spare.static_id != v.current.static_id
            self._dev.cip_step(self.dt, self.dx, self.re, self._v_spare, vx.next, vy.next, v.next, v.current, p.current, vx.current, vy.current, full=full)
            self._v_spare.static_id = v.current.static_id
            v.current, self._v_spare = self._v_spare, v.current
            vx.swap()
            vy.swap()
            return
        self._non_advection_phase(v.next, v.current, p.current)
        if fused34:
            # one pass instead of K3 + swap + K4 + swap.  End state as in the reference: v.current = advected velocity with the
            # pre-K2 values on non-fluid cells, v.next = post-K2 velocity, vx/vy.current = new gradients (their .next: dead data)
            # (the pass carries only the cells some kernel writes; after an upload into v.current or the spare buffer - Field.static_id -
            #  the two may differ anywhere: one pass then carries every cell)
            full = self._v_spare.static_id != v.current.static_id
            self._dev.cip_grad_advect(self.dt, self.dx, self._v_spare, vx.next, vy.next, v.next, v.current, vx.current, vy.current, full=full)
            self._v_spare.static_id = v.current.static_id
            v.current, self._v_spare = self._v_spare, v.current
            vx.swap()
            vy.swap()
            return
        self._non_advection_phase_grad(vx.next, vy.next, vx.current, vy.current, v.current, v.next)
        for buf in (v, vx, vy):
            buf.swap()
        self._advection_phase(v.next, vx.next, vy.next, v.current, vx.current, vy.current, v.current)
        for buf in (v, vx, vy):
            buf.swap()

    def _non_advection_phase(self, fn, fc, pc):
        self._dev.cip_nonadv(self.dt, self.dx, self.re, fn, fc, pc)

    def _non_advection_phase_grad(self, fxn, fyn, fxc, fyc, fc, fn):
        self._dev.cip_nonadv_grad(self.dx, fxn, fyn, fxc, fyc, fc, fn)

    def _advection_phase(self, fn, fxn, fyn, fc, fxc, fyc, v):
        self._dev.cip_advect(self.dt, self.dx, fn, fxn, fyn, fc, fxc, fyc, v)


class DyeCipMacSolver(CipMacSolver):
    """CipMacSolver + CIP-advected dye with its own gradient fields (fs/solver.py:335-401)."""

    def __init__(self, boundary_condition, pressure_updater, dt, dx, re, vorticity_confinement=None, fused_transport=None, fused_k2=None):
        super().__init__(boundary_condition, pressure_updater, dt, dx, re, vorticity_confinement, fused_transport, fused_k2)
        res = boundary_condition.get_resolution()
        self.dye = DoubleBuffer(res, 3, self._dev)
        self.dyex = DoubleBuffer(res, 3, self._dev)
        self.dyey = DoubleBuffer(res, 3, self._dev)
        self._set_grad(self.dyex.current, self.dyey.current, self.dye.current)
        # clamp_field(dye, 0, 1) folded into the advection store + a clamp of the inflow cells: after a step only fluid cells
        # (advected) and inflow cells (rewritten by the dye BC) can leave [0, 1]; walls / outflow cells of this buffer are never
        # written.  Holds for device-initialised buffers; user-uploaded dye data falls back to the full-grid clamp.
        self._fused_clamp = self.resolution[0] % 4 == 0 and os.environ.get("FS_MARCH", "1") != "0"
        # K3 + K4 of the dye in one pass as well (f32; a third dye buffer rotates like the velocity's): 355 + 459 us -> one launch
        self._fused_dye = (self._fused_transport and self._dev.dtype == np.float32)
        self._dye_spare = self._dev.alloc(3) if self._fused_dye else None

    def update(self):
        self._flow_step()
        self._bc.set_dye_boundary_condition(self.dye.current, self.v.current)
        fold = self._fused_clamp and not (self.dye.current.user_data or self.dye.next.user_data)
        self._update_dye(self.dye, self.dyex, self.dyey, self.v, clamp=fold)
        if fold:
            self._dev.clamp_inflow(0.0, 1.0, self.dye.current, defer=True)
        else:
            clamp_field(self.dye.current, 0.0, 1.0)
            if self._dye_spare is not None:      # (wall cells of an uploaded dye: clamped in both copies, see _flow_step)
                clamp_field(self._dye_spare, 0.0, 1.0)

    def get_fields(self):
        return self.v.current, self.p.current, self.dye.current

    def _non_advection_phase_dye(self, dn, dc):
        self._dev.cip_nonadv_dye(self.dt, self.dx, self.re, dn, dc)

    def _update_dye(self, dye, dyex, dyey, v, clamp=False):
        fused = self._fused_dye and not any(f.user_data for f in (dyex.current, dyex.next, dyey.current, dyey.next))
        if fused and self._fused_k2:
            # K12 + K3 + K4 as one call (fs_cip_step_dye): as the velocity's fs_cip_step - dye.next holds the dye after its non-advection phase only
            # where something reads it before the next step's K12 rewrites it
            full = self._dye_spare.static_id != dye.current.static_id
            self._dev.cip_step_dye(self.dt, self.dx, self.re, self._dye_spare, dyex.next, dyey.next, dye.next, dye.current,
                                   dyex.current, dyey.current, v.current, clamp01=clamp, full=full)
            self._dye_spare.static_id = dye.current.static_id
            dye.current, self._dye_spare = self._dye_spare, dye.current
            dyex.swap()
            dyey.swap()
            return
        self._non_advection_phase_dye(dye.next, dye.current)
        if fused:
            # one pass instead of K3 + swap + K4 + swap.  End state as in the reference: dye.current = advected dye with the previous
            # values on non-fluid cells, dye.next = the dye after its non-advection phase, dyex / dyey.current = new gradients (.next: dead)
            full = self._dye_spare.static_id != dye.current.static_id
            self._dev.cip_grad_advect_dye(self.dt, self.dx, self._dye_spare, dyex.next, dyey.next, dye.next, dye.current,
                                          dyex.current, dyey.current, v.current, clamp01=clamp, full=full)
            self._dye_spare.static_id = dye.current.static_id
            dye.current, self._dye_spare = self._dye_spare, dye.current
            dyex.swap()
            dyey.swap()
            return
        self._non_advection_phase_grad(dyex.next, dyey.next, dyex.current, dyey.current, dye.current, dye.next)
        for buf in (dye, dyex, dyey):
            buf.swap()
        if clamp:
            self._dev.cip_advect_dye_clamped(self.dt, self.dx, dye.next, dyex.next, dyey.next, dye.current, dyex.current, dyey.current, v.current)
        else:
            self._advection_phase(dye.next, dyex.next, dyey.next, dye.current, dyex.current, dyey.current, v.current)
        for buf in (dye, dyex, dyey):
            buf.swap()
