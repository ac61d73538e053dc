"""Pressure-Poisson relaxation (reference: fs/pressure_updater.py).

predict_p (fs/pressure_updater.py:23-38) is evaluated inside the HIP sweep kernels.  Two equivalent
kernel forms exist: reading v like the reference, or reading a per-step precomputed source pair (the
source term depends only on v, which is constant over the sweeps of one step).  Both keep the
reference's operation order and give bit-identical pressures; `precompute_source` only trades one
extra pass per step for cheaper sweeps.
"""
import os
from abc import ABCMeta, abstractmethod


class PressureUpdater(metaclass=ABCMeta):
    def __init__(self, boundary_condition, dt, dx):
        self._bc = boundary_condition
        self._dev = boundary_condition.device
        self.dt = dt
        self.dx = dx

    @abstractmethod
    def update(self, p, v_current):
        pass


class JacobiPressureUpdater(PressureUpdater):
    """Jacobi method: n_iter x { pressure BC on p.current; p.next <- predict_p(p.current) on not-wall cells; swap }
    (fs/pressure_updater.py:41-66)."""

    def __init__(self, boundary_condition, dt, dx, n_iter, precompute_source=None, lazy_bc=None):
        super().__init__(boundary_condition, dt, dx)
        self._n_iter = n_iter
        # one extra pass per step (~110 us at res 4096) buys cheaper sweeps (76 vs 87 us): worth it from 9 sweeps on - from 6 on where
        # two-sweep passes are in use (decided below): 110 + 2 x 98 + 2 x 82 against 6 x 93 us
        tentative = precompute_source is None and 6 <= n_iter < 9 and self._dev.lazy_bc_ok
        self._precompute = (n_iter >= 9 or tentative) if precompute_source is None else bool(precompute_source)
        self._src = self._dev.alloc(2) if self._precompute else None
        # Long runs: all but the last two sweeps evaluate the pressure boundary condition on the fly from the raw output of the
        # previous sweep instead of launching the boundary kernel in between (same bits; the last two sweeps run the real kernel,
        # which leaves both p buffers exactly as the reference's n x (BC, sweep, swap) does).  Needs the source-pair form, a mask
        # that admits it (fs_lazy_bc_ok) and a p that no user upload has put into an arbitrary state.
        self._lazy = (self._precompute and n_iter >= 6 and self._dev.lazy_bc_ok) if lazy_bc is None else \
            (bool(lazy_bc) and self._precompute and n_iter >= 3 and self._dev.lazy_bc_ok)
        # Two sweeps per launch (fs_jacobi_pair_lazy): bc2 res 1600 +50 %, res 4096 +51 %, bc5 res 4096 +30 %, bc1 res 2048 +27 %; a mask
        # full of small obstacles sends too many rows down its general path (bc3 res 1000: 31 % of the rows, -6 %).  Same bits either
        # way, so a single-GPU run times the three forms on this mask once (FS_JACOBI_PAIRS=0 / 1 / 2 decides instead; 2 = vertical recipes in the tiles too).
        want = os.environ.get("FS_JACOBI_PAIRS", "auto")
        self._pairs = (self._lazy and want != "0" and n_iter >= 6
                       and (self._dev.nranks == 1 or 2 * max(2, 1 + self._dev.bc_radius_p) <= self._dev.halo))   # a pass reaches 4 rows
        # The tiles of the two-sweep pass hand rows with floors / ceilings / corners nearby to a general path (one row per workgroup); the
        # variant whose tiles also apply the vertical recipes keeps only thin walls there but costs 6 % where it is not needed.  Decided from
        # the mask, not from a timing run (round 2 timed three forms inside this constructor: the launch sequence then differed from run to
        # run): vertical when more than 5 % of the wave-tile rows would take the general path otherwise (bc3's cylinders: 31 %; bc2 / bc5: < 1 %).
        self._vertical = want == "2"
        self.form = "single sweeps"
        if self._pairs:
            if want == "auto" and hasattr(self._dev, "lazy_flags"):
                flags, (general, general_v) = self._dev.lazy_flags()
                self._vertical = general > 0.05 * max(flags.size, 1) and general_v < general
            self.form = "two sweeps per pass" + (", vertical recipes in the tiles" if self._vertical else "")
        elif self._lazy:
            self.form = "single sweeps, boundary condition in the sweep"
        # Four sweeps per pass (fs_jacobi_quad_lazy) where the mask admits it: the passes of a long run are launch- and latency-bound
        # (BASELINE configs[1]: 24 two-sweep passes of 21.6 us = 78 % of the step)
        self._quads = (self._pairs and n_iter >= 10 and os.environ.get("FS_JACOBI_QUADS", "1") == "1" and getattr(self._dev, "jacobi_quad_ok", False))
        if self._quads:
            self.form = "four sweeps per pass"
        if tentative and not self._pairs:
            self._precompute, self._src, self._lazy = False, None, False
        # The last two rounds - 2 x (boundary kernel + sweep), which leave both buffers as the reference does - in ONE pass into a third
        # buffer (fs_jacobi_finish; bc2 res 1600: 47 -> 25 us of a 585 us step).  Single GPU, masks that admit the four-sweep pass.
        self._finish = self._quads and self._dev.nranks == 1
        self._spare = (self._dev.alloc(1),) if self._finish else None      # (a tuple, like RedBlackSorPressureUpdater's: FluidSimulator._signature)

    def update(self, p, v_current):
        if self._precompute:
            self._dev.poisson_source(self.dt, self.dx, self._src, v_current)
        n_lazy = self._n_iter - 2 if self._lazy else 0
        n_real = self._n_iter - n_lazy
        if self._quads and p.current.static_id == p.next.static_id:
            # four sweeps per pass (the pass writes not-wall cells only: the wall cells nothing writes must be equal in the two buffers,
            # Field.static_id); what does not fill a pass runs as single lazily-bounded sweeps
            for _ in range(n_lazy // 4):
                self._dev.jacobi_quad_lazy(p.next, p.current, self._src)
                p.swap()
            n_lazy %= 4
            n_pairs = n_lazy // 2        # (equal wall histories: a single two-sweep pass is as good as an even number of them)
        else:
            # two sweeps per pass where possible; an even number of passes, so that every iterate lands in the physical buffer the
            # reference's rotation puts it in (the buffers differ in the wall cells nothing ever writes)
            n_pairs = (n_lazy // 4) * 2 if self._pairs else 0
        for k in range(n_pairs):
            self._dev.jacobi_pair_lazy(p.next, p.current, self._src, swapped=bool(k & 1), vertical=self._vertical)
            p.swap()
        for _ in range(n_lazy - 2 * n_pairs):
            self._dev.jacobi_sweep_lazy(p.next, p.current, self._src)
            p.swap()
        if (self._finish and n_real == 2 and self._n_iter - 2 >= 4
                and p.current.static_id == p.next.static_id == self._spare[0].static_id and not (p.current.user_data or p.next.user_data)):
            # (the pass stores the cells some kernel writes; the rest must agree in the three buffers: no uploads / fills in between)
            self._dev.jacobi_finish(self._spare[0], p.next, p.current, self._src)
            p.current, self._spare = self._spare[0], (p.current,)
            return
        for _ in range(n_real):
            self._bc.set_pressure_boundary_condition(p.current)
            self._update(p.next, p.current, v_current)
            p.swap()

    def _update(self, p_next, p_current, v_current):
        if self._precompute:
            self._dev.jacobi_sweep_src(p_next, p_current, self._src)
        else:
            self._dev.jacobi_sweep(self.dt, self.dx, p_next, p_current, v_current)


class RedBlackSorPressureUpdater(PressureUpdater):
    """Red-black SOR (fs/pressure_updater.py:69-114): per iteration the odd cells are relaxed from p.current
    into p.next, then the even cells are relaxed IN PLACE on p.next (blending with that buffer's stale
    value), then the buffers swap.  Not a textbook single-buffer SOR - reproduced literally."""

    def __init__(self, boundary_condition, dt, dx, relaxation_factor, n_iter, precompute_source=False, fused=True, pair=None):
        super().__init__(boundary_condition, dt, dx)
        self._n_iter = n_iter
        self._relaxation_factor = relaxation_factor
        # fused: odd + even pass of one iteration in a single kernel (same bits, fewer bytes)
        fast = fused and not precompute_source and os.environ.get("FS_MARCH", "1") != "0"
        self._fused = fast and boundary_condition.get_resolution()[0] % 2 == 0          # (lanes of 2 cells: any even width, like the pair pass below)
        self._precompute = bool(precompute_source)
        self._src = self._dev.alloc(2) if self._precompute else None
        # pair: TWO iterations and the two boundary passes between them in one pass over HBM (csrc/fs_rbpair.h; same bits, 25 instead of
        # 2 x 21 B per fluid cell and two launches less).  Out of place: the pressure rotates through a second pair of buffers.  Needs a
        # mask that admits it (Device.rb_pair_ok: no one-cell-thin walls between fluid regions), f32, and on slabs a halo of 4 rows.
        # Slab runs: the pass wants v and p.current 4 rows deep at the END of the step, where the ghost rows are at their shallowest.  At halo 16
        # the tracker then exchanges 0.6 times per step in a 5-step pattern that, times the 6-step buffer rotation, no tape can hold; at halo 20
        # (the default for slabs of 160 rows and more since round 4) the pattern is one exchange every 2 steps with and without the pass
        # (tools/slab_period.py) and the pass is on - loop-back, middle slab of the 8-way cut of bc5 res 4096: 131-132 us per step against
        # 133-134 at halo 16 without it (profiles/r3_loopback_slab_step.txt).  Shallower halos keep the single iterations.
        if pair is None:
            pair = os.environ.get("FS_RBSOR_PAIR", "1" if (self._dev.nranks == 1 or self._dev.halo >= 20) else "0") == "1"
        self._pair = (bool(pair) and fast and n_iter >= 2 and getattr(self._dev, "rb_pair_ok", False)
                      and (self._dev.nranks == 1 or self._dev.halo >= 4))
        self._spare = (self._dev.alloc(1), self._dev.alloc(1)) if self._pair else None

    def update(self, p, v_current):
        if self._precompute:
            self._dev.poisson_source(self.dt, self.dx, self._src, v_current)
        n = self._n_iter
        while self._pair and n >= 2:
            c_out, n_out = self._spare
            # the pass stores fluid cells and boundary targets only: every other cell must already be equal in the buffer it reads and
            # the one it writes (Field.static_id).  After an upload / fill into one of them the pass carries every cell once.
            full = c_out.static_id != p.current.static_id or n_out.static_id != p.next.static_id
            self._dev.rbsor_pair(self.dt, self.dx, self._relaxation_factor, c_out, n_out, p.current, p.next, v_current, full=full)
            c_out.static_id, n_out.static_id = p.current.static_id, p.next.static_id
            self._spare = (p.current, p.next)
            p.current, p.next = c_out, n_out
            n -= 2
        for _ in range(n):
            self._bc.set_pressure_boundary_condition(p.current)
            self._update(p.next, p.current, v_current)
            p.swap()

    def _update(self, p_next, p_current, v_current):
        if self._fused:
            self._dev.rbsor_iteration(self.dt, self.dx, self._relaxation_factor, p_next, p_current, v_current)
            return
        self._update_pressures_odd(p_next, p_current, v_current)
        self._update_pressures_even(p_next, p_next, v_current)

    def _half(self, parity, pn, pc, vc):
        if self._precompute:
            self._dev.rbsor_halfsweep_src(self._relaxation_factor, parity, pn, pc, self._src)
        else:
            self._dev.rbsor_halfsweep(self.dt, self.dx, self._relaxation_factor, parity, pn, pc, vc)

    def _update_pressures_odd(self, pn, pc, vc):
        self._half(1, pn, pc, vc)

    def _update_pressures_even(self, pn, pc, vc):
        self._half(0, pn, pc, vc)
