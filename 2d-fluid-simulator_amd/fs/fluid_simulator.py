"""Facade (reference: fs/fluid_simulator.py): FluidSimulator / DyeFluidSimulator.

`create()` composes boundary condition + vorticity confinement + red-black SOR(omega 1.3, 2 iterations)
+ solver exactly like the reference (fs/fluid_simulator.py:60-108, 129-176); `step()` is the hot path.
`pressure_updater=` is an extension (the reference hard-codes RB-SOR): ("jacobi", n_iter) or
("rbsor", omega, n_iter) or a ready PressureUpdater factory.
"""
from .advection import advect_kk_scheme, advect_upwind
from .boundary_condition import get_boundary_condition
from .pressure_updater import JacobiPressureUpdater, RedBlackSorPressureUpdater
from .solver import CipMacSolver, DyeCipMacSolver, DyeMacSolver, MacSolver
from .vorticity_confinement import VorticityConfinement

_WALL_COLOR = (0.5, 0.7, 0.5)   # fs/fluid_simulator.py:17 (applied by the visualisation kernels, csrc/fs_kernels.h k_visualize)


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
        self._dev = solver._dev
        self.rgb_buf = self._dev.alloc(3)      # image buffer (fs/fluid_simulator.py:16), device resident
        self._wall_color = _WALL_COLOR
        self._graph = None         # (signature, graph id, period) of the most recent capture
        self._graph_long = None    # (graph id, steps) of the same capture's long form: several periods in one graph (capture_period)
        self._graphs = {}          # signature -> (graph id, period, long form or None)
        self._tapes = {}           # (signature, ghost-row bookkeeping state) -> tape
        self._steps = 0
        self._eager_seen = False   # one step has run outside a capture (the library's compact launch lists exist)
        self._pending_after_step = (False, False)
        self._since_hot_check = 0

    def step(self):
        self._solver.update()
        self._eager_seen = True        # (captures call the solver directly: see capture_period)
        self._pending_after_step = self._limit_pending()
        self._since_hot_check += 1
        if self._since_hot_check >= 256:
            self._check_hot()

    def _check_hot(self):
        """Deferred limit passes are for runs that never need them.  Once a velocity buffer's flag is up (a speed above 9.95 - or one component above 7.04 - was stored; it stays
        up) the pass runs on every step, and as its own full-grid launch it is three times faster than inside a boundary launch at res 4096:
        looked at between launch sequences - at the start of run() / capture_period() and every 256 eager steps (one 12-byte download)."""
        dev = self._dev
        if getattr(dev, "capturing", False):
            return            # (a user capture of step() in progress: the look is a download - it waits for the next eager step)
        self._since_hot_check = 0
        if not getattr(dev, "limit_deferral", False):
            return
        s = self._solver
        target = getattr(getattr(s, "v", None), "current", None)      # the buffer limit_field is applied to at a step boundary - the others' flags do not
        if target is not None and dev.field_hot(target):             # decide whether a pass runs (a K3+K4 output carries word [3] until it is rewritten)
            bufs = [f for f in (target, getattr(getattr(s, "v", None), "next", None), getattr(s, "_v_spare", None)) if f is not None]
            dev.stop_limit_deferral(bufs)
            self._pending_after_step = self._limit_pending()

    def _limit_pending(self):
        """What the solver's fields owe: a deferred limit_field of the velocity (runtime.DeviceBase.limit_field; after a step of the plain
        solvers - the dye solvers' boundary kernel has taken it along), a deferred clamp_inflow of the dye (DeviceBase.clamp_inflow)."""
        v, dye = getattr(self._solver, "v", None), getattr(self._solver, "dye", None)
        return (v is not None and v.current.pending_limit is not None, dye is not None and dye.current.pending_clamp is not None)

    def _counted_step(self):
        self.step()
        self._steps += 1

    # -- replay of the step as a hipGraph (new; the reference steps from a Python GUI loop) --------------------------
    def _signature(self):
        """Identity of every device buffer behind the solver's DoubleBuffers + the flags that select kernel variants: a
        captured launch sequence is valid exactly while this is what it was at capture time."""
        s, sig = self._solver, []
        for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
            db = getattr(s, name, None)
            if db is not None:
                for f in (db.current, db.next):
                    # serial: a Field's identity for life (id() is recycled); static_id: which carry decisions the capture baked in
                    sig.append((f.serial, f.user_data, f.static_id, f.pending_limit, f.pending_clamp, f.bc_parity))      # (pending_limit: a deferred limit_field, runtime.DeviceBase.limit_field)
        for spare in (getattr(s, "_v_spare", None), getattr(s, "_dye_spare", None)) + tuple(getattr(s.pressure_updater, "_spare", None) or ()):
            sig.append((spare.serial, spare.static_id, spare.bc_parity) if spare is not None else 0)
        return tuple(sig)

    _LONG_STEPS = 16     # steps per long-form graph (capture_period)
    _MAX_CACHED = 8      # captured graphs / tapes kept per simulator (one per phase of the buffer rotation; the oldest is freed first)

    def run(self, nsteps, graph=True):
        """`nsteps` x step().  With graph=True (single GPU) the launches of a whole number of steps are captured once into a
        hipGraph and replayed, so no Python runs between kernels - for small grids (res 200: 8 kernels of ~3 us per step) that
        is the difference between launch-bound and GPU-bound.  Results are identical to calling step() nsteps times.

        A graph is valid for one phase of the solver's buffer rotation (self._signature()).  A chunk that is not a multiple of the
        period ends in another phase; graphs are therefore cached per phase (at most `period` of them exist) instead of being
        re-captured - and leaked - chunk after chunk, and a capture is only started when the chunk is long enough to pay for it."""
        dev = self._dev
        self._check_hot()
        if graph and dev.nranks > 1 and nsteps >= 24:
            # slab run: a hipGraph cannot carry the RCCL exchange; the period of the step (launches + exchanges) is logged
            # once and replayed from C++ instead (runtime.py tape_period / replay_tape)
            key = (self._signature(), dev._state_signature())
            tape = self._tapes.get(key)
            if tape is None:
                before = self._steps
                # the search executes the steps it logs: at most 2 * tries of them, never more than this call was asked for (ADVICE r3:
                # with the default of 14 blocks a run(24 .. 27) that found no period took 28 steps)
                tape = dev.tape_period(self._counted_step, nsteps=2, tries=min(14, nsteps // 2))
                nsteps -= self._steps - before
                assert nsteps >= 0, "tape_period executed more steps than run() was asked for"
                if tape is not None:
                    self._remember(self._tapes, (self._signature(), dev._state_signature()), tape, lambda t: dev.free_tape(t))
            if tape is not None:
                per = tape["nsteps"]
                dev.replay_tape(tape, nsteps // per)
                self._steps += nsteps - nsteps % per
                nsteps %= per
        if not graph or dev.nranks > 1 or not hasattr(dev, "capture"):
            for _ in range(nsteps):
                self._counted_step()
            return
        if nsteps > 0 and self._eager_seen and self._limit_pending() != self._pending_after_step:
            self._counted_step()        # (the cached graphs start from the state a step leaves behind, see capture_period)
            nsteps -= 1
        entry = self._graphs.get(self._signature())
        if entry is None and nsteps >= 16:          # (periods 1 - 6 need at most 16 steps to be found; 12 - an odd red-black count with the pair pass
            # on top of the fused transport - is only tried when the chunk has 28)
            nsteps -= self.capture_period(budget=nsteps)
            entry = self._graphs.get(self._signature())
        if entry is not None:
            gid, period, long = entry
            self._graph, self._graph_long = (self._signature(), gid, period), long
            if long is not None:
                dev.replay(long[0], nsteps // long[1])
                nsteps %= long[1]
            dev.replay(gid, nsteps // period)
            nsteps %= period
        for _ in range(nsteps):
            self.step()

    def _remember(self, cache, key, value, free):
        if key in cache:
            free(cache.pop(key))
        while len(cache) >= self._MAX_CACHED:
            free(cache.pop(next(iter(cache))))
        cache[key] = value

    def capture_period(self, budget=24):
        """Capture the launches of one PERIOD of the step into a hipGraph -> number of steps this took (they are executed).

        A replayed graph re-issues kernels on fixed device buffers, so it must cover a whole period of the solver's buffer
        rotation: 2 steps when every DoubleBuffer just swaps (two swaps restore it), 6 steps for the CIP solver with its fused
        gradient + advection pass and vorticity confinement (the velocity rotates through three buffers, the gradients and - with the
        two-iteration red-black pass - the pressure pairs through two).  Periods are tried in increasing order; a capture whose steps
        do not bring every buffer back to its place is still executed once (the host-side swaps have happened) and freed.  Leaves
        self._graph = (signature, graph id, period) - also cached for run() - or None if nothing within the budget repeats.

        Round 4: every replay of a graph costs ~5 us of GPU idle time whatever it holds (tools/r4_chain.py) - a quarter of a res-200 step
        when the graph is a 2-step period.  Where the budget allows, the same period is therefore captured a second time, repeated to
        >= 16 steps (self._graph_long = (graph id, steps)); run() and bench.py replay that one and finish with the short one."""
        dev, done = self._dev, 0
        self._check_hot()
        self._graph = self._graph_long = None
        if (not self._eager_seen or self._limit_pending() != self._pending_after_step) and budget >= 2:
            # (also: the steady state of a run with deferred limit passes starts every step with one pending - after a download there is
            #  none, and a period captured from there would not close)
            # libfs_hip builds the compact tile lists of a launch geometry the first time the geometry is launched EAGERLY (building one
            # synchronises the stream, which a capture forbids): a graph captured before any eager step would carry the dense launches for
            # life.  One eager step first - it counts as one of the steps this call takes.
            self.step()
            done = 1
        for period in (1, 2, 3, 4, 6, 12):
            if done + period > budget:
                break
            sig = self._signature()
            gid = dev.capture(lambda: [self._solver.update() for _ in range(period)])       # host-side swaps happen, nothing executes
            back = self._signature() == sig
            dev.replay(gid, 1)                                                    # now the captured steps run once
            done += period
            if back:
                long, reps = None, -(-self._LONG_STEPS // period)
                if reps > 1 and done + reps * period <= budget:
                    lid = dev.capture(lambda: [self._solver.update() for _ in range(reps * period)])
                    dev.replay(lid, 1)
                    done += reps * period
                    long = (lid, reps * period)
                self._remember(self._graphs, sig, (gid, period, long), lambda e: [dev.free_graph(e[0])] + ([dev.free_graph(e[2][0])] if e[2] else []))
                self._graph, self._graph_long = (sig, gid, period), long
                break
            dev.free_graph(gid)
        return done

    def field_to_numpy(self):
        fields = self._solver.get_fields()
        return {"v": fields[0].to_numpy(), "p": fields[1].to_numpy()}

    # -- visualisation (fs/fluid_simulator.py:22-58): device kernels; like the reference these return the image FIELD ----
    def get_norm_field(self):
        v, p = self._solver.get_fields()[:2]
        self._dev.vis_norm(self.rgb_buf, v, p)
        return self.rgb_buf

    def get_pressure_field(self):
        self._dev.vis_pressure(self.rgb_buf, self._solver.get_fields()[1])
        return self.rgb_buf

    def get_vorticity_field(self):
        self._dev.vis_vorticity(self._solver.dx, self.rgb_buf, self._solver.get_fields()[0])
        return self.rgb_buf

    @staticmethod
    def create(num, resolution, dt, dx, re, vor_eps, scheme, pressure_updater=None):
        return FluidSimulator(_compose(num, resolution, dt, dx, re, vor_eps, scheme, False, pressure_updater))


class DyeFluidSimulator(FluidSimulator):
    def field_to_numpy(self):
        fields = self._solver.get_fields()
        return {"v": fields[0].to_numpy(), "p": fields[1].to_numpy(), "dye": fields[2].to_numpy()}

    def get_dye_field(self):
        self._dev.vis_dye(self.rgb_buf, self._solver.get_fields()[2])
        return self.rgb_buf

    @staticmethod
    def create(num, resolution, dt, dx, re, vor_eps, scheme, pressure_updater=None):
        return DyeFluidSimulator(_compose(num, resolution, dt, dx, re, vor_eps, scheme, True, pressure_updater))
