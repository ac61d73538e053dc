"""FluidSimulator.run() on slabs takes exactly the steps it was asked for: the period search of the command tape executes the
steps it logs, and with its default of 14 two-step blocks a run(24 .. 27) that found no period used to take 28.  Host logic only: a
stand-in device counts the steps."""
import importlib
import sys

import pytest

import conftest  # noqa: F401  (paths)

importlib.import_module("2d-fluid-simulator_amd")
from fs.fluid_simulator import FluidSimulator  # noqa: E402


class _Field:
    serial, user_data, static_id = 1, False, 0


class _Dev:
    nranks = 2

    def __init__(self, find_after=None):
        self.find_after = find_after      # the search "finds" a 2-step period after this many blocks (None: never)
        self.replayed = 0

    def alloc(self, n):
        return _Field()

    def _state_signature(self):
        return ()

    def tape_period(self, step_fn, nsteps=2, tries=14, **kw):
        for b in range(tries):
            for _ in range(nsteps):
                step_fn()
            if self.find_after is not None and b + 1 >= self.find_after:
                return {"nsteps": 2, "id": None}
        return None

    def replay_tape(self, tape, times):
        self.replayed += max(times, 0) * tape["nsteps"]

    def free_tape(self, tape):
        pass


class _Solver:
    def __init__(self, dev):
        self._dev = dev
        self.updates = 0
        self.pressure_updater = type("PU", (), {"_spare": None})()

    def update(self):
        self.updates += 1


@pytest.mark.parametrize("n", [23, 24, 25, 26, 27, 28, 29, 40, 41])
@pytest.mark.parametrize("find_after", [None, 3, 14])
def test_run_takes_exactly_n_steps(n, find_after):
    dev = _Dev(find_after)
    solver = _Solver(dev)
    sim = FluidSimulator(solver)
    sim.run(n)
    assert solver.updates + dev.replayed == n, f"run({n}) took {solver.updates} eager + {dev.replayed} replayed steps"
    sim.run(n)      # a second chunk (cached tape or another fruitless search) must not overshoot either
    assert solver.updates + dev.replayed == 2 * n


def test_tape_shape_agreement_does_not_alias():
    """ranks agree on (len(prologue), len(ops)) as separate elements - packed as p * 1000 + n, a tape of (1, 0) and one of
    (0, 1000) operations looked the same, and values above 2^20 broke the min/max trick."""
    from fs.runtime import DeviceBase

    class Two(DeviceBase):
        def __init__(self, mine, other):
            self.nranks, self.mine, self.other = 2, mine, other

        def _p_max_over_ranks(self, values):        # element-wise maximum over "both ranks"
            assert list(values) == [x for v in self.mine for x in (v, self._BIG - v)]
            theirs = [x for v in self.other for x in (v, self._BIG - v)]
            return [max(a, b) for a, b in zip(values, theirs)]

    assert Two([0, 1000], [0, 1000])._p_same_over_ranks([0, 1000])
    assert not Two([1, 0], [0, 1000])._p_same_over_ranks([1, 0])
    assert not Two([0, 1000], [1, 0])._p_same_over_ranks([0, 1000])
    assert Two([3, 2_500_000], [3, 2_500_000])._p_same_over_ranks([3, 2_500_000])        # far beyond 2^20
    assert not Two([3, 2_500_000], [3, 2_500_001])._p_same_over_ranks([3, 2_500_000])
    d = Two([7], [9])
    d._p_max_over_ranks = lambda values: [max(values[0], 9), max(values[1], d._BIG - 9)]
    assert d._p_min_max_over_ranks(7) == (7, 9)


class _GraphDev:
    """Single-GPU stand-in with hipGraph capture: a capture runs the host side of the steps (nothing executes), a replay executes them."""
    nranks = 1

    def __init__(self, solver_ref):
        self.solver_ref, self.graphs, self.replayed, self.freed = solver_ref, {}, 0, []

    def alloc(self, n):
        return _Field()

    def capture(self, fn):
        s = self.solver_ref[0]
        before = s.updates
        fn()
        steps, s.updates = s.updates - before, before      # captured, not executed
        gid = len(self.graphs)
        self.graphs[gid] = steps
        return gid

    def replay(self, gid, times=1):
        self.replayed += self.graphs[gid] * max(times, 0)

    def free_graph(self, gid):
        self.freed.append(gid)


@pytest.mark.parametrize("n", [1, 7, 15, 16, 17, 18, 33, 34, 35, 64, 100, 101])
def test_run_with_graphs_takes_exactly_n_steps_and_prefers_the_long_graph(n):
    """Round 4: capture_period also captures the period repeated to >= 16 steps (a replay costs ~5 us of GPU idle time whatever it holds);
    run() replays that one first, then single periods, then steps eagerly - n steps in total, whatever n."""
    ref = [None]
    dev = _GraphDev(ref)
    solver = _Solver(dev)
    ref[0] = solver
    sim = FluidSimulator(solver)
    sim.run(n)
    assert solver.updates + dev.replayed == n, (solver.updates, dev.replayed)
    if n >= 34:          # 1 eager step + the period (1 step here) + its long form (16 steps) fit the first chunk
        assert sim._graph is not None and sim._graph_long is not None and sim._graph_long[1] >= 16 and sim._graph_long[1] % sim._graph[2] == 0
    sim.run(n)
    assert solver.updates + dev.replayed == 2 * n
    if sim._graph_long is not None and n >= 16:
        assert dev.replayed >= 16          # the long graph did the bulk
