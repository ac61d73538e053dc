"""hipGraph replay (what bench.py times): a captured pair of step() calls, replayed, must advance the simulation exactly
like the same number of eager steps - otherwise the timed region would not be doing the work it claims."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bc,scheme,vc,dye,updater", [(5, "cip", 5.0, False, None), (3, "kk", 10.0, False, None),
                                                      (2, "cip", 5.0, True, None), (2, "cip", None, False, ("jacobi", 7)),
                                                      (1, "upwind", None, False, None)])
def test_graph_replay_equals_eager(bc, scheme, vc, dye, updater, hip_lib):
    import fs
    res = 128
    dt, dx, re = 0.05 / res, 1.0 / res, 1e6
    fs.runtime.init(gpu=0, dtype="f32")
    cls = fs.DyeFluidSimulator if dye else fs.FluidSimulator
    eager = cls.create(bc, res, dt, dx, re, vc, scheme, pressure_updater=updater)
    graph = cls.create(bc, res, dt, dx, re, vc, scheme, pressure_updater=updater)
    dev = graph._solver._bc.device
    try:
        for _ in range(4):                       # same warm-up on both
            eager.step(); graph.step()
        done = graph.capture_period(budget=24)    # one period of the buffer rotation (2 or 6 steps); the captured steps are executed once
        assert graph._graph is not None
        _, gid, period = graph._graph
        assert period in (1, 2, 6) and done <= 16
        dev.replay(gid, 3)
        for _ in range(done + 3 * period):
            eager.step()
        a, b = eager.field_to_numpy(), graph.field_to_numpy()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        assert float(np.abs(a["p"]).max()) > 0
        # and the python-side buffer references are back in place: further eager steps keep agreeing
        eager.step(); eager.step(); graph.step(); graph.step()
        a, b = eager.field_to_numpy(), graph.field_to_numpy()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    finally:
        eager._solver._bc.device.close()
        dev.close()
