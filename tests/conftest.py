import os
import sys

import numpy as np
import pytest

# the oracle's OpenMP regions run on tiny grids here: a team of hundreds of threads costs far more than the work
os.environ.setdefault("OMP_NUM_THREADS", "8")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, os.path.join(REPO, "2d-fluid-simulator_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite (the driver runs `pytest -m gpu -x -q`: the first failure ends the run).  Parity evidence first, in the
# order of SURVEY.md 8c's chain - per-kernel golden vectors, golden trajectories, the fast launch forms against each other and the oracle,
# the BASELINE sizes, fuzzing, slabs / RCCL - and LAST the tests that start bench.py / the CLI as subprocesses (slow, and about the format of
# a report rather than about a value a kernel computed).  Unlisted modules keep their alphabetical place in the middle.
_ORDER = ["test_gpu_kernels", "test_gpu_traj", "test_gpu_cip_step", "test_gpu_rbpair", "test_gpu_jquad", "test_gpu_variants",
          "test_gpu_limit_gate", "test_gpu_lazy_bc", "test_gpu_odd_res", "test_gpu_random_masks", "test_vis", "test_f64div",
          "test_gpu_errors", "test_gpu_graph", "test_deferred_passes", "test_gpu_fullsize", "test_gpu_fuzz",
          "test_gpu_slab_threads", "test_gpu_rccl_loopback", "test_gpu_rccl_overlap"]
_LAST = ["test_cli", "test_gpu_bench_multirank"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _ORDER:
            return _ORDER.index(mod)
        if mod in _LAST:
            return 1000 + _LAST.index(mod)
        return 500
    items.sort(key=rank)      # (stable: the order inside a module stays)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  GPU tests must run the HIP path: a missing extension is an error, not a skip."""
    from fs import _lib
    return _lib.load()
