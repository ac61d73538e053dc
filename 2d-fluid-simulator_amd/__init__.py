"""2d-fluid-simulator_amd: MI355X-native implementation of the takah29/2d-fluid-simulator step() path.

The directory name is not a Python identifier, so load it with
    importlib.import_module("2d-fluid-simulator_amd")
which puts this directory on sys.path and exposes the drop-in package as plain `fs`
(`from fs.fluid_simulator import FluidSimulator`, as in the reference).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import fs  # noqa: E402  (the drop-in package, top-level on purpose: one copy per process)

runtime = fs.runtime
__all__ = ["fs", "runtime"]
