#!/usr/bin/env python3
"""Round 6: is the fast / slow state of a run a property of WHERE its fields live?  Three simulators of the headline workload alive at once in ONE process
(different allocations), each timed over 300 graph-replayed steps, round robin, three rounds: if a simulator keeps its rate across the rounds and the
simulators differ, the state belongs to the allocation; if all agree within a process and processes differ, it does not.   python tools/r6_placement.py"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("2d-fluid-simulator_amd")
import fs
from fs.boundary_condition import BoundaryCondition, create_scene_arrays

res = 4096
dt, dx = 0.05 / res, 1.0 / res
const, mask, _ = create_scene_arrays(5, res)
sims = []
for k in range(3):
    fs.runtime.init(gpu=0)
    bc = BoundaryCondition(const, mask)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
    sim = fs.FluidSimulator(fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc))
    sim.run(60); bc.device.sync()
    sims.append((sim, bc.device))
for rnd in range(3):
    row = []
    for sim, dev in sims:
        t0 = time.perf_counter(); sim.run(300); dev.sync(); t = time.perf_counter() - t0
        row.append(300 / t)
    print(f"round {rnd}: " + "  ".join(f"sim{k} {v:7.1f}" for k, v in enumerate(row)) + " steps/s", flush=True)
