#!/usr/bin/env python3
"""Does WHERE the fields live in HBM decide the box's fast / slow state?  The headline solver built after N dummy 268 MB allocations (which push its
fields to other physical pages), steps/s of 600 graph-replayed steps each.     python tools/r4_placement.py [counts ...]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("2d-fluid-simulator_amd")
import fs
from fs.boundary_condition import BoundaryCondition, create_scene_arrays

res = 4096
dt, dx = 0.05 / res, 1.0 / res
const, mask, _ = create_scene_arrays(5, res)
for count in [int(a) for a in sys.argv[1:]] or [0, 16, 100, 400, 0]:
    fs.runtime.init(gpu=0)
    bc = BoundaryCondition(const, mask)
    dev = bc.device
    dummies = [dev.alloc(2) for _ in range(count)]
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
    vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
    sim = fs.FluidSimulator(fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc))
    sim.run(60); dev.sync()
    t0 = time.perf_counter(); sim.run(600); dev.sync(); t = time.perf_counter() - t0
    print(f"{count:4d} dummy fields ({count * 0.268:6.1f} GB) before the solver's: {600 / t:7.1f} steps/s", flush=True)
    del sim, pu, vc, dummies
    dev.close()
