#!/usr/bin/env python3
"""Isolated literal Jacobi sweep on the headline grid: span time per sweep (one event pair around 200 sweeps) per kernel form; checks the forms against each other bit for bit.
usage: r5_jac.py   (env FS_JACOBI_N2 / FS_JACOBI select the form)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("2d-fluid-simulator_amd")
import fs
res = 4096; dt, dx = 0.05 / res, 1.0 / res
fs.runtime.init(gpu=0)
sim = fs.FluidSimulator.create(5, res, dt, dx, 1e6, 5.0, "cip")
dev = sim._solver._bc.device
for _ in range(60): sim.step()
v, p = sim._solver.get_fields()
pa, pb = dev.alloc(1), dev.alloc(1)
pa.from_numpy(p.to_numpy())
for _ in range(4): dev.jacobi_sweep(dt, dx, pb, pa, v); dev.jacobi_sweep(dt, dx, pa, pb, v)
best = 1e9
for rep in range(5):
    dev.span_begin()
    for _ in range(100): dev.jacobi_sweep(dt, dx, pb, pa, v); dev.jacobi_sweep(dt, dx, pa, pb, v)
    best = min(best, dev.span_end() / 200 * 1e3)
mask = sim._solver._bc.mask
alg = mask.size + int((mask != 1).sum()) * 16
print(f"N2={os.environ.get('FS_JACOBI_N2','0')} FS_JACOBI={os.environ.get('FS_JACOBI','-')}: {best:7.2f} us/sweep  frac {alg / (best * 1e-6) / 8e12:.4f}  checksum {int(pa.to_numpy().view(np.uint32).astype(np.uint64).sum()):x}")
dev.close()
