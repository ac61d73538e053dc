#!/usr/bin/env python3
"""Compute-side cost of ONE slab of an N-way y-cut, timed on a single GPU (ghost-row exchanges replaced by no-ops, so the
ghost rows hold stale data - timing only, results are meaningless).  Predicts the per-step GPU time of `bench.py --gpus N`
minus communication, for choosing the halo depth.   usage: slab_cost.py [res] [world] [rank] [halo ...]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("2d-fluid-simulator_amd")
import fs  # noqa: E402
from fs import _lib  # noqa: E402
from fs.boundary_condition import BoundaryCondition, create_scene_arrays  # noqa: E402
from fs.runtime import Device, DeviceBase  # noqa: E402


class LoneSlab(Device):
    def __init__(self, nx, ny, dtype, rank, world, halo):
        DeviceBase.__init__(self, nx, ny, dtype, 0, rank, world, halo, None, None)
        self._lib = _lib.load()
        ctx = ctypes.c_void_p()
        _lib.call("fs_create", ctypes.byref(ctx), 0, self.nx, self.ny, 0, self.y0, self.nyl, self.halo)
        self._ctx = ctx
        self._graphs = []

    def _p_exchange(self, h, nchan, depth):
        pass

    def _p_max_over_ranks(self, values):
        return list(values)

    def _p_exchange_many(self, handles, depth):
        pass

    def _p_exchange_begin(self, handles, depth):
        pass

    def _p_exchange_wait(self):
        pass

    def _p_exchange_mark(self):
        pass


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
    halos = [int(a) for a in sys.argv[4:]] or [2, 4, 8, 16]
    const, mask, _ = create_scene_arrays(5, res)
    dt, dx = 0.05 / res, 1.0 / res
    for halo in halos:
        dev = LoneSlab(mask.shape[0], mask.shape[1], np.float32, rank, world, halo)
        bc = BoundaryCondition(const, mask, device=dev)
        vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
        pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2)
        solver = fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc)
        for _ in range(40):
            solver.update()
        dev.sync()
        n0, b0, t0 = dev.n_exchanges, dev.n_exchanged_bytes, time.perf_counter()
        steps = 400
        for _ in range(steps):
            solver.update()
        dev.sync()
        el = time.perf_counter() - t0
        print(f"res {res} slab {rank}/{world} halo {halo}: {el / steps * 1e6:.1f} us/step compute, "
              f"{(dev.n_exchanges - n0) / steps:.2f} exchanges/step, {(dev.n_exchanged_bytes - b0) / steps / 1024:.0f} KB/step/neighbour (no-op here)", flush=True)
        dev.profile(True); dev.profile_reset()
        for _ in range(60):
            solver.update()
        dev.sync()
        rep = dev.profile_report()
        print("    per step (HIP events, us): " + "  ".join(f"{k}={ms / 60 * 1e3:.1f}x{n / 60:g}" for k, (n, ms) in rep.items())
              + f"   sum {sum(ms for _, ms in rep.values()) / 60 * 1e3:.1f}", flush=True)
        dev.close()


if __name__ == "__main__":
    main()
