#!/usr/bin/env python3
"""One slab of the 8-way cut of bc5 res 4096 on ONE GPU with its ghost-row exchanges going through the real RCCL path in
loop-back (the rank is its own neighbour: same calls, same streams, copies stay on the GPU): per-step time with blocking
exchanges, with overlapped exchanges, and with the exchanges removed.  usage: overlap_bench.py [halo ...]"""
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


class LoopbackSlab(Device):
    def __init__(self, nx, ny, rank, world, halo, mode):
        DeviceBase.__init__(self, nx, ny, np.float32, 0, rank, world, halo, None, None)
        os.environ["FS_OVERLAP"] = "1" if mode in ("overlap", "blocking-commstream") else "0"      # fs_comm_init: communication stream or in line
        self.overlap = mode == "overlap"
        self.mode = mode
        self._lib = _lib.load()
        ctx = ctypes.c_void_p()
        _lib.call("fs_create", ctypes.byref(ctx), 0, self.nx, self.ny, 0, self.y0, self.nyl, self.halo)
        self._ctx = ctx
        self._graphs = []
        uid = ctypes.create_string_buffer(128)
        _lib.call("fs_comm_unique_id", uid)
        saved = os.dup(1); os.dup2(2, 1)
        _lib.call("fs_comm_init", ctx, 0, 1, ctypes.c_char_p(uid.raw))
        ctypes.CDLL(None).fflush(None); os.dup2(saved, 1); os.close(saved)
        _lib.call("fs_comm_loopback", ctx, 1)

    def _p_max_over_ranks(self, values):
        return list(values)

    def _p_exchange_many(self, handles, depth):
        if self.mode != "none":
            super()._p_exchange_many(handles, depth)

    def _p_exchange(self, h, nchan, depth):
        if self.mode != "none":
            super()._p_exchange(h, nchan, depth)


def main():
    res, world, rank = 4096, int(os.environ.get("OB_WORLD", "8")), int(os.environ.get("OB_RANK", "3"))      # (OB_WORLD=2 OB_RANK=1: the upper half)
    const, mask, _ = create_scene_arrays(5, res)
    dt, dx = 0.05 / res, 1.0 / res
    modes = os.environ.get("OB_MODES", "none,blocking-commstream,blocking,overlap,tape").split(",")
    for halo in [int(a) for a in sys.argv[1:]] or [4, 8, 16]:
        for mode in modes:
            dev = LoopbackSlab(mask.shape[0], mask.shape[1], rank, world, halo, mode)
            bc = BoundaryCondition(const, mask, device=dev)
            solver = fs.CipMacSolver(bc, fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2), dt, dx, 1e6, fs.VorticityConfinement(bc, dt, dx, 5.0))
            for _ in range(40):
                solver.update()
            tape = dev.tape_period(solver.update, nsteps=2) if mode == "tape" else None
            dev.sync()
            n0, t0, steps = dev.n_exchanges, time.perf_counter(), 400
            if tape is not None:
                dev.replay_tape(tape, steps // tape["nsteps"])
            else:
                for _ in range(steps):
                    solver.update()
            dev.sync()
            el = time.perf_counter() - t0
            print(f"halo {halo:2d} {mode:9s} all={int(dev.exchange_all)} pair={int(getattr(solver.pressure_updater, '_pair', False))}: {el / steps * 1e6:7.1f} us/step   {(dev.n_exchanges - n0) / steps:.2f} exchanges/step, "
                  f"{dev.n_overlapped / max(dev.n_exchanges, 1) * 100:.0f} % overlapped", flush=True)
            dev.close()


if __name__ == "__main__":
    main()
