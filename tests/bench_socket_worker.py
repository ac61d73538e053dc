"""Run bench.py's main() as rank R of W on ONE GPU (test harness, not product).

A 1-GPU box cannot form an RCCL communicator of W ranks (RCCL refuses duplicate devices), so bench.py's N > 1 control flow
(rendezvous, barriers, max-over-ranks timing, per-slab profiling, checksum / residual collectives, JSON assembly on rank 0)
would otherwise run for the first time on the driver's multi-GPU node.  Here every rank is its own process on GPU 0; the
ghost rows and the scalar collectives travel over local sockets instead of RCCL; kernels, slab geometry, the validity
tracker and bench.py itself are the product code.   usage: RANK= WORLD_SIZE= FS_FAKE_PORT= python bench_socket_worker.py <bench args>
"""
import ctypes
import importlib
import os
import sys
import time
from multiprocessing.connection import Client, Listener

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("2d-fluid-simulator_amd")
import fs  # noqa: E402
from fs import _lib, runtime  # noqa: E402

RANK, WORLD, PORT = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["FS_FAKE_PORT"])
KEY = b"fs-bench-test"


def _mesh():
    conns = {}
    listener = Listener(("127.0.0.1", PORT + RANK), authkey=KEY, backlog=64)     # (the default backlog of 1 drops the connections of 7 ranks arriving at once)
    for j in range(RANK):                       # connect to every lower rank
        t0 = time.time()
        while True:
            try:
                c = Client(("127.0.0.1", PORT + j), authkey=KEY)
                break
            except (ConnectionRefusedError, OSError):
                if time.time() - t0 > 120:
                    raise
                time.sleep(0.05)
        c.send(RANK)
        conns[j] = c
    for _ in range(RANK + 1, WORLD):            # accept every higher rank
        c = listener.accept()
        conns[c.recv()] = c
    return conns


CONNS = _mesh()


def _swap(peer, payload, first):
    """Exchange payloads with `peer`; the lower rank of a pair sends first (no deadlock on messages larger than the socket buffer)."""
    if first:
        CONNS[peer].send(payload)
        return CONNS[peer].recv()
    got = CONNS[peer].recv()
    CONNS[peer].send(payload)
    return got


class SocketSlabDevice(runtime.Device):
    def __init__(self, nx, ny, dtype, gpu=0, rank=0, nranks=1, halo=None, bcast=None, allgather=None):
        runtime.DeviceBase.__init__(self, nx, ny, dtype, gpu, rank, nranks, halo, bcast, allgather)
        self._lib = _lib.load()
        ctx = ctypes.c_void_p()
        _lib.call("fs_create", ctypes.byref(ctx), gpu, self.nx, self.ny, 0 if self.dtype == np.float32 else 1,
                  self.y0, self.nyl, self.halo)
        self._ctx = ctx
        self._graphs = []

    def _p_exchange(self, h, nchan, depth):
        self._p_exchange_many([(h, nchan, 0)], depth)

    def _p_exchange_many(self, handles, depth):
        # depth offsets [v, depth) of every field travel (v = rows the tracker still trusts), as in fs_halo_exchange_begin_partial
        H, n, r = self.halo, self.nyl, self.rank
        live = [(h, c, v) for h, c, v in handles if v < depth]
        lo = [self._p_download(h, c, H + v, depth - v) for h, c, v in live]
        hi = [self._p_download(h, c, H + n - depth, depth - v) for h, c, v in live]
        # pairs (even, even+1) first, then (odd, odd+1): every rank is in at most one pair per phase
        for phase in (0, 1):
            if r % 2 == phase and r + 1 < self.nranks:
                got = _swap(r + 1, hi, True)
                for (h, c, v), a in zip(live, got):
                    self._p_upload(h, c, np.ascontiguousarray(a), H + n + v, depth - v)
            elif r % 2 != phase and r - 1 >= 0:
                got = _swap(r - 1, lo, False)
                for (h, c, v), a in zip(live, got):
                    self._p_upload(h, c, np.ascontiguousarray(a), H - depth, depth - v)

    def _p_exchange_begin(self, handles, depth):     # no asynchronous transport here: the split of the kernel is still exercised
        self._p_exchange_many(handles, depth)

    def _p_exchange_wait(self):
        pass

    def _p_exchange_mark(self):
        pass

    def _p_max_over_ranks(self, values):
        vals = np.asarray(values, dtype=np.float64)
        if self.rank == 0:
            for j in range(1, self.nranks):
                vals = np.maximum(vals, CONNS[j].recv())
            for j in range(1, self.nranks):
                CONNS[j].send(vals)
        else:
            CONNS[0].send(vals)
            vals = CONNS[0].recv()
        return vals.tolist()

    def _p_allreduce(self, values):
        self.sync()
        vals = np.asarray(values, dtype=np.float64)
        if self.rank == 0:
            for j in range(1, self.nranks):
                vals = vals + CONNS[j].recv()
            for j in range(1, self.nranks):
                CONNS[j].send(vals)
        else:
            CONNS[0].send(vals)
            vals = CONNS[0].recv()
        return tuple(float(x) for x in vals)


runtime.Device = SocketSlabDevice
sys.argv = ["bench.py"] + sys.argv[1:]
import bench  # noqa: E402

bench.main()
