#!/usr/bin/env python3
"""Latency of one grouped RCCL send/recv launch on this box, measured in loop-back (fs_halo_exchange_self on a 1-rank
communicator: the copies stay on the GPU, so this is the fixed cost of the grouped call - host API, RCCL kernel launch,
channel setup - not xGMI transfer time)."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("2d-fluid-simulator_amd")
from fs import _lib  # noqa: E402

nx, ny, halo = 8192, 512, 16
print("FS_PACK_HALO =", os.environ.get("FS_PACK_HALO", "1 (default)"))
ctx = ctypes.c_void_p()
_lib.call("fs_create", ctypes.byref(ctx), 0, nx, ny, 0, 0, ny, halo)
uid = ctypes.create_string_buffer(128)
_lib.call("fs_comm_unique_id", uid)
saved = os.dup(1); os.dup2(2, 1)
_lib.call("fs_comm_init", ctx, 0, 1, ctypes.c_char_p(uid.raw))
ctypes.CDLL(None).fflush(None); os.dup2(saved, 1)
hs = []
for nchan in (2, 2, 2, 2, 2, 2, 1):
    h = ctypes.c_void_p(); _lib.call("fs_field_alloc", ctx, nchan, ctypes.byref(h)); hs.append((h, nchan))
for label, sel, depth in (("1 field  x 1 ch  x 1 row ", hs[6:7], 1), ("1 field  x 1 ch  x 8 rows", hs[6:7], 8),
                          ("3 fields x 4 ch  x 8 rows", hs[4:7], 8), ("6 fields x 12 ch x 8 rows", hs[0:6], 8),
                          ("6 fields x 12 ch x 16 rows", hs[0:6], 16), ("6 fields x 12 ch x 2 rows", hs[0:6], 2)):
    arr = (ctypes.c_void_p * len(sel))(*[h for h, _ in sel])
    for _ in range(20):
        _lib.call("fs_halo_exchange_self", ctx, arr, len(sel), depth)
    _lib.call("fs_sync", ctx)
    t0 = time.perf_counter(); n = 300
    for _ in range(n):
        _lib.call("fs_halo_exchange_self", ctx, arr, len(sel), depth)
    t_host = time.perf_counter() - t0
    _lib.call("fs_sync", ctx)
    t_all = time.perf_counter() - t0
    kb = depth * nx * 4 * sum(c for _, c in sel) / 1024
    print(f"{label}: {kb:7.0f} KB per direction   host issue {t_host / n * 1e6:6.1f} us   stream-serialised {t_all / n * 1e6:6.1f} us per exchange", flush=True)
_lib.load().fs_comm_destroy(ctx)
