"""The RCCL leg (csrc/fs_comm.hip) on a single GPU: a 1-rank communicator whose rank is its own slab neighbour.
Checks what a 1-GPU box can check of the real ncclSend/ncclRecv path - symbol loading, ncclCommInitRank, grouped send/recv on
the context's stream, row offsets / counts / dtype of the ghost-row blocks, several fields per group - and the ncclAllReduce."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,depth,halo", [(np.float32, 8, 8), (np.float32, 3, 8), (np.float64, 2, 4)])
def test_self_exchange_fills_ghost_rows(hip_lib, dtype, depth, halo):
    from fs import _lib
    nx, ny = 200, 24
    ctx = ctypes.c_void_p()
    _lib.call("fs_create", ctypes.byref(ctx), 0, nx, ny, 0 if dtype == np.float32 else 1, 0, ny, halo)
    saved = os.dup(1)
    try:
        uid = ctypes.create_string_buffer(128)
        _lib.call("fs_comm_unique_id", uid)
        os.dup2(2, 1)                       # RCCL prints its banner on C stdout
        try:
            _lib.call("fs_comm_init", ctx, 0, 1, ctypes.c_char_p(uid.raw))
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved, 1)
        rng = np.random.default_rng(3)
        rows = ny + 2 * halo
        fields, before = [], []
        for nchan in (1, 2, 3):
            h = ctypes.c_void_p()
            _lib.call("fs_field_alloc", ctx, nchan, ctypes.byref(h))
            a = rng.uniform(-1, 1, (nx, rows, nchan)).astype(dtype)
            _lib.call("fs_field_upload", h, a.ctypes.data_as(ctypes.c_void_p), 0, rows)
            fields.append((h, nchan))
            before.append(a)
        arr = (ctypes.c_void_p * len(fields))(*[h for h, _ in fields])
        _lib.call("fs_halo_exchange_self", ctx, arr, len(fields), depth)
        _lib.call("fs_sync", ctx)
        H = halo
        for (h, nchan), a in zip(fields, before):
            got = np.empty_like(a)
            _lib.call("fs_field_download", h, got.ctypes.data_as(ctypes.c_void_p), 0, rows)
            exp = a.copy()
            exp[:, H - depth:H] = a[:, H:H + depth]                    # lower ghost rows <- first owned rows
            exp[:, H + ny:H + ny + depth] = a[:, H + ny - depth:H + ny]  # upper ghost rows <- last owned rows
            assert np.array_equal(got, exp), nchan
        vals = (ctypes.c_double * 3)(1.5, -2.0, 1e300)
        _lib.call("fs_allreduce_sum", ctx, vals, 3)
        assert list(vals) == [1.5, -2.0, 1e300]
        with pytest.raises(_lib.FsError):
            _lib.call("fs_halo_exchange_self", ctx, arr, len(fields), halo + 1)
    finally:
        os.close(saved)
        _lib.load().fs_comm_destroy(ctx)
        for h, _ in fields:
            _lib.load().fs_field_free(h)
        _lib.load().fs_destroy(ctx)
