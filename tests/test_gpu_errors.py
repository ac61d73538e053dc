"""Error behaviour of the C-ABI through the Python shell: every misuse is a negative status + message (FsError), never a
crash or a silent no-op.  (The reference raises plain Python exceptions only for unknown scheme / scene.)"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev(hip_lib):
    import fs
    from fs.boundary_condition import BoundaryCondition
    fs.runtime.init(gpu=0, dtype="f32")
    mask = np.zeros((32, 16), np.uint8); mask[:, :2] = 1; mask[:, -2:] = 1
    bc = BoundaryCondition(np.zeros((32, 16, 2), np.float32), mask)
    yield bc.device
    bc.device.close()


def test_wrong_channel_count_and_aliasing(dev):
    from fs._lib import FsError
    v, p, d = dev.alloc(2), dev.alloc(1), dev.alloc(3)
    with pytest.raises(FsError, match="channel count"):
        dev.jacobi_sweep(0.01, 0.1, v, p, v)                # pn must be 1-channel
    with pytest.raises(FsError, match="alias"):
        dev.cip_nonadv(0.01, 0.1, 100.0, v, v, p)           # fn == fc
    with pytest.raises(FsError, match="distinct"):
        dev.jacobi_sweep(0.01, 0.1, p, p, v)
    with pytest.raises(FsError):
        dev.mac_update(7, 0.01, 0.1, 100.0, dev.alloc(2), v, p)   # unknown scheme code
    with pytest.raises(FsError, match="bc_dye"):
        dev.dye_bc(d)                                        # no dye scene uploaded


def test_row_range_and_foreign_field(dev, hip_lib):
    import fs
    from fs import _lib
    from fs.boundary_condition import BoundaryCondition
    v = dev.alloc(2)
    assert hip_lib.fs_limit_field(dev._ctx, 10.0, v._h, 0, 17) < 0           # 17 > rows
    assert b"row range" in hip_lib.fs_last_error()
    other = BoundaryCondition(np.zeros((32, 16, 2), np.float32), np.ones((32, 16), np.uint8)).device
    try:
        assert hip_lib.fs_limit_field(dev._ctx, 10.0, other.alloc(2)._h, 0, 16) < 0
        assert b"another context" in hip_lib.fs_last_error()
    finally:
        other.close()
    with pytest.raises(ValueError, match="expected array of shape"):
        v.from_numpy(np.zeros((16, 32, 2), np.float32))


def test_kernel_before_mask_upload(hip_lib):
    from fs import _lib
    ctx = ctypes.c_void_p()
    _lib.call("fs_create", ctypes.byref(ctx), 0, 32, 16, 0, 0, 16, 0)
    f = ctypes.c_void_p()
    _lib.call("fs_field_alloc", ctx, 2, ctypes.byref(f))
    assert hip_lib.fs_limit_field(ctx, 10.0, f, 0, 16) == -3                 # FS_ERR_STATE
    assert b"mask not uploaded" in hip_lib.fs_last_error()
    assert hip_lib.fs_create(ctypes.byref(ctypes.c_void_p()), 0, 2, 2, 0, 0, 2, 0) == -1     # grid too small
    assert hip_lib.fs_create(ctypes.byref(ctypes.c_void_p()), 0, 32, 16, 5, 0, 16, 0) == -1    # bad dtype
    hip_lib.fs_destroy(ctx)


def test_exchange_misuse(dev, hip_lib):
    """Ghost-row exchange entry points without a communicator, and the begin / wait protocol on a 1-rank communicator."""
    import os
    from fs import _lib
    from fs._lib import FsError
    v = dev.alloc(2)
    arr = (ctypes.c_void_p * 1)(v._h)
    for name, args in (("fs_halo_exchange_multi", (arr, 1, 0)), ("fs_halo_exchange_begin", (arr, 1, 0)),
                       ("fs_halo_exchange_self", (arr, 1, 0))):
        with pytest.raises(FsError, match="fs_comm_init"):
            _lib.call(name, dev._ctx, *args)
    with pytest.raises(FsError, match="fs_comm_init"):
        _lib.call("fs_halo_exchange_mark", dev._ctx)
    _lib.call("fs_halo_exchange_wait", dev._ctx)                      # nothing in flight: a no-op, not an error
    with pytest.raises(FsError, match="communicator"):
        _lib.call("fs_comm_loopback", dev._ctx, 1)
    # 1-rank communicator: protocol errors
    uid = ctypes.create_string_buffer(128)
    _lib.call("fs_comm_unique_id", uid)
    saved = os.dup(1); os.dup2(2, 1)
    try:
        _lib.call("fs_comm_init", dev._ctx, 0, 1, ctypes.c_char_p(uid.raw))
    finally:
        ctypes.CDLL(None).fflush(None); os.dup2(saved, 1); os.close(saved)
    with pytest.raises(FsError, match="already initialised"):
        _lib.call("fs_comm_init", dev._ctx, 0, 1, ctypes.c_char_p(uid.raw))
    with pytest.raises(FsError, match="depth"):
        _lib.call("fs_halo_exchange_begin", dev._ctx, arr, 1, 1)      # this context has no ghost rows (halo 0)
    _lib.call("fs_halo_exchange_begin", dev._ctx, arr, 1, 0)
    with pytest.raises(FsError, match="in flight"):
        _lib.call("fs_halo_exchange_begin", dev._ctx, arr, 1, 0)
    with pytest.raises(FsError, match="in flight"):
        _lib.call("fs_halo_exchange_mark", dev._ctx)
    _lib.call("fs_halo_exchange_wait", dev._ctx)
    neg = (ctypes.c_int * 1)(-1)
    with pytest.raises(FsError, match="negative"):
        _lib.call("fs_halo_exchange_begin_partial", dev._ctx, arr, neg, 1, 0)
    _lib.load().fs_comm_destroy(dev._ctx)


def test_box_probes_give_plausible_rates_and_refuse_nonsense(dev):
    """The three probes bench.py runs before its timed region (include/fs_hip.h fs_box_rates, fs_box_valu_rate, fs_box_mixed_rate): on an
    MI355X a float4 stream moves TB/s, a SIMD issues tenths of a G wave-instruction per second; a zero budget / a buffer below 1 MiB is an error."""
    from fs._lib import FsError
    rd, cp = dev.box_rates(64 << 20, 5.0)
    assert 500.0 < rd < 20000.0 and 500.0 < cp < 20000.0
    assert 0.05 < dev.box_valu_rate(5.0) < 5.0
    assert 500.0 < dev.box_mixed_rate(64 << 20, 5.0) < 20000.0
    with pytest.raises(FsError):
        dev.box_rates(1024, 5.0)
    with pytest.raises(FsError):
        dev.box_valu_rate(0.0)
    with pytest.raises(FsError):
        dev.box_mixed_rate(64 << 20, -1.0)
