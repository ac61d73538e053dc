"""Device runtime: what `ti.init()` + Taichi's field allocator are to the reference (main.py:65-69).

A `Device` owns one libfs_hip context = one y-slab of the grid on one MI355X.  `Field` is the
subset of the Taichi field API the reference uses (`shape`, `to_numpy`, `from_numpy`, `fill`;
fs/fluid_simulator.py:36, fs/boundary_condition.py:81-83, fs/double_buffer.py:17-18).

Multi-GPU (new - the reference is single-device): one process per GPU, the grid is cut along y into
`nranks` slabs with `halo` ghost rows on each side.  Ghost-row validity is tracked per field: every
kernel wrapper declares the stencil radius it reads each input with and which fields it writes; a
halo exchange (RCCL send/recv inside libfs_hip) is issued only when an input's ghost rows are stale.
With nranks == 1 all of this is a no-op and the launches are exactly the reference's sequence.

`DeviceBase` holds that host logic and is backend-agnostic (the CPU test-suite drives it with a
numpy/gloo stand-in to check the slab logic without a GPU); `Device` binds it to libfs_hip.so.
"""
import ctypes
import itertools
import os
import time
import weakref

import numpy as np

from . import _lib

_DTYPES = {"f32": np.float32, "f64": np.float64, np.float32: np.float32, np.float64: np.float64,
           np.dtype("float32"): np.float32, np.dtype("float64"): np.float64}

# process-wide configuration, set by init() (the analogue of ti.init)
_config = {
    "gpu": 0, "dtype": np.float32, "rank": 0, "nranks": 1, "halo": None,
    "bcast": None,        # bcast(bytes | None) -> bytes : rank 0's payload on every rank (rendezvous only)
    "allgather": None,    # allgather(obj) -> list[obj]  : host-side gather for Field.to_numpy() on slabs
    "device_cls": None,   # test hook: a DeviceBase subclass
}
_devices = []
_serials = itertools.count(1)


def init(gpu=0, dtype="f32", rank=0, nranks=1, halo=None, bcast=None, allgather=None, device_cls=None):
    """Select GPU / precision / slab decomposition for the contexts created afterwards."""
    if dtype not in _DTYPES:
        raise ValueError(f"dtype must be 'f32' or 'f64', got {dtype!r}")
    if nranks > 1 and bcast is None and device_cls is None:
        raise ValueError("nranks > 1 needs a `bcast` callable to share the RCCL unique id")
    _config.update(gpu=gpu, dtype=_DTYPES[dtype], rank=rank, nranks=nranks, halo=halo, bcast=bcast,
                   allgather=allgather, device_cls=device_cls)


def config():
    return dict(_config)


def slab_rows(ny, rank, nranks):
    """Rows [y0, y0 + n) owned by `rank`: as even as possible, lower ranks take the remainder."""
    base, rem = divmod(ny, nranks)
    n = base + (1 if rank < rem else 0)
    y0 = rank * base + min(rank, rem)
    return y0, n


def create_device(resolution):
    cls = _config["device_cls"] or Device
    dev = cls(resolution[0], resolution[1], _config["dtype"], gpu=_config["gpu"], rank=_config["rank"],
              nranks=_config["nranks"], halo=_config["halo"], bcast=_config["bcast"],
              allgather=_config["allgather"])
    _devices[:] = [r for r in _devices if r() is not None]
    _devices.append(weakref.ref(dev))
    return dev


def current_device(resolution=None):
    """Most recently created device (optionally: with this (X, Y) resolution)."""
    for ref in reversed(_devices):
        dev = ref()
        if dev is not None and (resolution is None or tuple(resolution) == (dev.nx, dev.ny)):
            return dev
    raise RuntimeError(
        "no device context for this resolution yet: construct the BoundaryCondition first "
        "(it creates the context, like the reference's boundary condition is the first object built)")


class Field:
    """Device-resident field of shape (X, Y) with `nchan` channels."""

    def __init__(self, dev, nchan):
        self.dev = dev
        self.nchan = nchan
        self._h = dev._p_alloc(nchan)
        self.serial = next(_serials)   # unique for life (id() is recycled after garbage collection): identity in signatures and op keys
        dev._handle_serial[id(self._h)] = self.serial
        self.valid = dev.halo          # ghost rows valid to this depth (zero-filled == consistent everywhere)
        self.user_data = False         # set once the user uploads / fills data (disables fusions that rely on invariants)
        self.pending_limit = None      # limit_field(self, limit) issued by the solver but not launched yet: it rides with the next step's velocity
                                       # boundary kernel (DeviceBase.limit_field); anything else that touches the field launches it first
        self.pending_clamp = None      # clamp_inflow(self, low, high) issued by the dye solver but not launched: the next dye boundary kernel rewrites
                                       # exactly those cells (DeviceBase.clamp_inflow); anything else that touches the field launches it first
        self.bc_parity = 0             # parity of the next merged limit + boundary launch on this buffer (DeviceBase.velocity_bc)
        self.static_id = 0             # content class of the cells NO kernel ever writes (deep wall cells): 0 = still the zeros of the
                                       # allocation, a fresh token after every upload / fill.  Passes that store only the cells that can
                                       # change (fs_rbsor_pair, the fused gradient + advection pass) need source and target buffer in
                                       # the same class, and run one carrying pass when they are not (which then copies the class).

    @property
    def shape(self):
        return (self.dev.nx, self.dev.ny)

    def _full_shape(self, nrows):
        return (self.dev.nx, nrows) if self.nchan == 1 else (self.dev.nx, nrows, self.nchan)

    def fill(self, value):
        self.pending_limit = self.pending_clamp = None      # (every cell is overwritten: what the deferred passes would have done is gone either way)
        self.dev._p_fill(self._h, float(value))
        self.valid = self.dev.halo
        self.static_id = next(_serials)
        self.user_data = self.user_data or float(value) != 0.0

    def from_numpy(self, arr):
        """Upload a GLOBAL (X, Y[, C]) array; each slab keeps its rows (ghost rows included)."""
        dev = self.dev
        arr = np.asarray(arr)
        if arr.shape != self._full_shape(dev.ny):
            raise ValueError(f"expected array of shape {self._full_shape(dev.ny)}, got {arr.shape}")
        win = np.ascontiguousarray(arr[:, dev.g_lo:dev.g_hi], dtype=dev.dtype)
        self.pending_limit = self.pending_clamp = None      # (as fill)
        dev._p_upload(self._h, self.nchan, win, dev.g_lo - (dev.y0 - dev.halo), dev.g_hi - dev.g_lo)
        self.valid = dev.halo
        self.user_data = True
        self.static_id = next(_serials)

    def to_numpy(self, local=False):
        """Global (X, Y[, C]) array (gathered over slabs) or, with local=True, this slab's owned rows."""
        dev = self.dev
        dev.flush_limit(self)
        mine = dev._p_download(self._h, self.nchan, dev.halo, dev.nyl).reshape(self._full_shape(dev.nyl))
        if local or dev.nranks == 1:
            return mine
        if dev.allgather is None:
            raise RuntimeError("to_numpy() on a slab needs runtime.init(allgather=...) or local=True")
        return np.concatenate(dev.allgather(mine), axis=1)

    def local_window(self):
        """All local rows that lie inside the global domain (ghost rows included) - diagnostics/tests."""
        dev = self.dev
        dev.flush_limit(self)
        r0, n = dev.g_lo - (dev.y0 - dev.halo), dev.g_hi - dev.g_lo
        return dev._p_download(self._h, self.nchan, r0, n).reshape(self._full_shape(n))

    def __del__(self):
        try:
            self.dev._handle_serial.pop(id(self._h), None)
            self.dev._p_free(self._h)
        except Exception:
            pass


class DeviceBase:
    """Slab geometry + ghost-row bookkeeping + one method per reference kernel (backend-agnostic)."""

    MIN_HALO = 2       # deepest stencil on the path: Kawamura-Kuwahara / velocity-BC mirror (fs/advection.py:39-55)
    DEFAULT_HALO = 20  # slabs: ghost rows per side (round 4: 20 for slabs of 160 rows and more, 16 from 128 rows: at 20 the exchange pattern of
                       # the default solver is one grouped exchange every 2 steps WITH the two-iteration red-black pass, whose reach of 4 rows
                       # breaks the period of the pattern at 16 - tools/slab_period.py).  Deeper = fewer, larger exchanges (6 / 3 / 2 / 1 grouped send/recv
                       # launches per CIP+VC step at depth 2 / 4 / 8 / 16) for (depth/rows) redundant compute.  One grouped
                       # exchange costs ~45 us of GPU-side latency (pack, RCCL kernel, unpack, stream hand-offs) whatever
                       # its size, a 512-row slab step 130 us: the fewest exchanges win (tools/overlap_bench.py).
                       # Slabs thinner than 128 rows default to 8.

    def __init__(self, nx, ny, dtype, gpu=0, rank=0, nranks=1, halo=None, bcast=None, allgather=None):
        self.nx, self.ny = int(nx), int(ny)
        self.dtype = np.dtype(_DTYPES[dtype])
        self.gpu, self.rank, self.nranks = gpu, rank, nranks
        self.bcast, self.allgather = bcast, allgather
        self.y0, self.nyl = slab_rows(self.ny, rank, nranks)
        if halo is None:
            # the default must be the SAME number on every rank (neighbours exchange `halo` rows with each other and run the
            # same validity bookkeeping): derive it from the thinnest slab of the decomposition, not from this rank's own height
            thinnest = self.ny // nranks
            halo = 0 if nranks == 1 else int(os.environ.get("FS_HALO", self.DEFAULT_HALO if thinnest >= 160 else (16 if thinnest >= 128 else min(8, thinnest))))
        self.halo = int(halo)
        if nranks > 1 and self.halo < self.MIN_HALO:
            raise ValueError(f"slab decomposition needs halo >= {self.MIN_HALO}")
        if nranks > 1 and self.ny // nranks < self.halo:      # rank-independent test: every rank raises, or none does
            raise ValueError("slab thinner than its halo: use fewer ranks or a larger grid")
        self.rows = self.nyl + 2 * self.halo
        # global rows covered by local rows (ghost rows included), clipped to the domain
        self.g_lo = max(0, self.y0 - self.halo)
        self.g_hi = min(self.ny, self.y0 + self.nyl + self.halo)
        self.bc_radius_v, self.bc_radius_p = 2, 1
        self.lazy_bc_ok = False
        self.rb_pair_ok = False       # the mask admits the two-iteration red-black pass (rbsor_pair)
        self.n_exchanges = 0          # grouped send/recv launches issued
        self.n_exchanged_fields = 0   # fields refreshed by them
        self.n_exchanged_bytes = 0    # payload sent to ONE neighbour by them (an interior rank sends twice that)
        self.n_overlapped = 0         # exchanges that ran behind the interior rows of the kernel that needed them
        self._oplog = None            # list of primitive operations while a period is being logged (see tape_period)
        self._cur_writes = ()
        self._fields = weakref.WeakSet()
        self._handle_serial = {}      # id(handle) -> serial of the Field that owns it (dropped with the Field)
        # exchange behind the interior rows of the kernel that needs it (communication stream + events + two extra strip
        # launches): measured neutral in loop-back (181 vs 187 us per slab step), while running the exchange in line on the
        # compute stream saves the stream hand-offs -> off by default; FS_OVERLAP=1 turns it (and the communication stream) on
        self.overlap = nranks > 1 and os.environ.get("FS_OVERLAP", "0") == "1"
        self.overlap_stream = self.overlap      # the exchanges run on a communication stream of their own (tape_period switches `overlap`
                                                # off while it logs - this remembers what the context was created with)
        self.partial = True           # send only the ghost rows beyond a field's validity (tests flip the attribute; its switch went in round 6)
        # "refresh everything": when an exchange is due anyway, every ghost-read field below full depth travels with it (see _run).
        # Loop-back, middle slab of the 8-way cut of bc5 res 4096, halo 16: 1.0 -> 0.67 grouped exchanges per step, 141.8 -> 133.0 us per
        # step (compute alone 116); the price is +47 % bytes per step and neighbour (4.05 instead of 2.77 MB) - exchange_all = False for a
        # link-bound node.
        self.exchange_all = True
        self.clamp_deferral = False
        self.limit_deferral = False   # set by upload_scene: limit_field may ride with the next step's velocity boundary kernel (limit_field below)

    # ---- ghost-row bookkeeping --------------------------------------------------------------------
    def exchange(self, field, depth=None):
        self.exchange_many([field], depth)

    def _handles(self, fields, depth):
        """(handle, channels, rows already valid) per field: only the ghost rows beyond a field's validity have to travel."""
        return [(f._h, f.nchan, min(max(f.valid, 0), depth) if self.partial else 0) for f in fields]

    def _account(self, fields, handles, depth):
        for f in fields:
            f.valid = depth
        self.n_exchanges += 1
        self.n_exchanged_fields += len(fields)
        self.n_exchanged_bytes += self.nx * self.dtype.itemsize * sum((depth - v) * c for _, c, v in handles)

    def exchange_many(self, fields, depth=None):
        """Refresh the ghost rows of several fields with ONE grouped send/recv (one launch, one latency)."""
        depth = self.halo if depth is None else depth
        handles = self._handles(fields, depth)
        if self._oplog is not None:       # a logged period keeps the two halves apart (the tape moves the begin up)
            self._oplog += [("begin", handles, depth), ("wait",)]
        self._p_exchange_many(handles, depth)
        self._account(fields, handles, depth)

    def _p_exchange_many(self, handles, depth):     # backends without a grouped primitive: one by one, full depth
        for h, nchan, _ in handles:
            self._p_exchange(h, nchan, depth)

    def exchange_begin(self, fields, depth=None):
        """Start refreshing the ghost rows of `fields`; until exchange_wait() only kernels that neither read ghost rows nor
        write the `depth` outermost owned rows of these fields may be launched (see _run)."""
        depth = self.halo if depth is None else depth
        handles = self._handles(fields, depth)
        self._p_exchange_begin(handles, depth)
        self._account(fields, handles, depth)
        self.n_overlapped += 1

    def exchange_wait(self):
        self._p_exchange_wait()

    def set_overlap(self, on):
        """Exchanges behind the interior rows of the kernel that needs them, on the communication stream (True), or in line on the compute
        stream (False).  Same bits either way; every rank must choose the same (bench.py decides from max-over-ranks timings).  Tapes recorded
        before the switch belong to the old setting."""
        self.overlap = self.overlap_stream = bool(on) and self.nranks > 1
        self._p_set_overlap(self.overlap_stream)

    def _p_set_overlap(self, on):                   # backends without a communication stream: the flags above are all there is
        pass

    def _p_exchange_begin(self, handles, depth):    # backends without an asynchronous primitive: blocking
        self._p_exchange_many(handles, depth)

    def _p_exchange_wait(self):
        pass

    def _after_kernel(self, name, written, lo, hi):   # test hook: `written` fields were computed on local rows [lo, hi)
        pass

    def _p_exchange_mark(self):
        pass

    def _p_lazy_bc_ok(self):        # backends without the lazy pressure boundary condition
        return False

    def _p_rb_pair_ok(self):        # backends without the two-iteration red-black pass
        return False

    def _p_max_over_ranks(self, values):
        """Element-wise maximum of a short list of non-negative numbers over all ranks (collective)."""
        if self.nranks == 1:
            return list(values)
        raise NotImplementedError("this backend needs _p_max_over_ranks for slab runs")

    _BIG = 1 << 40      # far above any count that is agreed on (tape lengths, periods); exact in the f64 the collectives carry

    def _p_min_max_over_ranks(self, value):
        """(min, max) of one non-negative integer below 2^40 over all ranks (collective; built on _p_max_over_ranks)."""
        if self.nranks == 1:
            return value, value
        if not 0 <= value < self._BIG:
            raise ValueError("_p_min_max_over_ranks: value out of range")
        hi, neg_lo = self._p_max_over_ranks([value, self._BIG - value])
        return self._BIG - int(neg_lo), int(hi)

    def _p_same_over_ranks(self, values):
        """True iff every rank passed the same list of non-negative integers (ONE collective: [v, BIG - v] per element - element-wise
        maxima agree with the local values exactly when minimum and maximum coincide)."""
        if self.nranks == 1:
            return True
        if any(not 0 <= v < self._BIG for v in values):
            raise ValueError("_p_same_over_ranks: value out of range")
        got = self._p_max_over_ranks([x for v in values for x in (v, self._BIG - v)])
        return all(int(got[2 * k]) == v and int(got[2 * k + 1]) == self._BIG - v for k, v in enumerate(values))

    def _run(self, name, args, reads=(), writes=(), pointwise=False, full_writes=(), split=True):
        """Launch one kernel on this slab.

        Single rank: rows [0, Y), nothing else happens.  Slabs: every field carries `valid` = how many ghost rows are
        currently correct.  A kernel reading field f with stencil radius r can be evaluated redundantly on
        e = min(f.valid - r) ghost rows as well (communication-avoiding: with a deep halo several kernels run between
        two exchanges); the cells a masked kernel does NOT write keep their old content, so the outputs' previous
        validity caps e too.  When an input lacks its radius, every field of this kernel that is below full depth is
        refreshed in ONE grouped send/recv (only the rows beyond each field's validity travel).  For `split` kernels (all
        but the op-list boundary kernels) that exchange runs behind the kernel's own interior rows: mark -> rows [2H, nyl)
        -> begin -> wait -> the two edge strips.
        """
        if name not in ("velocity_bc_limit", "dye_bc_limit"):      # a deferred limit_field / clamp_inflow runs before anything else looks at (or writes into) its field
            for f in [f for f, _ in reads] + list(writes) + list(full_writes):
                if f.pending_limit is not None or f.pending_clamp is not None:
                    self.flush_limit(f)
        multi = self.nranks > 1
        off = self.y0 - self.halo
        # (identities, not the fields: a reference kept here would postpone the release of a temporary field to the NEXT launch - which may
        #  sit inside a hipGraph capture, where hipFree is illegal)
        self._cur_writes = tuple(id(f._h) for f in list(writes) + list(full_writes))
        if pointwise or not multi:      # pointwise: all in-domain local rows, ghost rows included -> validity preserved
            lo, hi = (self.g_lo - off, self.g_hi - off) if pointwise else (self.halo, self.halo + self.nyl)
            self._kernel(name, args, lo, hi)
            return
        H = self.halo
        for f, radius in reads:
            if radius > H:
                raise RuntimeError(f"{name}: stencil radius {radius} exceeds halo {H}")

        def unique(fields):
            out = []
            for f in fields:
                if not any(f is g for g in out):
                    out.append(f)
            return out

        for f, r in reads:
            if r > 0:
                f.ghost_read = True       # some kernel reads this field's ghost rows: a candidate of the "refresh everything" policy
        e_reads = min([f.valid - r for f, r in reads], default=H)
        need = []
        if e_reads < 0:
            need = unique([f for f, _ in reads if f.valid < H] + [f for f in writes if f.valid < H])
            if self.exchange_all:
                # One grouped exchange costs ~26 us of latency whatever it carries: when one is due anyway, refresh EVERY field of the
                # step's working set that is below full depth - the next exchange is then as far away as the halo allows (halo 16, CIP
                # + VC + red-black SOR: one exchange every two steps instead of one per step, at the same number of rows moved)
                # (in creation order - the same on every rank; the WeakSet's own order is by address and differs between processes,
                # and both sides of a slab boundary must pack the same fields in the same order)
                need = unique(need + [f for f in sorted(self._fields, key=lambda f: f.serial)
                                      if getattr(f, "ghost_read", False) and f.valid < H and not any(f is w for w in full_writes)])
        elif e_reads >= 2:
            need = unique([f for f in writes if f.valid < e_reads])     # lift the outputs so that the extension is not wasted
        pending = False
        if need:
            if self.overlap and split and self.nyl >= 2 * H + 8:
                # The exchange (depth H) runs on the communication stream.  Rows [2H, nyl) of this slab depend on owned rows only
                # and their outputs are not among the rows being sent: launch them FIRST (the GPU is busy while the host issues
                # the exchange), the two edge strips after the exchange has landed.
                self._p_exchange_mark()
                self._kernel(name, args, 2 * H, self.nyl)
                self.exchange_begin(need)
                pending = True
            else:
                self.exchange_many(need)
            e_reads = min([f.valid - r for f, r in reads], default=H)
        e = max(0, min([e_reads] + [f.valid for f in writes]))
        lo = max(H - e, self.g_lo - off)
        hi = min(H + self.nyl + e, self.g_hi - off)
        if pending:
            self.exchange_wait()
            self._kernel(name, args, lo, 2 * H)
            self._kernel(name, args, self.nyl, hi)
        else:
            self._kernel(name, args, lo, hi)
        for f in list(writes) + list(full_writes):      # full_writes: every cell of the computed rows is overwritten
            f.valid = e
        self._after_kernel(name, list(writes) + list(full_writes), lo, hi)

    def alloc(self, nchan):
        f = Field(self, nchan)
        self._fields.add(f)
        return f

    # ---- command tapes: a logged period of primitive operations, replayed without the bookkeeping above --------------------
    def _kernel(self, name, args, lo, hi):
        if self._oplog is not None:
            self._oplog.append(("k", name, tuple(args) + (lo, hi), self._cur_writes))
        self._p_kernel(name, *args, lo, hi)

    def _state_signature(self):
        return tuple(sorted((f.serial, f.valid, f.user_data, f.static_id) for f in self._fields))

    def _op_key(self, op):
        """Hashable identity of a logged operation.  Handles are named by the serial number of the Field that owns them (a Field keeps
        its handle for life; id() alone would be recycled after a garbage collection and let a stale tape match a new Field)."""
        name = lambda h: self._handle_serial.get(id(h), ("anon", id(h)))
        if op[0] == "k":
            return ("k", op[1], tuple(a if isinstance(a, (int, float, type(None))) else name(a) for a in op[2]))
        if op[0] == "begin":
            return ("begin", tuple((name(h), c, v) for h, c, v in op[1]), op[2])
        return (op[0],)

    def tape_period(self, step_fn, nsteps=2, tries=14, hoist=True, max_blocks=6):
        """Log blocks of `nsteps` x step_fn() (executed normally) until the last P blocks (P = 1 .. max_blocks) repeat the P blocks
        before them - the same primitive operations from the same bookkeeping state - then compile those P blocks into a tape of
        P * nsteps steps (see replay_tape).  Returns None if nothing repeated within `tries` blocks (the caller keeps stepping
        eagerly).  nsteps = 2 returns every DoubleBuffer to its parity; the ghost-row bookkeeping settles into a period of 1, 2 or
        4 steps after a transient of up to ~8 steps, depending on halo depth and solver.  Collective on slab runs: every rank
        takes the same decisions because every rank runs the same validity tracker."""
        saved_overlap, self.overlap = self.overlap, False       # blocking begin + wait pairs: the tape moves the begins itself
        try:
            blocks = []          # (state signature before the block, op keys, op log, exchange counters spent)
            for _ in range(tries):
                sig0 = self._state_signature()
                self._oplog = []
                c0 = (self.n_exchanges, self.n_exchanged_fields, self.n_exchanged_bytes)
                for _ in range(nsteps):
                    step_fn()
                log, self._oplog = self._oplog, None
                c1 = (self.n_exchanges, self.n_exchanged_fields, self.n_exchanged_bytes)
                blocks.append((sig0, [self._op_key(op) for op in log], log, tuple(b - a for a, b in zip(c0, c1))))
                found = 0
                for P in range(1, max_blocks + 1):
                    if len(blocks) < 2 * P:
                        break
                    a, b = blocks[-2 * P:-P], blocks[-P:]
                    if (all(x[0] == y[0] and x[1] == y[1] for x, y in zip(a, b)) and self._state_signature() == b[0][0]):
                        found = P
                        break
                # every rank runs the same tracker and should take the same decision - but a rank that found the period one block later
                # (or not at all) would issue a different number of exchanges from here on, and the unpaired send / recv would hang:
                # agree on it (one small collective per block on slab runs), and carry on logging unless ALL ranks found the same period
                lo, hi = self._p_min_max_over_ranks(found)
                if lo == hi and found:
                    b = blocks[-found:]
                    log = [op for blk in b for op in blk[2]]
                    per_period = tuple(sum(blk[3][k] for blk in b) for k in range(3))
                    tape = self._compile_tape(log, found * nsteps, per_period, hoist)
                    # the shape of the compiled tape must be the same on every rank (the two lengths travel as separate
                    # elements - packed into one number, tapes of 1000 operations and more would alias)
                    if not self._p_same_over_ranks([len(tape["prologue"]), len(tape["ops"])]):
                        self.free_tape(tape)
                        return None
                    return tape
            return None
        finally:
            self._oplog = None
            self.overlap = saved_overlap

    def choose_exchange_mode(self, step_fn, record_tries=20, trial_steps=120, margin=0.02):
        """N > 1: record the period of step_fn() and replay it for ~trial_steps steps with the exchanges in line on the compute stream, then on
        the communication stream; keep the faster (max-over-ranks time; the communication stream must win by more than `margin`) and return
        (tape of the chosen mode or None, report).  Collective: every rank times both modes and sees the same max-over-ranks numbers, so every
        rank chooses alike.  Same bits in both modes; the caller counts the steps through step_fn, the replayed trial steps are in the report."""
        trial, tapes, last, replayed = {}, {}, None, 0
        for mode in (False, True):
            self.set_overlap(mode)
            tp = self.tape_period(step_fn, nsteps=2, tries=record_tries)
            if tp is None:
                continue
            reps = max(1, trial_steps // tp["nsteps"])
            sync = getattr(self, "sync", lambda: None)
            sync()
            self._p_max_over_ranks([0.0])               # every rank has arrived
            t0 = time.perf_counter()
            self.replay_tape(tp, reps)
            sync()
            el = float(self._p_max_over_ranks([time.perf_counter() - t0])[0])
            replayed += reps * tp["nsteps"]
            trial[mode] = el / (reps * tp["nsteps"])
            tapes[mode], last = tp, mode
        if not trial:
            self.set_overlap(False)
            return None, {"chosen": "in line (compute stream)", "replayed_steps": 0, "note": "no steady period found in either mode"}
        chosen = True in trial and (False not in trial or trial[True] < (1.0 - margin) * trial[False])
        report = {"in_line_us_per_step": round(trial[False] * 1e6, 2) if False in trial else None,
                  "overlapped_us_per_step": round(trial[True] * 1e6, 2) if True in trial else None,
                  "trial_steps_per_mode": trial_steps, "replayed_steps": replayed,
                  "chosen": "overlapped (communication stream)" if chosen else "in line (compute stream)",
                  "rule": f"max-over-ranks time of the replayed period in each mode; the communication stream is taken when it wins by more than {margin:.0%}"}
        for mode, tp in tapes.items():
            if mode != chosen or chosen != last:
                self.free_tape(tp)
        if chosen == last:
            return tapes[chosen], report
        self.set_overlap(chosen)        # the other mode was recorded last: back, and the period once more (a tape belongs to the setting it was recorded under)
        return self.tape_period(step_fn, nsteps=2, tries=record_tries), report

    @staticmethod
    def hoist_exchanges(log):
        """Order a logged period for replay -> (ops, prologue).

        A blocking exchange (begin, wait) sits right in front of the kernel that needed it.  Its `begin` only depends on the last
        kernel that WROTE one of its fields (it sends owned rows of those fields; the ghost rows it fills lie beyond the fields'
        validity, which no kernel in between reads - the tracker sized their row ranges from that validity), so it is moved up
        to just behind that kernel, or behind the previous exchange: the pack / RCCL / unpack chain then runs on the
        communication stream behind several kernels instead of in front of one.  The `wait` stays where it was.  The period is
        cyclic: a begin that moves past the start of the period is issued at its END, for the next period; it is returned in
        `prologue` (issued once before the first replay - the last replay then leaves one exchange in flight, which
        replay_tape completes).  At most one begin crosses the boundary (one exchange in flight per context)."""
        ops = list(log)
        prologue = []
        for b in [op for op in ops if op[0] == "begin"]:
            n = next(i for i, op in enumerate(ops) if op is b)
            L = len(ops)
            fields = {id(h) for h, _, _ in b[1]}
            k = 0
            for d in range(1, L - 1):
                if d > n and prologue:
                    break
                o = ops[(n - d) % L]
                if o[0] != "k" or fields & set(o[3]):
                    break
                k = d
            if k == 0:
                continue
            ops.pop(n)
            if k <= n:
                ops.insert(n - k, b)
            else:                         # crossed the start of the period: in front of the op that was at cyclic index n - k
                ops.insert(L + n - k - 1, b)
                prologue.append(b)
        return ops, prologue

    def _compile_tape(self, log, nsteps, per_period, hoist):
        # Moving a begin up only pays when the exchange runs on its own stream (FS_OVERLAP=1), and it is only safe with partial-depth
        # exchanges: a full-depth exchange rewrites ghost rows that are still valid while the kernels it was moved across read them.
        hoist = hoist and self.hoist_ok()
        ops, prologue = self.hoist_exchanges(log) if hoist else (list(log), [])
        return {"ops": ops, "prologue": prologue, "nsteps": nsteps, "per_period": per_period, "id": self._p_tape_build(ops),
                "sig": self._state_signature()}      # the bookkeeping state the period starts (and ends) in

    def hoist_ok(self):
        return bool(self.partial and self.overlap_stream)

    def _issue(self, op):
        if op[0] == "k":
            self._p_kernel(op[1], *op[2])
        elif op[0] == "begin":
            self._p_exchange_begin(op[1], op[2])
        elif op[0] == "wait":
            self._p_exchange_wait()
        elif op[0] == "mark":
            self._p_exchange_mark()

    def _p_tape_build(self, ops):       # backends without a native tape: replay issues the operations one by one
        return None

    def free_tape(self, tape):
        pass

    def _p_tape_replay(self, tape, times):
        for _ in range(times):
            for op in tape["ops"]:
                self._issue(op)

    def replay_tape(self, tape, times):
        """`times` periods (= times * tape['nsteps'] steps) of the logged operation sequence."""
        if times <= 0:
            return
        if self._state_signature() != tape["sig"]:
            raise RuntimeError("replay_tape: the ghost-row bookkeeping is not in the state the tape was recorded from (steps were "
                               "taken since that are not a whole number of periods): record a new tape")
        for op in tape["prologue"]:
            self._issue(op)
        self._p_tape_replay(tape, times)
        if tape["prologue"]:
            self._p_exchange_wait()       # the exchange the last period started for its successor: complete it (the ghost rows it
                                          # filled are valid deeper than the tracker assumes - harmless)
        e, f, b = tape["per_period"]
        self.n_exchanges += e * times
        self.n_exchanged_fields += f * times
        self.n_exchanged_bytes += b * times

    # ---- scene --------------------------------------------------------------------------------------
    def upload_scene(self, bc_mask, bc_const, bc_dye=None):
        shape = (self.nx, self.ny)
        bc_mask = np.ascontiguousarray(bc_mask, dtype=np.uint8)
        bc_const = np.ascontiguousarray(bc_const, dtype=self.dtype)
        if bc_mask.shape != shape or bc_const.shape != shape + (2,):
            raise ValueError("bc_mask must be (X, Y) and bc_const (X, Y, 2)")
        if bc_dye is not None:
            bc_dye = np.ascontiguousarray(bc_dye, dtype=self.dtype)
            if bc_dye.shape != shape + (3,):
                raise ValueError("bc_dye must be (X, Y, 3)")
        rv, rp = self._p_upload_scene(bc_mask, bc_const, bc_dye)
        # every slab analysed its own rows of the mask: the reach of chained thin walls differs from slab to slab, but the
        # ranks must run the SAME validity bookkeeping (same exchanges, same message sizes) -> one global pair of radii
        # (and one answer to "may the pressure boundary condition be evaluated lazily?")
        rv, rp, no_lazy, no_pair = self._p_max_over_ranks([rv, rp, 0 if self._p_lazy_bc_ok() else 1, 0 if self._p_rb_pair_ok() else 1])
        self.lazy_bc_ok = not no_lazy
        self.rb_pair_ok = not no_pair
        self.limit_deferral = self.nranks == 1 and os.environ.get("FS_LIMIT_DEFER", "1") == "1" and self._p_limit_deferral_ok()
        self.dye_limit_merge = self.limit_deferral and bc_dye is not None and self._p_dye_bc_limit_ok()
        self.clamp_deferral = self.limit_deferral        # (the dye's inflow clamp: stays deferred when a hot run stops deferring limit_field)
        self.bc_radius_v, self.bc_radius_p = max(2, int(rv)), max(1, int(rp))
        if self.nranks > 1 and max(self.bc_radius_v, self.bc_radius_p) > self.halo:
            raise RuntimeError(
                f"the boundary kernels reach {max(self.bc_radius_v, self.bc_radius_p)} rows on this mask (chained thin "
                f"walls); slab runs need at least that many ghost rows, have {self.halo}")

    # ---- one wrapper per reference kernel: (C-ABI name, scalars + fields, reads with radius, writes) ----
    def velocity_bc(self, v):                                   # fs/boundary_condition.py:16-39
        if v.pending_limit is not None and self.nranks == 1:
            # the limit_field the last step ended with + this step's boundary kernel in one launch (csrc/fs_march.h k_velocity_bc_limit)
            # (bc_parity: consecutive merged launches on one buffer alternate it, include/fs_hip.h - part of FluidSimulator._signature, so
            #  a captured period always holds an even number of them per buffer)
            limit, v.pending_limit = v.pending_limit, None
            self._run("velocity_bc_limit", (limit, v._h, v.bc_parity, 0, self.rows), reads=[(v, self.bc_radius_v)], writes=[v], split=False)
            v.bc_parity ^= 1
            return
        self._run("velocity_bc", (v._h,), reads=[(v, self.bc_radius_v)], writes=[v], split=False)   # op list with serial hazards

    def _p_limit_deferral_ok(self):      # backends without the merged kernel
        return False

    def _p_dye_bc_limit_ok(self):        # backends without the merged kernel
        return False

    def field_hot(self, f):              # backends without the flag: never hot
        return False

    def stop_limit_deferral(self, fields=()):
        """A run whose velocity has exceeded a speed of 9.95 (one component: 7.04) keeps the buffers' flag up: the limit pass then really runs every step, and inside a
        boundary launch it is shared by a few dozen workgroups (res 4096: 130 us against 47 as its own full-grid launch).  From here on
        limit_field is launched where the solvers call it; what the fields still owe is launched now."""
        self.limit_deferral = self.dye_limit_merge = False
        for f in fields:
            self.flush_limit(f)

    def flush_limit(self, f):
        """Launch the limit_field / clamp_inflow a field still owes (see limit_field, clamp_inflow)."""
        if f.pending_limit is not None:
            limit, f.pending_limit = f.pending_limit, None
            self._run("limit_field", (limit, f._h), pointwise=True, writes=[f])
        if f.pending_clamp is not None:
            (low, high), f.pending_clamp = f.pending_clamp, None
            self._run("clamp_inflow", (low, high, f._h), pointwise=True, writes=[f])

    def pressure_bc(self, p):                                   # fs/boundary_condition.py:41-65
        self._run("pressure_bc", (p._h,), reads=[(p, self.bc_radius_p)], writes=[p], split=False)

    def dye_bc(self, dye, velocity=None):                       # fs/boundary_condition.py:94-99
        """velocity: the field whose deferred limit_field (the flow step just ended with it, fs/solver.py:148-155, 385-392) rides in this launch."""
        if self.nranks == 1:
            dye.pending_clamp = None        # the op list rewrites exactly the cells a deferred clamp_inflow would clamp, from constants
        if velocity is not None and velocity.pending_limit is not None and self.dye_limit_merge:
            limit, velocity.pending_limit = velocity.pending_limit, None
            self._run("dye_bc_limit", (limit, velocity._h, dye._h, 0, self.rows), reads=[], writes=[dye, velocity], split=False)
            return
        self._run("dye_bc", (dye._h,), reads=[], writes=[dye], split=False)

    def mac_update(self, scheme, dt, dx, re, vn, vc, pc):       # fs/solver.py:94-107
        r = 1 if scheme == 0 else 2
        self._run("mac_update", (scheme, dt, dx, re, vn._h, vc._h, pc._h), reads=[(vc, r), (pc, 1)], writes=[vn])

    def mac_dye(self, scheme, dt, dx, dn, dc, vc):              # fs/solver.py:157-161
        r = 1 if scheme == 0 else 2
        self._run("mac_dye", (scheme, dt, dx, dn._h, dc._h, vc._h), reads=[(dc, r), (vc, 0)], writes=[dn])

    def cip_set_grad(self, dx, fx, fy, f):                      # fs/solver.py:207-211
        self._run("cip_set_grad", (dx, fx._h, fy._h, f._h), reads=[(f, 1)], writes=[fx, fy])

    def cip_nonadv(self, dt, dx, re, fn, fc, pc):               # fs/solver.py:229-240
        self._run("cip_nonadv", (dt, dx, re, fn._h, fc._h, pc._h), reads=[(fc, 1), (pc, 1)], writes=[fn])

    def cip_nonadv_dye(self, dt, dx, re, dn, dc):               # fs/solver.py:378-383
        self._run("cip_nonadv_dye", (dt, dx, re, dn._h, dc._h), reads=[(dc, 1)], writes=[dn])

    def cip_nonadv_grad(self, dx, fxn, fyn, fxc, fyc, fc, fn):  # fs/solver.py:242-261
        self._run("cip_nonadv_grad", (dx, fxn._h, fyn._h, fxc._h, fyc._h, fc._h, fn._h),
                  reads=[(fc, 1), (fn, 1), (fxc, 0), (fyc, 0)], writes=[fxn, fyn])

    def cip_advect(self, dt, dx, fn, fxn, fyn, fc, fxc, fyc, v):  # fs/solver.py:267-332
        self._run("cip_advect", (dt, dx, fn._h, fxn._h, fyn._h, fc._h, fxc._h, fyc._h, v._h),
                  reads=[(fc, 1), (fxc, 1), (fyc, 1), (v, 1)], writes=[fn, fxn, fyn])

    def cip_grad_advect(self, dt, dx, v_out, gx_out, gy_out, fn, fc, gxc, gyc, full=False):
        """K3 + K4 of the velocity field in one pass (build-side fusion): see csrc/fs_k34n.h k_cip_grad_advect_n.  v_out receives every
        cell that can differ from fc (full: every cell - the carrying pass after an upload, include/fs_hip.h)."""
        self._run("cip_grad_advect", (dt, dx, v_out._h, gx_out._h, gy_out._h, fn._h, fc._h, gxc._h, gyc._h, 1 if full else 0),
                  reads=[(fn, 2), (fc, 2), (gxc, 1), (gyc, 1)], writes=[gx_out, gy_out], full_writes=[v_out])

    def cip_step(self, dt, dx, re, v_out, gx_out, gy_out, fn, fc, pc, gxc, gyc, full=False):
        """K2 + K3 + K4 of the velocity as one call (fs/solver.py:213-227; include/fs_hip.h fs_cip_step): on large single-GPU f32 grids the
        post-K2 velocity of the all-fluid tiles stays in registers (csrc/fs_k234.h) - fn then holds it only where something reads it.
        Slabs: the same call on the row range K3 + K4 run on, where the library evaluates K2 in registers there too (cip_step_fused; the call then
        computes K2 for the rows within 2 of the range from rows within 3 of it - the radii below), else the two calls; devices without the entry
        point: the two calls."""
        if not getattr(self, "has_cip_step", False) or (self.nranks > 1 and (self.halo < 3 or not self.cip_step_fused)):
            self.cip_nonadv(dt, dx, re, fn, fc, pc)
            self.cip_grad_advect(dt, dx, v_out, gx_out, gy_out, fn, fc, gxc, gyc, full=full)
            return
        self._run("cip_step", (dt, dx, re, v_out._h, gx_out._h, gy_out._h, fn._h, fc._h, pc._h, gxc._h, gyc._h, 1 if full else 0),
                  reads=[(fc, 3), (pc, 3), (gxc, 1), (gyc, 1)], writes=[gx_out, gy_out, fn], full_writes=[v_out])

    def cip_step_dye(self, dt, dx, re, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, clamp01=False, full=False):
        """K12 + K3 + K4 of the dye as one call (fs/solver.py:385-401; include/fs_hip.h fs_cip_step_dye); slabs as cip_step, other devices: the two calls."""
        if not getattr(self, "has_cip_step", False) or (self.nranks > 1 and (self.halo < 3 or not self.cip_step_fused)):
            self.cip_nonadv_dye(dt, dx, re, fn, fc)
            self.cip_grad_advect_dye(dt, dx, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, clamp01=clamp01, full=full)
            return
        self._run("cip_step_dye", (dt, dx, re, d_out._h, gx_out._h, gy_out._h, fn._h, fc._h, gxc._h, gyc._h, v._h, 1 if clamp01 else 0, 1 if full else 0),
                  reads=[(fc, 3), (gxc, 1), (gyc, 1), (v, 1)], writes=[gx_out, gy_out, fn], full_writes=[d_out])

    def cip_grad_advect_dye(self, dt, dx, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, clamp01=False, full=False):
        """K3 + K4 of the dye in one pass (csrc/fs_k34n.h k_cip_grad_advect_n<3>); clamp01 folds clamp_field(dye, 0, 1) into the store."""
        self._run("cip_grad_advect_dye", (dt, dx, d_out._h, gx_out._h, gy_out._h, fn._h, fc._h, gxc._h, gyc._h, v._h, 1 if clamp01 else 0, 1 if full else 0),
                  reads=[(fn, 2), (fc, 2), (gxc, 1), (gyc, 1), (v, 1)], writes=[gx_out, gy_out], full_writes=[d_out])

    def vort_calc(self, dx, vort, vort_abs, vc):                # fs/vorticity_confinement.py:27-32
        self._run("vort_calc", (dx, vort._h, vort_abs._h, vc._h), reads=[(vc, 1)], writes=[vort, vort_abs])

    def vort_add(self, dt, dx, weight, vn, vc, vort, vort_abs):  # fs/vorticity_confinement.py:34-55
        self._run("vort_add", (dt, dx, weight, vn._h, vc._h, vort._h, vort_abs._h),
                  reads=[(vort_abs, 1), (vort, 0), (vc, 0)], writes=[vn])

    def vort_confine(self, dt, dx, weight, vn, vc, vort=None, vort_abs=None):
        """K5 + K6 in one pass (build-side fusion, same bits); vort / vort_abs are written only when given."""
        store = vort is not None
        self._run("vort_confine", (dt, dx, weight, vn._h, vc._h, vort._h if store else None, vort_abs._h if store else None),
                  reads=[(vc, 2)], writes=[vn] + ([vort, vort_abs] if store else []))

    def jacobi_sweep(self, dt, dx, pn, pc, vc):                 # fs/pressure_updater.py:62-66
        self._run("jacobi_sweep", (dt, dx, pn._h, pc._h, vc._h), reads=[(pc, 1), (vc, 1)], writes=[pn])

    def rbsor_halfsweep(self, dt, dx, omega, parity, pn, pc, vc):  # fs/pressure_updater.py:98-114
        self._run("rbsor_halfsweep", (dt, dx, omega, parity, pn._h, pc._h, vc._h),
                  reads=[(pc, 1), (vc, 1)], writes=[pn])

    def rbsor_iteration(self, dt, dx, omega, pn, pc, vc):
        """Odd pass (pc -> pn) + even pass (in place on pn) fused into one kernel; same bits as the two half-sweeps."""
        self._run("rbsor_iteration", (dt, dx, omega, pn._h, pc._h, vc._h), reads=[(pc, 2), (pn, 1), (vc, 2)], writes=[pn])

    def rbsor_pair(self, dt, dx, omega, pc_out, pn_out, pc, pn, vc, full=False):
        """Two red-black iterations and the pressure boundary passes between them in one pass (csrc/fs_rbpair.h): (pc_out, pn_out) <-
        what fs/pressure_updater.py:86-96 with n_iter = 2 leaves in (p.current, p.next), starting from (pc, pn).  Every cell that can
        differ between the buffers is stored (fluid cells and boundary targets; all cells with `full`), hence full_writes."""
        self._run("rbsor_pair", (dt, dx, omega, pc_out._h, pn_out._h, pc._h, pn._h, vc._h, 1 if full else 0),
                  reads=[(pc, 4), (pn, 3), (vc, 4)], full_writes=[pc_out, pn_out])

    def poisson_source(self, dt, dx, src, vc):                  # build-side: source of predict_p, once per step
        self._run("poisson_source", (dt, dx, src._h, vc._h), reads=[(vc, 1)], writes=[src])

    def jacobi_sweep_src(self, pn, pc, src):
        self._run("jacobi_sweep_src", (pn._h, pc._h, src._h), reads=[(pc, 1), (src, 0)], writes=[pn])

    def jacobi_sweep_lazy(self, pn, pc, src):
        """Source-pair sweep that evaluates the pressure boundary condition on the fly from the raw buffer (csrc/fs_march.h
        k_jacobi_lazy): the boundary value of a stencil neighbour may come from two cells away -> radius 2 (plus the reach of the
        boundary kernel on this mask, were it larger)."""
        self._run("jacobi_sweep_lazy", (pn._h, pc._h, src._h), reads=[(pc, max(2, 1 + self.bc_radius_p)), (src, 0)], writes=[pn])

    def jacobi_pair_lazy(self, pn, pc, src, swapped=False, vertical=False):
        """Two such sweeps in one pass, pn <- sweep(sweep(pc)) (csrc/fs_march.h k_jacobi_pair).  `swapped`: this is the 2nd, 4th ... pass
        of a pc -> pn -> pc sequence (include/fs_hip.h: which buffer's never-written wall cells belong to which iterate); `vertical`: the
        kernel variant whose tiles also apply the recipes that read the row below / above."""
        r = max(2, 1 + self.bc_radius_p)
        self._run("jacobi_pair_lazy", (pn._h, pc._h, src._h, (1 if swapped else 0) | (2 if vertical else 0)),
                  reads=[(pc, 2 * r), (src, r), (pn, 2 * r)], writes=[pn])

    def jacobi_quad_lazy(self, pn, pc, src):
        """Four lazily-bounded sweeps in one pass, pn[not wall] <- sweep^4(pc) (csrc/fs_jquad.h)."""
        self._run("jacobi_quad_lazy", (pn._h, pc._h, src._h), reads=[(pc, 4), (src, 3)], writes=[pn])

    def jacobi_finish(self, pc_out, pn, pc, src):
        """The last two (K7, sweep, swap) rounds of a lazily-bounded run in one pass (csrc/fs_jquad.h k_jacobi_finish): pc = raw iterate n-2;
        pc_out <- what the reference leaves in p.current, pn <- what it leaves in p.next."""
        self._run("jacobi_finish", (pc_out._h, pn._h, pc._h, src._h), reads=[(pc, 2), (src, 1)], writes=[pc_out, pn])

    def rbsor_halfsweep_src(self, omega, parity, pn, pc, src):
        self._run("rbsor_halfsweep_src", (omega, parity, pn._h, pc._h, src._h), reads=[(pc, 1), (src, 0)], writes=[pn])

    def limit_field(self, limit, v, defer=False):               # fs/solver.py:38-43
        """defer=True (the solvers' end-of-step call, single GPU): the pass is not launched now - behind the buffer's flag it does nothing
        in a healthy run, yet its launch is a fifth of a small-grid step - but rides with the next kernel that touches the field: the
        velocity boundary kernel of the next step takes it along in its own launch (velocity_bc above), anything else - downloads,
        visualisation, the dye kernels, a hipGraph boundary - makes it run first (_run / Field.to_numpy).  Same results, one launch less."""
        if defer and self.limit_deferral and v.pending_limit is None and float(limit) ** 2 > 99.01:
            v.pending_limit = float(limit)
            return
        self.flush_limit(v)
        self._run("limit_field", (limit, v._h), pointwise=True, writes=[v])

    def clamp_field(self, low, high, f):                        # fs/solver.py:46-49
        self._run("clamp_field", (low, high, f._h), pointwise=True, writes=[f])

    def cip_advect_dye_clamped(self, dt, dx, fn, fxn, fyn, fc, fxc, fyc, v):
        """CIP advection of the dye with clamp_field(dye, 0, 1) folded into the store (fluid cells)."""
        self._run("cip_advect_dye_clamped", (dt, dx, fn._h, fxn._h, fyn._h, fc._h, fxc._h, fyc._h, v._h),
                  reads=[(fc, 1), (fxc, 1), (fyc, 1), (v, 1)], writes=[fn, fxn, fyn])

    def clamp_inflow(self, low, high, dye, defer=False):
        """defer=True (the dye solver's end-of-step call, single GPU): the cells this pass clamps are the targets of the dye boundary op list,
        which the next step's set_dye_boundary_condition overwrites with the scene's constants before anything reads them - the launch is
        dropped there (dye_bc), and made up for when anything else touches the field first (_run / Field.to_numpy)."""
        if defer and self.clamp_deferral and dye.pending_clamp is None:
            dye.pending_clamp = (float(low), float(high))
            return
        dye.pending_clamp = None        # (an older pending clamp to the same bounds is subsumed; the solvers use one pair of bounds)
        self._run("clamp_inflow", (low, high, dye._h), pointwise=True, writes=[dye])

    # visualisation kernels (fs/fluid_simulator.py:38-58, 121-126): every cell of the computed rows is written
    def vis_norm(self, rgb, v, p):
        self._run("vis_norm", (rgb._h, v._h, p._h), reads=[(v, 0), (p, 0)], full_writes=[rgb])

    def vis_pressure(self, rgb, p):
        self._run("vis_pressure", (rgb._h, p._h), reads=[(p, 0)], full_writes=[rgb])

    def vis_vorticity(self, dx, rgb, v):
        self._run("vis_vorticity", (dx, rgb._h, v._h), reads=[(v, 1)], full_writes=[rgb])

    def vis_dye(self, rgb, dye):
        self._run("vis_dye", (rgb._h, dye._h), reads=[(dye, 0)], full_writes=[rgb])

    def poisson_residual(self, dt, dx, p, vc):
        """(sum of squared Jacobi residuals, cell count) over all not-wall cells of the GLOBAL grid."""
        self.flush_limit(vc)
        if self.nranks > 1:
            stale = [f for f in (p, vc) if f.valid < 1]
            if stale:
                self.exchange_many(stale)
        s, n = self._p_residual(dt, dx, p._h, vc._h)
        return self._p_allreduce([s, n]) if self.nranks > 1 else (s, n)


class Device(DeviceBase):
    """DeviceBase bound to libfs_hip.so (HIP kernels on one MI355X; RCCL for the ghost rows)."""

    def __init__(self, nx, ny, dtype, gpu=0, rank=0, nranks=1, halo=None, bcast=None, allgather=None):
        super().__init__(nx, ny, dtype, gpu, rank, nranks, halo, bcast, allgather)
        self._lib = _lib.load()
        ctx = ctypes.c_void_p()
        _lib.call("fs_create", ctypes.byref(ctx), gpu, self.nx, self.ny, 0 if self.dtype == np.float32 else 1,
                  self.y0, self.nyl, self.halo)
        self._ctx = ctx
        self._graphs = []
        if nranks > 1 or (os.environ.get("FS_TEST_COMM") == "1" and bcast is not None):   # FS_TEST_COMM: 1-rank communicator (debug)
            self._has_comm = True
            uid = None
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                _lib.call("fs_comm_unique_id", buf)
                uid = buf.raw
            uid = bcast(uid)
            # RCCL prints a version banner on C stdout while initialising; keep stdout clean for callers that emit
            # machine-readable output (bench.py's single JSON line): route fd 1 to stderr for the duration.
            import sys
            sys.stdout.flush()
            libc = ctypes.CDLL(None)
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                _lib.call("fs_comm_init", self._ctx, rank, nranks, ctypes.c_char_p(uid))
            finally:
                libc.fflush(None)
                os.dup2(saved, 1)
                os.close(saved)

    # -- primitives -------------------------------------------------------------------------------------
    def _p_alloc(self, nchan):
        h = ctypes.c_void_p()
        _lib.call("fs_field_alloc", self._ctx, nchan, ctypes.byref(h))
        return h

    def _p_free(self, h):
        if self._ctx is not None and h:
            self._lib.fs_field_free(h)

    def _p_fill(self, h, value):
        _lib.call("fs_field_fill", h, value)

    def _p_upload(self, h, nchan, window, row_begin, nrows):
        _lib.call("fs_field_upload", h, window.ctypes.data_as(ctypes.c_void_p), row_begin, nrows)

    def _p_download(self, h, nchan, row_begin, nrows):
        out = np.empty((self.nx, nrows, nchan), dtype=self.dtype)
        _lib.call("fs_field_download", h, out.ctypes.data_as(ctypes.c_void_p), row_begin, nrows)
        return out

    def _p_upload_scene(self, bc_mask, bc_const, bc_dye):
        _lib.call("fs_upload_mask", self._ctx, bc_mask.ctypes.data_as(ctypes.c_void_p))
        _lib.call("fs_upload_bc_const", self._ctx, bc_const.ctypes.data_as(ctypes.c_void_p))
        if bc_dye is not None:
            _lib.call("fs_upload_bc_dye", self._ctx, bc_dye.ctypes.data_as(ctypes.c_void_p))
        rv, rp = ctypes.c_int(), ctypes.c_int()
        _lib.call("fs_bc_radius", self._ctx, ctypes.byref(rv), ctypes.byref(rp))
        return rv.value, rp.value

    def _p_kernel(self, name, *args):
        _lib.check(getattr(self._lib, "fs_" + name)(self._ctx, *args))

    def lazy_flags(self):
        """(flags[wave column, local row] uint8, rows the two-sweep kernel's general path owns without / with its vertical tile path) -
        diagnostic, include/fs_hip.h."""
        nw, rows, ng = ctypes.c_int(), ctypes.c_int(), (ctypes.c_int * 2)()
        _lib.call("fs_lazy_flags", self._ctx, None, 0, ctypes.byref(nw), ctypes.byref(rows), ng)
        out = np.empty((nw.value, rows.value), np.uint8)
        _lib.call("fs_lazy_flags", self._ctx, out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(nw), ctypes.byref(rows), ng)
        return out, (ng[0], ng[1])

    def _p_lazy_bc_ok(self):
        ok = ctypes.c_int()
        _lib.call("fs_lazy_bc_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    def _p_rb_pair_ok(self):
        ok = ctypes.c_int()
        _lib.call("fs_rbsor_pair_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    def _p_limit_deferral_ok(self):
        ok = ctypes.c_int()
        _lib.call("fs_velocity_bc_limit_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    def _p_dye_bc_limit_ok(self):
        ok = ctypes.c_int()
        _lib.call("fs_dye_bc_limit_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    has_cip_step = True      # fs_cip_step exists (include/fs_hip.h)

    @property
    def cip_step_fused(self):
        """fs_cip_step calls (whole grid; on a slab: any row range) evaluate K2 in registers: the post-K2 buffer then holds that velocity only on
        the inflow / outflow cells (csrc/fs_k234.h)."""
        ok = ctypes.c_int()
        _lib.call("fs_cip_step_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    def cip_step_tiles(self):
        """(all-fluid tiles, boundary tiles, tiles of the stand-alone K2 launch, rows per tile, cells per tile row) of a whole-grid fs_cip_step
        launch in its multi-part form (zeros otherwise; the third count: FS_FUSE_K2=1 only) - bench.py prices each part against the bytes of its own tiles."""
        v = [ctypes.c_int() for _ in range(5)]
        _lib.call("fs_cip_step_tiles", self._ctx, *[ctypes.byref(x) for x in v])
        return tuple(x.value for x in v)

    @property
    def jacobi_quad_ok(self):
        """The mask admits the four-sweep Jacobi pass (single GPU)."""
        ok = ctypes.c_int()
        _lib.call("fs_jacobi_quad_ok", self._ctx, ctypes.byref(ok))
        return bool(ok.value) and self.nranks == 1

    def _p_exchange(self, h, nchan, depth):
        _lib.call("fs_halo_exchange", self._ctx, h, depth)

    def _p_exchange_many(self, handles, depth):
        self._p_exchange_begin(handles, depth)
        self._p_exchange_wait()

    def _p_exchange_begin(self, handles, depth):
        arr = (ctypes.c_void_p * len(handles))(*[h for h, _, _ in handles])
        val = (ctypes.c_int * len(handles))(*[v for _, _, v in handles])
        _lib.call("fs_halo_exchange_begin_partial", self._ctx, arr, val, len(handles), depth)

    def _p_exchange_wait(self):
        _lib.call("fs_halo_exchange_wait", self._ctx)

    def _p_exchange_mark(self):
        _lib.call("fs_halo_exchange_mark", self._ctx)

    def _p_set_overlap(self, on):
        if getattr(self, "_has_comm", False):
            _lib.call("fs_comm_set_overlap", self._ctx, 1 if on else 0)

    def _p_max_over_ranks(self, values):
        if self.nranks == 1:
            return list(values)
        return [max(self.allgather_scalars(v)) for v in values]

    def _p_residual(self, dt, dx, ph, vh):
        s, n = ctypes.c_double(), ctypes.c_double()
        _lib.call("fs_poisson_residual", self._ctx, dt, dx, ph, vh, ctypes.byref(s), ctypes.byref(n))
        return s.value, n.value

    def _p_allreduce(self, values):
        arr = (ctypes.c_double * len(values))(*values)
        _lib.call("fs_allreduce_sum", self._ctx, arr, len(values))
        return tuple(arr)

    def sync(self):
        _lib.call("fs_sync", self._ctx)

    # -- tiny collectives over the RCCL communicator (benchmark plumbing: barrier, max-over-ranks) -------------
    def allgather_scalars(self, value):
        """[value of rank 0, value of rank 1, ...] on every rank (one ncclAllReduce of a one-hot vector)."""
        if self.nranks == 1:
            return [float(value)]
        if self.nranks > 16:
            raise ValueError("allgather_scalars supports up to 16 ranks")
        slots = [0.0] * self.nranks
        slots[self.rank] = float(value)
        return list(self._p_allreduce(slots))

    def barrier(self):
        """Device sync + rendezvous of all ranks."""
        self.sync()
        self.allgather_scalars(0.0)

    # -- hipGraph capture of a launch sequence (single GPU) ---------------------------------------------
    def capture(self, fn):
        """Run fn() once in stream-capture mode and return a graph id replayable with replay()."""
        _lib.call("fs_graph_begin", self._ctx)
        self.capturing = True
        try:
            fn()
        finally:
            self.capturing = False
            gid = ctypes.c_int(-1)
            _lib.call("fs_graph_end", self._ctx, ctypes.byref(gid))
        return gid.value

    def replay(self, graph_id, times=1):
        _lib.call("fs_graph_launch", self._ctx, graph_id, times)

    def free_graph(self, graph_id):
        _lib.call("fs_graph_free", self._ctx, graph_id)

    # -- command tape (slab runs): the logged period as C++ closures inside libfs_hip, replayed without Python ----------
    def _native_tape(self):
        """The C++ tape records what goes through libfs_hip.  A subclass that carries the ghost rows some other way (the
        single-GPU test harnesses: host copies, sockets) must replay its exchanges from Python."""
        cls = type(self)
        return all(getattr(cls, m) is f for m, f in _NATIVE_PRIMITIVES.items())

    def _p_tape_build(self, ops):
        if not self._native_tape():
            return None
        _lib.call("fs_tape_begin", self._ctx, 0)          # record only: nothing executes while the operations are re-issued
        try:
            for op in ops:
                self._issue(op)
        finally:
            tid = ctypes.c_int(-1)
            _lib.call("fs_tape_end", self._ctx, ctypes.byref(tid))
        return tid.value

    def _p_tape_replay(self, tape, times):
        if tape["id"] is None:
            return DeviceBase._p_tape_replay(self, tape, times)
        _lib.call("fs_tape_replay", self._ctx, tape["id"], times)

    def free_tape(self, tape):
        if tape.get("id") is not None and self._ctx is not None:
            _lib.call("fs_tape_free", self._ctx, tape["id"])
            tape["id"] = None

    def field_hot(self, f):
        if getattr(self, "capturing", False):      # (a download: not inside a capture - the caller asks again later)
            return False
        h = ctypes.c_int()
        _lib.call("fs_field_hot", f._h, ctypes.byref(h))
        return bool(h.value)

    def box_rates(self, nbytes, budget_ms=30.0):
        """(float4 read GB/s, float4 copy GB/s) of this GPU on buffers of `nbytes`, ~budget_ms each (measurement hygiene, include/fs_hip.h)."""
        rd, cp = ctypes.c_double(), ctypes.c_double()
        _lib.call("fs_box_rates", self._ctx, ctypes.c_size_t(int(nbytes)), float(budget_ms), ctypes.byref(rd), ctypes.byref(cp))
        return rd.value, cp.value

    def box_valu_rate(self, budget_ms=10.0):
        """1e9 f32 wave-instructions per second and SIMD at 4 waves per SIMD, no memory traffic: the box's clock under load, as the
        issue-bound kernels see it (include/fs_hip.h fs_box_valu_rate)."""
        r = ctypes.c_double()
        _lib.call("fs_box_valu_rate", self._ctx, float(budget_ms), ctypes.byref(r))
        return r.value

    def box_valu_pk_rate(self, budget_ms=10.0):
        """The same for PACKED f32 instructions (v_pk_mul_f32 / v_pk_add_f32, two operations per lane): what the packed transport kernels issue."""
        r = ctypes.c_double()
        _lib.call("fs_box_valu_pk_rate", self._ctx, float(budget_ms), ctypes.byref(r))
        return r.value

    def box_mixed_rate(self, nbytes, budget_ms=100.0):
        """GB/s of a float4 copy with K3+K4's instruction density on the way: memory system and SIMDs loaded together (fs_box_mixed_rate)."""
        r = ctypes.c_double()
        _lib.call("fs_box_mixed_rate", self._ctx, ctypes.c_size_t(int(nbytes)), float(budget_ms), ctypes.byref(r))
        return r.value

    # -- per-kernel HIP-event timing ---------------------------------------------------------------------
    def profile(self, on=True):
        _lib.call("fs_prof_enable", self._ctx, 1 if on else 0)

    def profile_reset(self):
        _lib.call("fs_prof_reset", self._ctx)

    def span_begin(self):
        """One HIP-event pair around everything queued until span_end() (which waits and returns the milliseconds)."""
        _lib.call("fs_span_begin", self._ctx)

    def span_end(self):
        ms = ctypes.c_double()
        _lib.call("fs_span_end", self._ctx, ctypes.byref(ms))
        return ms.value

    def profile_report(self):
        """{kernel name: (launches, total_ms)} accumulated since the last reset."""
        n = ctypes.c_int()
        _lib.call("fs_prof_count", self._ctx, ctypes.byref(n))
        out = {}
        for k in range(n.value):
            name = ctypes.create_string_buffer(64)
            launches, ms = ctypes.c_int(), ctypes.c_double()
            _lib.call("fs_prof_get", self._ctx, k, name, 64, ctypes.byref(launches), ctypes.byref(ms))
            out[name.value.decode()] = (launches.value, ms.value)
        return out

    def tile_list_stats(self):
        """(launch lists built so far, launches that wanted one and ran dense): fs_tile_list_stats."""
        built, misses = ctypes.c_int(), ctypes.c_int()
        _lib.call("fs_tile_list_stats", self._ctx, ctypes.byref(built), ctypes.byref(misses))
        return built.value, misses.value

    def profile_kernels(self, name):
        """The __global__ functions launched under profile name `name` since profiling was switched on, demangled without their signature
        ("fs::k_jacobi_ov2<4, 4>"): the names a rocprofv3 --kernel-trace of the same run shows (fs_prof_kernels)."""
        buf, n = ctypes.create_string_buffer(4096), ctypes.c_int()
        _lib.call("fs_prof_kernels", self._ctx, name.encode(), buf, 4096, ctypes.byref(n))
        return [l for l in buf.value.decode().split("\n") if l]

    def close(self):
        if self._ctx is not None:
            ctx, self._ctx = self._ctx, None
            self._lib.fs_destroy(ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# the libfs_hip-backed primitives a native tape can record (captured here: test harnesses rebind `runtime.Device`)
_NATIVE_PRIMITIVES = {m: Device.__dict__[m] for m in ("_p_exchange_begin", "_p_exchange_wait", "_p_exchange_mark", "_p_kernel")}
