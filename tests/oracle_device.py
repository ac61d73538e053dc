"""CPU stand-in for fs.runtime.Device, for testing the HOST logic of the y-slab decomposition without a GPU.

TEST INFRASTRUCTURE: implements the primitive hooks of fs.runtime.DeviceBase with NumPy arrays, the CPU
oracle as the kernel provider and torch.distributed (gloo) as the ghost-row transport.  Everything above the
primitives - slab geometry, ghost-row validity tracking, which exchange happens when, the reference's solver
orchestration - is the product's own Python code, exercised unchanged.

A kernel launch changes only the local rows [lo, hi) it was asked for (the oracle computes all rows; the others are
restored).  After every kernel (all its launches - an overlapped exchange splits it into three) the rows of the fields
it wrote that lie OUTSIDE the computed range are POISONED with NaN: if the host logic ever lets a stencil read a stale
ghost row, the NaN reaches the owned rows and the comparison with the single-domain result fails.
"""
import numpy as np
import torch
import torch.distributed as dist
from fs.runtime import DeviceBase
from oracle import oracle as O


class _Arr:
    def __init__(self, a):
        self.a = a


class OracleSlabDevice(DeviceBase):
    def __init__(self, nx, ny, dtype, gpu=0, rank=0, nranks=1, halo=None, bcast=None, allgather=None):
        super().__init__(nx, ny, dtype, gpu, rank, nranks, halo, bcast, allgather)
        self.r_off = self.g_lo - (self.y0 - self.halo)     # local row of the first in-domain row
        self.nloc = self.g_hi - self.g_lo                  # rows actually stored (in-domain only)
        self.poison = nranks > 1
        self.overlap_stream = True     # (stands for a context whose exchanges run on their own stream: the tape compiler may hoist begins)

    # ---- primitives -------------------------------------------------------------------------------
    def _shape(self, nchan):
        return (self.nx, self.nloc) if nchan == 1 else (self.nx, self.nloc, nchan)

    def _p_alloc(self, nchan):
        return _Arr(np.zeros(self._shape(nchan), self.dtype))

    def _p_free(self, h):
        pass

    def _p_fill(self, h, value):
        h.a[...] = value

    def _p_upload(self, h, nchan, window, row_begin, nrows):
        a0 = row_begin - self.r_off
        h.a[:, a0:a0 + nrows] = window.reshape(self._shape(nchan)[:1] + (nrows,) + self._shape(nchan)[2:])

    def _p_download(self, h, nchan, row_begin, nrows):
        a0 = row_begin - self.r_off
        return np.ascontiguousarray(h.a[:, a0:a0 + nrows]).reshape(self.nx, nrows, nchan)

    def _p_upload_scene(self, bc_mask, bc_const, bc_dye):
        sl = slice(self.g_lo, self.g_hi)
        self.obc = O.OracleBC(bc_const[:, sl], bc_mask[:, sl], None if bc_dye is None else bc_dye[:, sl], self.dtype)
        # cells the pressure boundary kernel assigns (every cell gets a unique value; an assigned cell takes a neighbour's, a mean or 0)
        probe = (np.arange(self.nx * self.nloc, dtype=np.float64).reshape(self.nx, self.nloc) + 1.0).astype(self.dtype)
        after = probe.copy()
        self.obc.set_pressure_boundary_condition(after)
        self.p_targets = after != probe
        probe2 = np.stack([probe + 0.25, -probe - 0.75], axis=2).astype(self.dtype)              # likewise for the velocity boundary kernel
        after2 = probe2.copy()
        self.obc.set_velocity_boundary_condition(after2)
        self.v_touched = (after2 != probe2).any(axis=2) | (self.obc.mask != 1)
        # the two-iteration red-black pass reads 4 rows beyond what it writes only on masks without one-cell-thin walls between fluid
        # regions (csrc/fs_rbpair.h; the library decides per recipe - this is the conservative restatement on the global mask)
        m = np.pad(np.asarray(bc_mask), 1, constant_values=1)
        fl, solid = m == 0, m[1:-1, 1:-1] != 0
        thin = solid & ((fl[:-2, 1:-1] & fl[2:, 1:-1]) | (fl[1:-1, :-2] & fl[1:-1, 2:]))
        thin |= (m[1:-1, 1:-1] == 2) & fl[:-2, 1:-1]
        self.pair_ok = not thin.any() and bool((bc_mask[:, 0] == 1).all() and (bc_mask[:, -1] == 1).all())
        return 2, 1

    def _p_rb_pair_ok(self):
        return self.pair_ok

    def _p_exchange(self, h, nchan, depth, v=0):
        """Ghost rows at depth offsets [v, depth) on each side (v = rows the tracker still trusts: they are NOT refreshed, so
        a wrong validity count leaves poisoned rows in place)."""
        if v >= depth:
            return
        a = h.a
        lo = self.halo - self.r_off                # array row of the first owned row
        hi = lo + self.nyl
        ops, recvs = [], []

        def rows(s):
            return torch.from_numpy(np.ascontiguousarray(a[:, s]))

        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, rows(slice(lo + v, lo + depth)), self.rank - 1))
            t = rows(slice(lo - depth, lo - v)); recvs.append((slice(lo - depth, lo - v), t))
            ops.append(dist.P2POp(dist.irecv, t, self.rank - 1))
        if self.rank < self.nranks - 1:
            ops.append(dist.P2POp(dist.isend, rows(slice(hi - depth, hi - v)), self.rank + 1))
            t = rows(slice(hi + v, hi + depth)); recvs.append((slice(hi + v, hi + depth), t))
            ops.append(dist.P2POp(dist.irecv, t, self.rank + 1))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for s, t in recvs:
            a[:, s] = t.numpy()

    def _p_exchange_many(self, handles, depth):
        # both sides of a slab boundary must name the same fields in the same order (the packed RCCL message has no labels): compare
        # the list with each neighbour before any row moves
        mine = torch.tensor([[self._handle_serial.get(id(h), -1), nchan, v] for h, nchan, v in handles] + [[-7, len(handles), depth]],
                            dtype=torch.int64)
        for nb in (self.rank - 1, self.rank + 1):
            if 0 <= nb < self.nranks:
                theirs = torch.full((64, 3), -99, dtype=torch.int64)
                pad = torch.full((64, 3), -99, dtype=torch.int64)
                pad[:len(mine)] = mine
                for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, pad, nb), dist.P2POp(dist.irecv, theirs, nb)]):
                    r.wait()
                assert torch.equal(pad, theirs), f"rank {self.rank} and {nb} disagree on the exchanged fields:\n{pad[:len(mine)]}\n{theirs[:len(mine) + 2]}"
        for h, nchan, v in handles:
            self._p_exchange(h, nchan, depth, v)

    def _p_max_over_ranks(self, values):
        if self.nranks == 1:
            return list(values)
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    # ---- kernels: same argument order as the C-ABI (include/fs_hip.h) --------------------------------
    def _p_kernel(self, name, *args):
        *args, lo, hi = args
        b, X, Y, dt_ = self.obc, self.nx, self.nloc, self.dtype
        A = [x.a if isinstance(x, _Arr) else x for x in args]
        saved = [(x, x.copy()) for x in A if isinstance(x, np.ndarray)]
        written = []
        if name == "velocity_bc":
            b.set_velocity_boundary_condition(A[0]); written = [A[0]]
        elif name == "pressure_bc":
            b.set_pressure_boundary_condition(A[0]); written = [A[0]]
        elif name == "dye_bc":
            b.set_dye_boundary_condition(A[0]); written = [A[0]]
        elif name == "mac_update":
            scheme, dt, dx, re, vn, vc, pc = A
            O._call("oracle_mac_update", dt_, X, Y, dt, dx, re, scheme, b.mask, vn, vc, pc); written = [vn]
        elif name == "mac_dye":
            scheme, dt, dx, dn, dc, vc = A
            O._call("oracle_mac_dye", dt_, X, Y, dt, dx, 1.0, scheme, b.mask, dn, dc, vc); written = [dn]
        elif name == "cip_set_grad":
            dx, fx, fy, f = A
            O._call("oracle_cip_set_grad", dt_, X, Y, dx, f.shape[2], fx, fy, f); written = [fx, fy]
        elif name == "cip_nonadv":
            dt, dx, re, fn, fc, pc = A
            O._call("oracle_cip_nonadv", dt_, X, Y, dt, dx, re, b.mask, fn, fc, pc); written = [fn]
        elif name == "cip_nonadv_dye":
            dt, dx, re, dn, dc = A
            O._call("oracle_cip_nonadv_dye", dt_, X, Y, dt, dx, re, b.mask, dn, dc); written = [dn]
        elif name == "cip_nonadv_grad":
            dx, fxn, fyn, fxc, fyc, fc, fn = A
            O._call("oracle_cip_nonadv_grad", dt_, X, Y, dx, fc.shape[2], b.mask, fxn, fyn, fxc, fyc, fc, fn); written = [fxn, fyn]
        elif name == "cip_advect":
            dt, dx, fn, fxn, fyn, fc, fxc, fyc, v = A
            O._call("oracle_cip_advect", dt_, X, Y, dt, dx, fc.shape[2], b.mask, fn, fxn, fyn, fc, fxc, fyc, v); written = [fn, fxn, fyn]
        elif name == "cip_grad_advect":
            dt, dx, vo, gxo, gyo, fn, fc, gxc, gyc, full = A
            O._call("oracle_cip_nonadv_grad", dt_, X, Y, dx, 2, b.mask, gxo, gyo, gxc, gyc, fc, fn)     # K3 into the output buffers
            tx, ty = gxc.copy(), gyc.copy()                                                             # K4 targets: old gradient buffers
            st = np.ones_like(self.v_touched) if full else self.v_touched       # what the kernel carries (include/fs_hip.h)
            vo[st] = fc[st]
            O._call("oracle_cip_advect", dt_, X, Y, dt, dx, 2, b.mask, vo, tx, ty, fn, gxo, gyo, fn)
            nw = b.mask != 1
            gxo[nw] = tx[nw]; gyo[nw] = ty[nw]                     # not-wall cells: K4 result (fluid) or carried old gradient
            written = [vo, gxo, gyo]
        elif name == "cip_grad_advect_dye":
            dt, dx, do, gxo, gyo, fn, fc, gxc, gyc, v, clamp01, full = A
            O._call("oracle_cip_nonadv_grad", dt_, X, Y, dx, 3, b.mask, gxo, gyo, gxc, gyc, fc, fn)     # K3 into the output buffers
            tx, ty = gxc.copy(), gyc.copy()                                                             # K4 targets: old gradient buffers
            st = np.ones_like(self.v_touched) if full else (b.mask != 1)
            do[st] = fc[st]
            O._call("oracle_cip_advect", dt_, X, Y, dt, dx, 3, b.mask, do, tx, ty, fn, gxo, gyo, v)
            if clamp01:
                fl = b.mask == 0
                do[fl] = np.fmin(np.fmax(do[fl], dt_.type(0)), dt_.type(1))
            nw = b.mask != 1
            gxo[nw] = tx[nw]; gyo[nw] = ty[nw]                     # not-wall cells: K4 result (fluid) or carried old gradient
            written = [do, gxo, gyo]
        elif name == "vort_calc":
            dx, w, wa, vc = A
            O._call("oracle_vort_calc", dt_, X, Y, dx, b.mask, w, wa, vc); written = [w, wa]
        elif name == "vort_add":
            dt, dx, weight, vn, vc, w, wa = A
            O._call("oracle_vort_add", dt_, X, Y, dt, dx, weight, b.mask, vn, vc, w, wa); written = [vn]
        elif name == "vort_confine":
            dt, dx, weight, vn, vc, w, wa = A
            if w is None:
                w, wa = np.zeros((X, Y), dt_), np.zeros((X, Y), dt_)
            else:
                written = [w, wa]
            O._call("oracle_vort_calc", dt_, X, Y, dx, b.mask, w, wa, vc)
            O._call("oracle_vort_add", dt_, X, Y, dt, dx, weight, b.mask, vn, vc, w, wa); written = written + [vn]
        elif name == "jacobi_sweep":
            dt, dx, pn, pc, vc = A
            O._call("oracle_jacobi_sweep", dt_, X, Y, dt, dx, b.mask, pn, pc, vc); written = [pn]
        elif name == "rbsor_halfsweep":
            dt, dx, omega, parity, pn, pc, vc = A
            O._call("oracle_rbsor_half", dt_, X, Y, dt, dx, omega, parity ^ (self.g_lo & 1), b.mask, pn, pc, vc); written = [pn]
        elif name == "rbsor_iteration":
            dt, dx, omega, pn, pc, vc = A
            par = self.g_lo & 1
            O._call("oracle_rbsor_half", dt_, X, Y, dt, dx, omega, 1 ^ par, b.mask, pn, pc, vc)
            O._call("oracle_rbsor_half", dt_, X, Y, dt, dx, omega, 0 ^ par, b.mask, pn, pn, vc); written = [pn]
        elif name == "rbsor_pair":
            dt, dx, omega, co, no, pc, pn, vc, full = A
            par = self.g_lo & 1
            a, n_ = pc.copy(), pn.copy()
            for cur, nxt in ((a, n_), (n_, a)):
                b.set_pressure_boundary_condition(cur)
                O._call("oracle_rbsor_half", dt_, X, Y, dt, dx, omega, 1 ^ par, b.mask, nxt, cur, vc)
                O._call("oracle_rbsor_half", dt_, X, Y, dt, dx, omega, 0 ^ par, b.mask, nxt, nxt, vc)
            st = np.ones_like(self.p_targets) if full else ((b.mask == 0) | self.p_targets)      # what the kernel stores (include/fs_hip.h)
            co[st] = a[st]; no[st] = n_[st]; written = [co, no]
        elif name == "cip_advect_dye_clamped":
            dt, dx, fn, fxn, fyn, fc, fxc, fyc, v = A
            O._call("oracle_cip_advect", dt_, X, Y, dt, dx, 3, b.mask, fn, fxn, fyn, fc, fxc, fyc, v)
            fl = b.mask == 0
            fn[fl] = np.fmin(np.fmax(fn[fl], dt_.type(0)), dt_.type(1)); written = [fn, fxn, fyn]
        elif name == "clamp_inflow":
            lo_, hi_, d = A
            m2 = b.mask == 2
            d[m2] = np.fmin(np.fmax(d[m2], dt_.type(lo_)), dt_.type(hi_))
        elif name == "limit_field":
            O.limit_field(A[1], A[0])
        elif name == "clamp_field":
            O.clamp_field(A[2], A[0], A[1])
        else:
            raise NotImplementedError(name)
        a_lo, a_hi = max(lo - self.r_off, 0), max(hi - self.r_off, 0)
        for arr, old in saved:                       # rows outside [lo, hi) keep their previous content
            arr[:, :a_lo] = old[:, :a_lo]
            arr[:, a_hi:] = old[:, a_hi:]

    def _after_kernel(self, name, written, lo, hi):
        if self.poison:
            a_lo, a_hi = max(lo - self.r_off, 0), max(hi - self.r_off, 0)
            for f in written:
                f._h.a[:, :a_lo] = np.nan
                f._h.a[:, a_hi:] = np.nan

    def _p_tape_replay(self, tape, times):
        """Replay through the same primitives, with the same poisoning as the eager path: rows of a kernel's outputs outside its
        computed range become NaN, so an exchange moved too far up (or dropped) by the tape compiler fails the comparison."""
        for _ in range(times):
            for op in tape["ops"]:
                self._issue(op)
                if op[0] == "k" and self.poison and op[3]:
                    lo, hi = op[2][-2:]
                    a_lo, a_hi = max(lo - self.r_off, 0), max(hi - self.r_off, 0)
                    for a in op[2]:
                        if isinstance(a, _Arr) and id(a) in op[3]:
                            a.a[:, :a_lo] = np.nan
                            a.a[:, a_hi:] = np.nan

    def sync(self):
        pass

    def close(self):
        pass
