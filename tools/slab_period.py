#!/usr/bin/env python3
"""Ghost-row bookkeeping of a slab run without a GPU: grouped exchanges per step and the period of the pattern (what a command tape
needs) for the headline solver at several halo depths, with and without the two-iteration red-black pass.  tools/slab_period.py"""
import sys, importlib, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); importlib.import_module("2d-fluid-simulator_amd")
import fs
from fs.runtime import DeviceBase
from fs.boundary_condition import BoundaryCondition, create_scene_arrays
class Null(DeviceBase):
    def _p_alloc(self, n): return object()
    def _p_free(self, h): pass
    def _p_fill(self, h, v): pass
    def _p_kernel(self, name, *a): pass
    def _p_exchange(self, *a): pass
    def _p_upload_scene(self, *a): return 2, 1
    def _p_lazy_bc_ok(self): return True
    def _p_rb_pair_ok(self): return PAIR
    def _p_max_over_ranks(self, v): return list(v)
    dtype_=np.float32
for PAIR in (True, False):
  for halo in (8, 12, 16, 20, 24, 32):
    res=512
    const, mask, _ = create_scene_arrays(5, res)
    dev = Null(mask.shape[0], mask.shape[1], np.float32, rank=1, nranks=4, halo=halo)
    dt, dx = 0.05/res, 1.0/res
    bc = BoundaryCondition(const, mask, device=dev)
    vc = fs.VorticityConfinement(bc, dt, dx, 5.0)
    pu = fs.RedBlackSorPressureUpdater(bc, dt, dx, 1.3, 2, pair=PAIR)
    solver = fs.CipMacSolver(bc, pu, dt, dx, 1e6, vc)
    ex=[]
    for step in range(60):
        n0=dev.n_exchanges; solver.update(); ex.append(dev.n_exchanges-n0)
        if step == 19: b0 = dev.n_exchanged_bytes
    # find period of the exchange-count sequence in the tail
    tail=ex[20:]
    per=next((P for P in range(1,21) if all(tail[i]==tail[i+P] for i in range(len(tail)-P))), None)
    t = dev.tape_period(solver.update, nsteps=2)
    print('pair',PAIR,'halo',halo,'exch/step', ''.join(map(str,ex[:40])), 'period',per, 'tape', None if t is None else t['nsteps'], 'avg', sum(tail)/len(tail), 'KB/step/neighbour at X=8192: %.0f' % ((dev.n_exchanged_bytes - b0) / 40 / 1024 * 8192 / dev.nx))
