#!/usr/bin/env python3
"""What a small grid's kernel is made of: K back-to-back launches of ONE kernel in a hipGraph, replayed; the marginal time per launch.
    python tools/r4_chain.py [--res 200] [--bc 1]
Env as for bench.py (FS_RBPAIR_RT, FS_RBPAIR_SPLIT ...)."""
import argparse, importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("2d-fluid-simulator_amd")
import fs

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=200); ap.add_argument("--bc", type=int, default=1)
a = ap.parse_args()
res = a.res; dt, dx = 0.0005, 1.0 / res
fs.runtime.init(gpu=0)
sim = fs.FluidSimulator.create(a.bc, res, dt, dx, 1000.0, None, "upwind")
s = sim._solver; dev = s._dev
for _ in range(10): sim.step()
dev.sync()
v = s.v.current
A, B, C, D = s.p.current, s.p.next, dev.alloc(1), dev.alloc(1)
vn = dev.alloc(2)
dye = None

def pair2():
    dev.rbsor_pair(dt, dx, 1.3, C, D, A, B, v)
    dev.rbsor_pair(dt, dx, 1.3, A, B, C, D, v)
def mac2():
    dev.mac_update(s._advect.code, dt, dx, 1000.0, vn, v, A)
    dev.mac_update(s._advect.code, dt, dx, 1000.0, vn, v, A)
def bc2():
    dev.pressure_bc(C); dev.pressure_bc(C)
def iter2():
    dev.rbsor_iteration(dt, dx, 1.3, B, A, v); dev.rbsor_iteration(dt, dx, 1.3, A, B, v)

def measure(name, fn2):
    out = []
    for K in (1, 4, 16):
        g = dev.capture(lambda: [fn2() for _ in range(K)])
        dev.replay(g, 50); dev.sync()
        n = 400
        t0 = time.perf_counter(); dev.replay(g, n); dev.sync(); t = time.perf_counter() - t0
        out.append(t / n * 1e6)
        dev.free_graph(g)
    per = (out[2] - out[0]) / (2 * 15)
    print(f"{name:10s} graph of 2 / 8 / 32 launches: {out[0]:7.2f} {out[1]:7.2f} {out[2]:7.2f} us  -> {per:5.2f} us per launch, {out[0] - 2 * per:5.2f} us per replay")
pair2(); mac2(); bc2(); iter2(); dev.sync()
measure("pair", pair2); measure("mac", mac2); measure("p_bc", bc2); measure("rb_iter", iter2)
dev.close()
