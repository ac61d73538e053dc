#!/usr/bin/env python3
"""Do the box probes read differently under SUSTAINED load?  The three probes of bench.py (copy, VALU, mixed) at process start, then again directly
after N seconds of the headline step, next to the per-kernel times of that run.  (the probes, run cold, read the same in the
box's fast and slow state.)   python tools/r4_sustained_probe.py [steps]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("2d-fluid-simulator_amd")
import fs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
res = 4096
fs.runtime.init(gpu=0)
sim = fs.FluidSimulator.create(5, res, 0.05 / res, 1.0 / res, 1e6, 5.0, "cip")
dev = sim._solver._bc.device
nb = 2 * 8192 * 4096 * 4


def probes(tag):
    rd, cp = dev.box_rates(nb, 30.0)
    print(f"{tag:28s} copy {cp:7.1f} GB/s  valu {dev.box_valu_rate(10.0):.3f} G/SIMD  mixed {dev.box_mixed_rate(nb, 100.0):7.1f} GB/s", flush=True)


probes("cold (process start)")
sim.run(60)
dev.sync()
t0 = time.perf_counter(); sim.run(steps); dev.sync(); t = time.perf_counter() - t0
print(f"{steps} steps: {steps / t:.1f} steps/s ({t:.2f} s of sustained load)", flush=True)
probes("directly after the load")
probes("again")
t0 = time.perf_counter(); sim.run(600); dev.sync(); t = time.perf_counter() - t0
print(f"600 more steps: {600 / t:.1f} steps/s", flush=True)
dev.close()
