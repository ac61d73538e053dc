#!/usr/bin/env python3
"""Headless counterpart of the reference's main.py (argparse flags of main.py:11-63, same defaults).

The reference opens a GGUI window and steps forever; here the loop runs `--steps` steps on the GPU and can
  * dump fields exactly like the reference's `d` key (main.py:129-132):  output/step_{step:06}.npz  with v, p[, dye]
  * write the visualisation the window would show (`-vis`, main.py:94-107) as PNG frames every `--frame-every` steps
  * write / read a full-state checkpoint (new: the reference cannot resume - its dump lacks the CIP gradient
    fields and the `next` buffers).
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import fs  # noqa: E402
from fs.fluid_simulator import DyeFluidSimulator, FluidSimulator  # noqa: E402

_STATE = ("v", "p", "vx", "vy", "dye", "dyex", "dyey")


def build_parser():
    p = argparse.ArgumentParser(description="Fluid Simulator (headless, MI355X)")
    p.add_argument("-bc", "--boundary_condition", type=int, choices=[1, 2, 3, 4, 5, 6], default=1, help="Boundary condition number")
    p.add_argument("-re", "--reynolds_num", type=float, default=1000000.0, help="Reynolds number")
    p.add_argument("-res", "--resolution", type=int, default=400, help="Resolution of y-axis")
    p.add_argument("-dt", "--time_step", type=float, default=0.0, help="Time step")
    p.add_argument("-vis", "--visualization", type=int, choices=[0, 1, 2, 3], default=0, help="Flow visualization type")
    p.add_argument("-vc", "--vorticity_confinement", type=float, default=5.0, help="Vorticity Confinement. 0.0 is disable.")
    p.add_argument("-scheme", "--advection_scheme", type=str, choices=["upwind", "kk", "cip"], default="cip", help="Advection Scheme")
    p.add_argument("-no_dye", "--no_dye", action="store_true", help="No dye calculation")
    p.add_argument("-cpu", "--cpu", action="store_true", help="accepted for compatibility; there is no CPU path (HIP only)")
    # headless additions
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--f64", action="store_true", help="double precision (the reference is f32 only)")
    p.add_argument("--dump-every", type=int, default=0, help="np.savez v, p[, dye] every N steps (reference key 'd')")
    p.add_argument("--frame-every", type=int, default=0, help="write the -vis image as PNG every N steps")
    p.add_argument("--graph", action="store_true",
                   help="replay the steps between two dumps / frames as a hipGraph (no Python between kernels; same results)")
    p.add_argument("--out", type=str, default="output")
    p.add_argument("--save-state", type=str, default=None, help="write a full-state checkpoint (.npz) after the last step")
    p.add_argument("--load-state", type=str, default=None, help="resume from a checkpoint written by --save-state")
    return p


def _npz_path(path):
    """np.savez appends '.npz' to a name without it; use the same name for writing and reading."""
    path = str(path)
    return path if path.endswith(".npz") else path + ".npz"


def save_state(sim, path, step):
    s = sim._solver
    arrays = {"step": np.array(step)}
    for name in _STATE:
        if hasattr(s, name):
            arrays[f"{name}.current"] = getattr(s, name).current.to_numpy()
            arrays[f"{name}.next"] = getattr(s, name).next.to_numpy()
    vc = s.vorticity_confinement
    if vc is not None:
        arrays["vorticity"] = vc.vorticity.to_numpy()
        arrays["vorticity_abs"] = vc.vorticity_abs.to_numpy()
    np.savez(_npz_path(path), **arrays)


def load_state(sim, path):
    s = sim._solver
    z = np.load(_npz_path(path))
    for name in _STATE:
        if hasattr(s, name):
            getattr(s, name).current.from_numpy(z[f"{name}.current"])
            getattr(s, name).next.from_numpy(z[f"{name}.next"])
    vc = s.vorticity_confinement
    if vc is not None and "vorticity" in z:
        vc.vorticity.from_numpy(z["vorticity"])
        vc.vorticity_abs.from_numpy(z["vorticity_abs"])
    return int(z["step"])


def frame(sim, vis):
    """The image the reference's window would show (main.py:93-107), downloaded as an (X, Y, 3) array."""
    if vis == 0:
        return sim.get_norm_field().to_numpy()
    if vis == 1:
        return sim.get_pressure_field().to_numpy()
    if vis == 2:
        return sim.get_vorticity_field().to_numpy()
    return sim.get_dye_field().to_numpy()


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.cpu:
        print("note: -cpu ignored; this build runs on the GPU only", file=sys.stderr)
    res = args.resolution
    dt = args.time_step if args.time_step != 0.0 else 0.05 / res
    dx = 1 / res
    vor_eps = args.vorticity_confinement if args.vorticity_confinement != 0.0 else None
    enable_dye = not args.no_dye
    if args.visualization == 3 and not enable_dye:
        raise SystemExit("-vis 3 (dye) needs dye transport (drop -no_dye)")
    if args.boundary_condition == 6:
        from fs.boundary_condition import _find_obstacle_image
        try:
            _find_obstacle_image()
        except FileNotFoundError as e:      # the obstacle image is an asset of the reference repository and is not shipped here
            parser.error(f"-bc 6: {e}")
    print(f"Boundary Condition: {args.boundary_condition}\ndt: {dt}\nRe: {args.reynolds_num}\nResolution: {res}\n"
          f"Scheme: {args.advection_scheme}\nVorticity confinement: {vor_eps}")
    fs.runtime.init(gpu=args.gpu, dtype="f64" if args.f64 else "f32")
    cls = DyeFluidSimulator if enable_dye else FluidSimulator
    sim = cls.create(args.boundary_condition, res, dt, dx, args.reynolds_num, vor_eps, args.advection_scheme)
    out = Path(args.out)
    step0 = load_state(sim, args.load_state) if args.load_state else 0
    dev = sim._solver._bc.device
    t0 = time.perf_counter()
    step, last = step0, step0 + args.steps
    while step < last:
        if args.frame_every and step % args.frame_every == 0:
            from PIL import Image
            out.mkdir(exist_ok=True)
            img = np.clip(frame(sim, args.visualization), 0.0, 1.0)
            Image.fromarray((np.flip(img.transpose(1, 0, 2), axis=0) * 255).astype(np.uint8)).save(out / f"{step:06}.png")
        # steps until the next frame / dump / end: one chunk (a hipGraph replay with --graph, a plain loop otherwise)
        nxt = last
        if args.frame_every:
            nxt = min(nxt, (step // args.frame_every + 1) * args.frame_every)
        if args.dump_every:
            nxt = min(nxt, (step // args.dump_every + 1) * args.dump_every)
        sim.run(nxt - step, graph=args.graph)
        step = nxt
        if args.dump_every and step % args.dump_every == 0:
            out.mkdir(exist_ok=True)
            np.savez(str(out / f"step_{step:06}.npz"), **sim.field_to_numpy())
    dev.sync()
    el = time.perf_counter() - t0
    print(f"{args.steps} steps in {el:.3f} s = {args.steps / el:.1f} steps/s")
    if args.save_state:
        save_state(sim, args.save_state, step0 + args.steps)
    dev.close()


if __name__ == "__main__":
    main()
