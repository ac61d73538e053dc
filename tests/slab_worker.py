"""Worker for tests/test_slab_gloo.py: one rank of a world_size-N gloo job running the product's host logic
(fs.runtime.DeviceBase + solvers) on the CPU stand-in device."""
import os
import sys

import numpy as np


def run(rank, world, port, fname, halo, out_dir, tape=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    for p in (repo, os.path.join(repo, "2d-fluid-simulator_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fs
    from helpers import dead_buffers, make_product, traj_config
    from oracle_device import OracleSlabDevice

    def allgather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    g = np.load(os.path.join(here, "golden", fname))
    cfg = traj_config(g)
    # the ranks of a real job do not share an address-space layout: shift this rank's heap so that anything ordered by object address
    # (sets of fields, id()-keyed dicts) comes out differently on every rank
    run._ballast = [type("Ballast", (), {"__slots__": ()})() for _ in range(rank * 1237)] + [{"k": i} for i in range(rank * 311)]
    fs.runtime.init(dtype="f64" if cfg["fp64"] else "f32", rank=rank, nranks=world, halo=halo, allgather=allgather,
                    device_cls=OracleSlabDevice)
    sim = make_product(g, cfg)
    dev = sim._solver._bc.device
    bad = []
    last = max(cfg["snaps"])
    hoisted = 0
    if tape:
        # command tape: log periods of 2 steps until steady, compile (exchange begins moved up), replay up to the last snapshot
        done = [0]

        def counted():
            sim.step()
            done[0] += 1
        t = dev.tape_period(counted, nsteps=2)
        assert t is not None, "no steady period found"
        hoisted = sum(1 for a, b in zip(t["ops"], t["ops"][1:]) if a[0] == "begin" and b[0] != "wait") + len(t["prologue"])
        dev.replay_tape(t, 2)
        total = done[0] + 2 * t["nsteps"]
        sim.step()                      # and the eager path carries on from the replayed state
        total += 1
        out = sim.field_to_numpy()
        if rank == 0:                   # the golden trajectories end at step 10-20: the single-domain ORACLE run is the reference
            from helpers import make_oracle
            ref = make_oracle(g, cfg)
            for _ in range(total):
                ref.update()
            for k, e in ref.fields().items():
                if not np.array_equal(out[k], e, equal_nan=True):
                    bad.append(f"step{total}.{k}")
        last = total
    for step in range(1, (0 if tape else last) + 1):
        sim.step()
        if step in cfg["snaps"]:
            for k, a in sim.field_to_numpy().items():
                if not np.array_equal(a, g[f"step{step}.{k}"]):
                    bad.append(f"step{step}.{k}")
    # internal buffers too (stale-cell choreography across slabs)
    s = sim._solver
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
        if f"final.{name}.current" in g and not tape:
            for which in ("current", "next"):
                if f"{name}.{which}" in dead_buffers(s):
                    continue
                if not np.array_equal(getattr(getattr(s, name), which).to_numpy(), g[f"final.{name}.{which}"]):
                    bad.append(f"final.{name}.{which}")
    if rank == 0:
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{len(bad)} {dev.n_exchanges / last:.2f} {' '.join(bad)}\n")
            if tape:
                f.write(f"{hoisted}\n")
    dist.barrier()
    dist.destroy_process_group()


def run_default_halo(rank, world, port, res, out_dir):
    """Slab run with the DEFAULT halo depth (halo=None) on a grid whose slab heights straddle the 128-row threshold of the
    default (ny % world != 0): every rank must pick the same depth, and the result must equal the single-domain oracle run."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    for p in (repo, os.path.join(repo, "2d-fluid-simulator_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fs
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    from oracle_device import OracleSlabDevice

    def allgather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    dt, dx, re, vc = 0.05 / res, 1.0 / res, 1.0e6, 5.0
    fs.runtime.init(dtype="f32", rank=rank, nranks=world, halo=None, allgather=allgather, device_cls=OracleSlabDevice)
    sim = fs.FluidSimulator.create(2, res, dt, dx, re, vc, "cip")
    dev = sim._solver._bc.device
    halos = allgather((dev.halo, dev.nyl))
    for _ in range(3):
        sim.step()
    out = sim.field_to_numpy()
    if rank == 0:
        const, mask, _ = create_scene_arrays(2, res)
        ref = O.make_simulator(const, mask, None, scheme="cip", dt=dt, dx=dx, re=re, vor_eps=vc)
        for _ in range(3):
            ref.update()
        bad = [k for k, e in ref.fields().items() if not np.array_equal(out[k], e)]
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{len(bad)} {sorted(set(h for h, _ in halos))} {sorted(set(n for _, n in halos))}\n")
    dist.barrier()
    dist.destroy_process_group()


def run_scene(rank, world, port, bc, res, scheme, vc, updater, halo, steps, tape, out_dir):
    """A reference scene at (bc, res) cut into `world` slabs (halo None: the default depth), `steps` steps eagerly - or, with `tape`, a logged
    period replayed as bench.py's N > 1 timed loop does - against the single-domain ORACLE run (no golden file needed: the 4- and 8-way cuts
    of BASELINE configs[3]'s scene, bc2 CIP + VC, at res 64 / 128)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    for p in (repo, os.path.join(repo, "2d-fluid-simulator_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fs
    from fs.boundary_condition import create_scene_arrays
    from oracle import oracle as O
    from oracle_device import OracleSlabDevice

    def allgather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    dt, dx, re = 0.05 / res, 1.0 / res, 1.0e6
    fs.runtime.init(dtype="f32", rank=rank, nranks=world, halo=halo, allgather=allgather, device_cls=OracleSlabDevice)
    sim = fs.FluidSimulator.create(bc, res, dt, dx, re, vc, scheme, pressure_updater=updater)
    dev = sim._solver._bc.device
    total = 0
    period = 0
    if tape:
        done = [0]

        def counted():
            sim.step()
            done[0] += 1
        t = dev.tape_period(counted, nsteps=2)
        assert t is not None, "no steady period found"
        period = t["nsteps"]
        dev.replay_tape(t, 2)
        total = done[0] + 2 * t["nsteps"]
    for _ in range(steps):
        sim.step()
    total += steps
    out = sim.field_to_numpy()
    geo = allgather((dev.halo, dev.nyl))
    if rank == 0:
        const, mask, _ = create_scene_arrays(bc, res)
        ref = O.make_simulator(const, mask, None, scheme=scheme, dt=dt, dx=dx, re=re, vor_eps=vc, updater=updater or ("rbsor", 1.3, 2))
        for _ in range(total):
            ref.update()
        bad = [k for k, e in ref.fields().items() if not np.array_equal(out[k], e, equal_nan=True)]
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{len(bad)} {total} {period} {dev.n_exchanges / total:.2f} {sorted(set(h for h, _ in geo))} {sorted(set(n for _, n in geo))}\n")
    dist.barrier()
    dist.destroy_process_group()
