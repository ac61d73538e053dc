"""Worker for tests/test_slab_gloo.py: one rank of a world_size-N gloo job running the product's host logic
(fs.runtime.DeviceBase + solvers) on the CPU stand-in device."""
import os
import sys

import numpy as np


def run(rank, world, port, fname, halo, out_dir):
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
    fs.runtime.init(dtype="f64" if cfg["fp64"] else "f32", rank=rank, nranks=world, halo=halo, allgather=allgather,
                    device_cls=OracleSlabDevice)
    sim = make_product(g, cfg)
    dev = sim._solver._bc.device
    bad = []
    last = max(cfg["snaps"])
    for step in range(1, last + 1):
        sim.step()
        if step in cfg["snaps"]:
            for k, a in sim.field_to_numpy().items():
                if not np.array_equal(a, g[f"step{step}.{k}"]):
                    bad.append(f"step{step}.{k}")
    # internal buffers too (stale-cell choreography across slabs)
    s = sim._solver
    for name in ("v", "p", "vx", "vy", "dye", "dyex", "dyey"):
        if f"final.{name}.current" in g:
            for which in ("current", "next"):
                if f"{name}.{which}" in dead_buffers(s):
                    continue
                if not np.array_equal(getattr(getattr(s, name), which).to_numpy(), g[f"final.{name}.{which}"]):
                    bad.append(f"final.{name}.{which}")
    if rank == 0:
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{len(bad)} {dev.n_exchanges / last:.2f} {' '.join(bad)}\n")
    dist.barrier()
    dist.destroy_process_group()
