"""bench.py end to end as 2 and 3 ranks on ONE GPU (ghost rows and scalar collectives over local sockets instead of RCCL,
tests/bench_socket_worker.py): the N > 1 control flow of the benchmark runs here before it runs on the driver's multi-GPU
node, and its `state_checksum` must equal the single-rank run's - the same comparison the --gpus 1/2/4/8 lines allow."""
import json
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--res", "256", "--steps", "10", "--warmup", "4", "--sweeps", "20", "--no-cpu", "--trial-steps", "24"]      # (the trial's default of 120 steps per mode is for real links)


def _single():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + ARGS, capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _multi(world, extra_env=None):
    port = random.randint(20000, 50000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", FS_FAKE_PORT=str(port),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 100), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "bench_socket_worker.py"), "--gpus", str(world)] + ARGS,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {se[-2000:]}"
    for r in range(1, world):
        assert outs[r][0].strip() == "", f"rank {r} printed to stdout"
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0][0]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single(hip_lib):
    return _single()


@pytest.mark.parametrize("world,halo,overlap", [(2, None, None), (3, "4", None), (4, "16", None), (2, None, "1")])
def test_bench_as_n_ranks_matches_single_rank(world, halo, overlap, single):
    """(overlap "1": FS_OVERLAP=1 - the exchanges on the communication stream without a trial; otherwise the run's own timing picks the mode, and
    whichever it picks must satisfy every assertion below)"""
    env = {"FS_HALO": halo} if halo else {}
    if overlap:
        env["FS_OVERLAP"] = overlap
    d = _multi(world, env or None)
    assert d["n_gpus"] == world and d["steps"] == 10 and d["warmup"] == 4 and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == f"y-slab x{world}"
    assert d["state_checksum"] == single["state_checksum"]
    assert d["poisson_residual"]["cells"] == single["poisson_residual"]["cells"]
    assert abs(d["poisson_residual"]["rms"] - single["poisson_residual"]["rms"]) <= 1e-12 * max(1.0, single["poisson_residual"]["rms"])
    assert d["halo_exchanges_per_step"]["grouped_launches"] > 0
    assert d["roofline"]["frac"] > 0 and d["poisson_jacobi_sweep"]["frac"] > 0 and d["roofline"]["kernel"] in d["kernels"]
    assert "cpu_baseline" not in d and single["value"] > 0
    # every launch list was built during the warm-up: none inside the timed region, and no launch of the taped period fell back to its dense grid
    assert d["launch_lists"]["built_in_timed_region"] == 0 and d["launch_lists"]["dense_fallbacks"] == 0, d["launch_lists"]
    assert single["launch_lists"]["built_in_timed_region"] == 0 and single["launch_lists"]["dense_fallbacks"] == 0, single["launch_lists"]
    # the run timed its period with the exchanges in line and on the communication stream and kept one of the two (same bits: the checksum above)
    tr = d["exchange_mode_trial"]
    if overlap:
        assert tr is None and d["halo_exchanges_per_step"]["overlapped_fraction"] >= 0
    else:
        assert tr and tr["in_line_us_per_step"] > 0 and tr["overlapped_us_per_step"] > 0 and tr["chosen"].split()[0] in ("in", "overlapped"), tr
    assert single["exchange_mode_trial"] is None


def test_bench_gpus_2_as_one_command(single):
    """`python bench.py --gpus 2 ...` with no launcher around it.  The parent never touches the GPU: it starts the two
    ranks itself (here the socket stand-in as the rank program, both pinned to GPU 0), relays rank 0's single JSON line and exits 0."""
    port = random.randint(20000, 50000)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(FS_BENCH_WORKER=os.path.join(REPO, "tests", "bench_socket_worker.py"), FS_BENCH_LOCAL_RANK="0", FS_FAKE_PORT=str(port),
               FS_BENCH_TIMEOUT="600")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "y-slab x2"
    assert d["state_checksum"] == single["state_checksum"]
    assert d["box"]["copy_GBps"] > 0 and d["roofline"]["frac_of_box_copy"] > 0


def test_bench_line_contract_with_the_large_grid_step(hip_lib):
    """One rank, the large-grid launch forms forced onto a small grid: the line prices the kernel it was ASKED to price (--roofline-kernel: by name,
    never "whichever launch was slowest in this run" - on a grid this small two 10 us kernels trade places from box to box), names the __global__
    functions the library launched (fs_prof_kernels), times the graded Jacobi sweep as one event span, and carries the CPU oracle's in-run
    parity incl. the graded sweep's values.  Structure and membership only: no assertion here compares two measured times."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["FS_RBPAIR_SPLIT"] = "2"
    env["FS_SMALL_CELLS"] = "0"          # (below 2 M cells the pair and fs_cip_step run 2-row tiles in one launch: the big grids' tile heights here)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--res", "512", "--bc", "2", "--steps", "12", "--warmup", "4", "--sweeps", "20",
                          "--cpu-seconds", "2", "--roofline-kernel", "cip_step"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["kernel"] == "cip_step" and rf["selection"] == "--roofline-kernel cip_step"
    assert rf["bound"] == "hbm" and rf["frac"] > 0 and rf["peak"] == 8000.0 and rf["alg_bytes_per_launch"] > 0
    assert rf["traffic"] is None                                      # (PMC numbers belong to the headline workload and to one build of the library)
    assert len(rf["gpu_kernels"]) == 1 and rf["gpu_kernels"][0].startswith("fs::k_cip_step_all<"), rf["gpu_kernels"]
    cs = d["kernels"]["cip_step"]
    assert "parts_us" not in cs                                        # one launch over both kinds of tile
    bk = rf["by_kind_of_tile"]      # diagnostic on a second context: one launch per kind of tile, the all-fluid body on the bytes of its own tiles
    assert bk["all_fluid_tiles"] > 0 and bk["all_fluid_us"] > 0 and bk["all_fluid_frac"] > 0, bk
    pair = d["kernels"]["rbsor_pair"]      # FS_RBPAIR_SPLIT=2: the large grids' form - ONE launch over the all-fluid (stacked) and the other tiles
    assert "parts_us" not in pair and len(pair["gpu_kernels"]) == 1 and pair["gpu_kernels"][0].startswith("fs::k_rbsor_pair_all<"), pair
    jac = d["poisson_jacobi_sweep"]
    assert jac["timing"].startswith("one HIP-event pair") and jac["per_launch_avg_us"] > 0 and jac["frac"] > 0
    assert jac["gpu_kernels"] == [k for k in jac["gpu_kernels"] if k.startswith("fs::k_jacobi_ov2<")] and len(jac["gpu_kernels"]) == 1, jac["gpu_kernels"]
    assert d["box"]["valu_pk_ginstr_per_simd"] > 0 and d["box"]["valu_ginstr_per_simd"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["parity_in_run"]["bit_identical"] is True
    assert cb["parity_in_run"]["graded_sweep_vs_oracle"] == {"jacobi_sweep": True, "jacobi_sweep_src": True}


def test_roofline_kernel_falls_back_by_name_when_the_workload_never_launches_it(hip_lib):
    """An upwind run (BASELINE configs[0]'s scheme) has no cip_step: the line says so and prices a kernel the run did launch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--res", "200", "--bc", "1", "--scheme", "upwind", "--vc", "0", "--re", "1000",
                          "--dt", "0.0005", "--steps", "10", "--warmup", "2", "--sweeps", "0", "--no-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    rf = d["roofline"]
    assert rf["kernel"] in d["kernels"] and rf["kernel"] != "cip_step" and "not launched by this workload" in rf["selection"]
    assert rf["gpu_kernels"] and all(k.startswith("fs::k_") for k in rf["gpu_kernels"])
