"""File rendezvous used to bootstrap the RCCL communicator (fs/rendezvous.py): rank 0's bytes reach every rank."""
import multiprocessing as mp
import os


def _rank(rank, world, key, d, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "2d-fluid-simulator_amd"))
    os.environ["FS_RDZV_DIR"] = d
    from fs.rendezvous import FileRendezvous
    r = FileRendezvous(rank, world, key=key, timeout=30)
    a = r.bcast(b"A" * 128 if rank == 0 else None)
    b = r.bcast(b"second" if rank == 0 else None)
    q.put((rank, a, b))


def test_bcast_reaches_all_ranks(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 3, "t1", str(tmp_path), q)) for r in (2, 1, 0)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-fluid-simulator_amd"))
    os.environ["FS_RDZV_DIR"] = str(tmp_path)
    from fs.rendezvous import FileRendezvous
    r0 = FileRendezvous(0, 3, key="t1")
    r0.calls = 2
    r0.cleanup()            # rank 0 removes the files only once every rank is through (bench.py: after the last barrier)
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[1] == b"A" * 128 and g[2] == b"second" for g in got)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("fs_rdzv_t1")]


def test_stale_file_of_an_earlier_job_is_ignored(tmp_path, monkeypatch):
    """a file left behind by a crashed job with the same key must not be taken for this job's payload."""
    import sys
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-fluid-simulator_amd"))
    monkeypatch.setenv("FS_RDZV_DIR", str(tmp_path))
    from fs.rendezvous import FileRendezvous
    stale = tmp_path / "fs_rdzv_k_0"
    stale.write_bytes(b"old id")
    old = time.time() - 3 * 86400.0          # older than any launcher that could still be alive in this container
    os.utime(stale, (old, old))
    r1 = FileRendezvous(1, 2, key="k", timeout=0.3)
    try:
        r1.bcast(None)
        raise AssertionError("stale payload accepted")
    except TimeoutError:
        pass
    r0 = FileRendezvous(0, 2, key="k")
    r0.bcast(b"new id")
    r1 = FileRendezvous(1, 2, key="k", timeout=5)
    assert r1.bcast(None) == b"new id"


def test_default_directory_is_private(monkeypatch):
    import stat
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-fluid-simulator_amd"))
    monkeypatch.delenv("FS_RDZV_DIR", raising=False)
    from fs import rendezvous
    d = rendezvous._private_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o077)


def test_job_nonce_separates_jobs_of_one_parent(tmp_path, monkeypatch):
    """two jobs started by ONE long-lived parent on ONE MASTER_PORT.  A fresh file of the first job (younger than the
    parent, so the age test cannot reject it) must not reach the second job's ranks: FS_RDZV_NONCE is part of the key."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-fluid-simulator_amd"))
    monkeypatch.setenv("FS_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29999")
    from fs.rendezvous import FileRendezvous
    monkeypatch.setenv("FS_RDZV_NONCE", "job-1")
    a0 = FileRendezvous(0, 2)
    a0.bcast(b"id of job 1")                      # job 1 dies here: its file stays
    monkeypatch.setenv("FS_RDZV_NONCE", "job-2")
    b1 = FileRendezvous(1, 2, timeout=0.3)        # job 2's rank 1 is up before its rank 0
    try:
        b1.bcast(None)
        raise AssertionError("rank 1 of job 2 read job 1's payload")
    except TimeoutError:
        pass
    b0 = FileRendezvous(0, 2)
    b0.bcast(b"id of job 2")
    b1 = FileRendezvous(1, 2, timeout=5)
    assert b1.bcast(None) == b"id of job 2"
    assert a0.base != b0.base


def _bench_module():
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(repo, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_WORKER = """
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["FS_RDZV_NONCE"]
assert os.environ["LOCAL_RANK"] == (os.environ.get("FS_BENCH_LOCAL_RANK") or str(r))
mode = os.environ.get("T_MODE", "ok")
if mode == "die" and r == 1:
    sys.exit(3)
if mode in ("die", "hang") and r != 1:
    time.sleep(600)
print(f"rank {r} of {w} args {sys.argv[1:]}" if r else '{"n_gpus": %d}' % w, flush=True)
"""


def test_bench_starts_its_own_ranks(tmp_path, monkeypatch, capfd):
    """`python bench.py --gpus N` with WORLD_SIZE unset starts N child ranks, relays rank 0's single line and returns the
    worst status; a rank that dies, or a job that hangs, ends with the others killed and a non-zero status (no GPU involved here:
    FS_BENCH_WORKER swaps the rank program for a stub)."""
    import sys
    import time
    bench = _bench_module()
    worker = tmp_path / "worker.py"
    worker.write_text(_WORKER)
    monkeypatch.setenv("FS_BENCH_WORKER", str(worker))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "7"])
    assert bench.spawn_ranks(3) == 0
    out = capfd.readouterr()
    assert out.out.strip() == '{"n_gpus": 3}'                       # rank 0's line, and nothing else, on stdout
    assert "rank 1 of 3 args ['--gpus', '3', '--steps', '7']" in out.err and "rank 2 of 3" in out.err
    monkeypatch.setenv("T_MODE", "die")
    t0 = time.time()
    rc = bench.spawn_ranks(3)
    assert rc == 3 and time.time() - t0 < 60                        # the sleeping ranks were killed after the grace period
    monkeypatch.setenv("T_MODE", "hang")
    monkeypatch.setenv("FS_BENCH_TIMEOUT", "2")
    t0 = time.time()
    assert bench.spawn_ranks(2) == 124 and time.time() - t0 < 30


def _preflight_rank(rank, world, key, ok, q):
    from fs.rendezvous import FileRendezvous
    r = FileRendezvous(rank, world, key=key, timeout=30)
    q.put((rank, r.preflight(ok, "" if ok else f"rank {rank}: LOCAL_RANK {rank} but 1 GPU(s) visible", timeout=20)))
    # (no cleanup here: a rank's status file must outlive the slowest reader - bench.py removes it at the end of a run, after the barriers)


def test_preflight_every_rank_learns_every_verdict(tmp_path, monkeypatch):
    """bench.py --gpus N, before ncclCommInitRank: each rank reports whether its GPU exists and librccl loads; every rank gets the
    list of failures (empty = go) - nobody is left waiting for a peer that already gave up.  A rank that never reports is a failure too."""
    import multiprocessing as mp
    monkeypatch.setenv("FS_RDZV_DIR", str(tmp_path))
    os.chmod(tmp_path, 0o700)
    ctx = mp.get_context("fork")
    for verdicts in ([True, True, True], [True, False, True]):
        q = ctx.Queue()
        procs = [ctx.Process(target=_preflight_rank, args=(r, 3, f"pf{sum(verdicts)}", verdicts[r], q)) for r in range(3)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=60) for _ in procs)
        for p in procs:
            p.join(timeout=30)
        expect = [] if all(verdicts) else ["rank 1: LOCAL_RANK 1 but 1 GPU(s) visible"]
        assert all(got[r] == expect for r in range(3)), got
    from fs.rendezvous import FileRendezvous
    lone = FileRendezvous(0, 2, key="pf_lone", timeout=30)
    failures = lone.preflight(True, timeout=0.5)
    assert len(failures) == 1 and failures[0].startswith("rank 1 did not report"), failures
    lone.cleanup()


def test_bench_parent_refuses_more_ranks_than_gpus(monkeypatch, capfd):
    """`python bench.py --gpus N` on a node with fewer GPUs: rc != 0 and ONE stderr line, no rank started (the GPUs are counted from the
    KFD topology - the parent never makes a HIP call)."""
    import sys
    bench = _bench_module()
    monkeypatch.delenv("FS_BENCH_LOCAL_RANK", raising=False)
    monkeypatch.setattr(bench, "visible_gpus_without_hip", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    assert bench.spawn_ranks(8) == 2
    err = capfd.readouterr().err.strip().splitlines()
    assert len(err) == 1 and "--gpus 8 but 1 GPU(s) visible" in err[0]
    assert bench.visible_gpus_without_hip.__name__ == "<lambda>"
    n = _bench_module().visible_gpus_without_hip()       # the real counter: None without a KFD topology (this container), else a count >= 1
    assert n is None or n >= 1


def test_preflight_files_live_next_to_the_rendezvous_files(monkeypatch):
    """Round 6, first GPU lease: the status file's name was derived by replacing 'fs_rdzv_' in the whole PATH - with the default directory
    /tmp/fs_rdzv_<uid> that renamed the directory.  Default directory, one rank."""
    from fs.rendezvous import FileRendezvous
    monkeypatch.delenv("FS_RDZV_DIR", raising=False)
    r = FileRendezvous(0, 1, key="pf_default_dir")
    assert r.preflight(True, timeout=5) == []
    assert os.path.dirname(r._pre) == os.path.dirname(r.base) and os.path.basename(r._pre).startswith("fs_pre_") and os.path.exists(r._pre)
    r.cleanup()
    assert not os.path.exists(r._pre)
