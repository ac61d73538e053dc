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
