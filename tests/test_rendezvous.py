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
