"""Out-of-band bootstrap for the RCCL communicator of a single-node, one-process-per-GPU job.

Only ONE thing has to travel outside RCCL: the 128-byte ncclUniqueId from rank 0 to the others.  The ranks
of a `torch.distributed.run` (or any) single-node launch share a parent process and MASTER_PORT, which key a
file in /tmp: rank 0 writes it atomically, the others poll.  A launcher that knows better exports FS_RDZV_NONCE, one
value per job (`bench.py --gpus N` does when it starts its own ranks): it becomes part of the key, and two jobs of one
long-lived parent on one port can no longer see each other's files.  Everything after that (barriers, max-over-ranks
of the timings) goes through the communicator itself (Device.barrier / Device.allgather_scalars), so the
benchmark process needs no torch / MPI import at all.
"""
import os
import stat
import time


def _private_dir():
    """A directory only this user can write: $FS_RDZV_DIR, else /tmp/fs_rdzv_<uid> (mode 0700, owned by us, not a symlink) -
    nobody else can pre-create or redirect the rendezvous files."""
    d = os.environ.get("FS_RDZV_DIR")
    if not d:
        d = os.path.join("/tmp", f"fs_rdzv_{os.getuid()}")
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"rendezvous directory {d} is not a private directory of this user")
    return d


def _launcher_start_time():
    """Wall-clock start of the parent process (the launcher all ranks share): a rendezvous file older than the launcher is a
    leftover of an earlier job whose key happens to repeat."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return None


class FileRendezvous:
    def __init__(self, rank, world, key=None, timeout=300.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if key is None:      # launcher identity + restart generation: a worker group restarted by the same launcher gets new files
            key = "_".join([os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"),
                            str(os.getppid()), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")])
            nonce = os.environ.get("FS_RDZV_NONCE", "")
            if nonce:        # per-job token from the launcher: nothing an earlier job left behind can carry it
                key += "_" + "".join(c for c in nonce if c.isalnum() or c in "-_")[:64]
        self.base = os.path.join(_private_dir(), f"fs_rdzv_{key}")
        self.calls = 0
        if rank == 0:
            # leftovers of a crashed earlier job whose key repeats (a long-lived parent launching several jobs on the same MASTER_PORT
            # WITHOUT an FS_RDZV_NONCE): rank 0 removes them before it writes anything.  That narrows the window, it does not close
            # it - a rank that starts before rank 0 can still read such a file if it is younger than the launcher; the ranks would
            # then wait in ncclCommInitRank until FS_COMM_TIMEOUT.  Launchers that start several jobs export FS_RDZV_NONCE.
            import glob
            for old in glob.glob(self.base + "_*"):
                try:
                    os.remove(old)
                except OSError:
                    pass
        t_launch = _launcher_start_time()
        # files written before this job's launcher existed are stale; without /proc fall back to "not much older than me"
        # (rank 0 may be well ahead of a rank whose first import is still paging in)
        self.t_valid = t_launch - 1.0 if t_launch is not None else time.time() - 120.0

    def bcast(self, payload):
        """Rank 0's bytes on every rank.  Collective: every rank must call it the same number of times."""
        path = f"{self.base}_{self.calls}"
        self.calls += 1
        if self.rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(payload)
            os.replace(tmp, path)
            return payload
        t0 = time.time()
        while True:
            try:    # a leftover file of an earlier job with a recycled key is older than this job's launcher: ignore it
                if os.path.getmtime(path) >= self.t_valid:
                    break
            except OSError:
                pass
            if time.time() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous file {path} did not appear within {self.timeout} s")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    def preflight(self, ok, message="", timeout=None):
        """Every rank reports whether it can run (its GPU is visible, librccl loads, ...) BEFORE anything blocks in ncclCommInitRank: a
        status file per rank next to the rendezvous files; every rank waits for all of them and gets the list of failure messages (empty =
        go).  A rank that never reports counts as failed after `timeout` seconds (default: the rendezvous timeout)."""
        timeout = self.timeout if timeout is None else timeout
        d, b = os.path.split(self.base)
        pre = os.path.join(d, "fs_pre_" + b[len("fs_rdzv_"):])      # (next to the rendezvous files, outside the pattern rank 0 clears at start-up)
        mine = f"{pre}_{self.rank}"
        self._pre = mine
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write("ok" if ok else (message or "failed"))
        os.replace(tmp, mine)
        failures, t0 = [], time.time()
        for r in range(self.world):
            path = f"{pre}_{r}"
            while True:
                try:
                    if os.path.getmtime(path) >= self.t_valid:
                        txt = open(path).read()
                        if txt != "ok":
                            failures.append(txt)
                        break
                except OSError:
                    pass
                if time.time() - t0 > timeout:
                    failures.append(f"rank {r} did not report within {timeout:.0f} s")
                    break
                time.sleep(0.01)
        return failures

    def cleanup(self):
        try:
            os.remove(getattr(self, "_pre", ""))
        except OSError:
            pass
        if self.rank == 0:
            for k in range(self.calls):
                try:
                    os.remove(f"{self.base}_{k}")
                except OSError:
                    pass
