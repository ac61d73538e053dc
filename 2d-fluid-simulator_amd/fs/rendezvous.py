"""Out-of-band bootstrap for the RCCL communicator of a single-node, one-process-per-GPU job.

Only ONE thing has to travel outside RCCL: the 128-byte ncclUniqueId from rank 0 to the others.  The ranks
of a `torch.distributed.run` (or any) single-node launch share a parent process and MASTER_PORT, which key a
file in /tmp: rank 0 writes it atomically, the others poll.  Everything after that (barriers, max-over-ranks
of the timings) goes through the communicator itself (Device.barrier / Device.allgather_scalars), so the
benchmark process needs no torch / MPI import at all.
"""
import os
import time


class FileRendezvous:
    def __init__(self, rank, world, key=None, timeout=300.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if key is None:
            key = f"{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self.base = os.path.join(os.environ.get("FS_RDZV_DIR", "/tmp"), f"fs_rdzv_{key}")
        self.calls = 0
        self.t_start = time.time()

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
            try:    # a leftover file of an earlier job with a recycled key is older than this process: ignore it
                if os.path.getmtime(path) >= self.t_start - 120.0:
                    break
            except OSError:
                pass
            if time.time() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous file {path} did not appear within {self.timeout} s")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    def cleanup(self):
        if self.rank == 0:
            for k in range(self.calls):
                try:
                    os.remove(f"{self.base}_{k}")
                except OSError:
                    pass
