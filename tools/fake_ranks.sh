#!/bin/bash
# bench.py as N socket-connected ranks on ONE GPU (tests/bench_socket_worker.py): tools/fake_ranks.sh N PORT TIMEOUT [bench args ...]; every
# rank runs under `timeout -s ABRT` with the fault handler on, so a hang ends with each rank's Python traceback in gpurun_out/fake_rank<r>.err
N=$1; PORT=$2; TMO=$3; shift 3
mkdir -p gpurun_out
for r in $(seq 1 $((N-1))); do
  PYTHONFAULTHANDLER=1 RANK=$r WORLD_SIZE=$N LOCAL_RANK=0 FS_FAKE_PORT=$PORT MASTER_PORT=$((PORT+100)) timeout -s ABRT $TMO python tests/bench_socket_worker.py --gpus $N "$@" > gpurun_out/fake_rank$r.out 2> gpurun_out/fake_rank$r.err &
done
PYTHONFAULTHANDLER=1 RANK=0 WORLD_SIZE=$N LOCAL_RANK=0 FS_FAKE_PORT=$PORT MASTER_PORT=$((PORT+100)) timeout -s ABRT $TMO python tests/bench_socket_worker.py --gpus $N "$@" > gpurun_out/fake_rank0.out 2> gpurun_out/fake_rank0.err
echo "rank 0 rc=$?"
wait
head -c 600 gpurun_out/fake_rank0.out; echo
for r in 0 1 $((N-1)); do echo "== rank $r stderr"; grep -v "^$" gpurun_out/fake_rank$r.err | tail -25; done
