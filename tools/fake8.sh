#!/bin/bash
# bench.py as 8 socket-connected ranks on ONE GPU at the headline size (see tests/bench_socket_worker.py); prints rank 0's JSON head
PORT=${1:-31000}
ARGS="--gpus 8 --steps 10 --warmup 4 --sweeps 20 --no-cpu"
TMO=${2:-600}      # every rank under a timeout: a hang must not eat the GPU budget
for r in 1 2 3 4 5 6 7; do
  RANK=$r WORLD_SIZE=8 LOCAL_RANK=0 FS_FAKE_PORT=$PORT MASTER_PORT=$((PORT+100)) timeout $TMO python tests/bench_socket_worker.py $ARGS > gpurun_out/fake8_rank$r.out 2> gpurun_out/fake8_rank$r.err &
done
RANK=0 WORLD_SIZE=8 LOCAL_RANK=0 FS_FAKE_PORT=$PORT MASTER_PORT=$((PORT+100)) timeout $TMO python tests/bench_socket_worker.py $ARGS > gpurun_out/fake8_rank0.out 2> gpurun_out/fake8_rank0.err
wait
python bench.py --steps 10 --warmup 4 --sweeps 20 --no-cpu > gpurun_out/fake8_single.out 2> gpurun_out/fake8_single.err
python - <<'PY'
import json
a=json.loads(open('gpurun_out/fake8_rank0.out').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/fake8_single.out').read().strip().splitlines()[-1])
print('8 ranks :', a['n_gpus'], a['state_checksum'], a['poisson_residual'], a['halo_exchanges_per_step'])
print('1 rank  :', b['n_gpus'], b['state_checksum'], b['poisson_residual'])
print('EQUAL' if a['state_checksum']==b['state_checksum'] else 'MISMATCH')
PY
