#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of environment settings on one bench configuration: tools/r6_env_ab.sh "<bench args>" <rounds> "name:VAR=val VAR2=val" ...
set -u
ARGS=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python3 bench.py $ARGS --no-cpu --sweeps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'round $r', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  done
done
