#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of library builds on one bench configuration, per-kernel times: tools/r6_ab.sh "<bench args>" <rounds> <name=lib.so | name=default> ...
set -u
ARGS=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = "default" ]; then unset FS_LIB; else export FS_LIB=$PWD/$lib; fi
    python3 bench.py $ARGS --no-cpu --sweeps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'round $r', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  done
done
