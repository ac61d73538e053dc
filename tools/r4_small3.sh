#!/bin/bash
# where the small-grid tile heights stop paying (ON THE GPU BOX): steps/s at several resolutions with the threshold below / above the grid
one() { echo -n "$1 | $2: "; env $1 python3 bench.py $2 --sweeps 0 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for res in 100 256 600 800 1024; do
  st=$((1200000 / res)); 
  for e in FS_SMALL_CELLS=0 FS_SMALL_CELLS=100000000; do one $e "--bc 2 --res $res --steps $st --warmup 60"; done
done
