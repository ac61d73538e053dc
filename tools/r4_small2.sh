#!/bin/bash
# steps/s of small / mid / headline grids under one or more env settings (ON THE GPU BOX):  bash tools/r4_small2.sh "X=1" "FS_RBPAIR_RT=2" ...
one() { echo -n "$1 | $2: "; env $1 python3 bench.py $2 --sweeps 0 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], {n: k[n]['avg_us'] for n in k if 'pair' in n or 'bc' in n})"; }
for e in "$@"; do
  one $e "--bc 1 --res 200 --scheme upwind --vc 0 --re 1000 --dt 0.0005 --steps 3000 --warmup 100"
  one $e "--bc 2 --res 400 --steps 3000 --warmup 100"
  one $e "--bc 2 --res 400 --dye --steps 3000 --warmup 100"
  one $e "--bc 2 --res 1600 --steps 600 --warmup 60"
  one $e "--steps 120 --warmup 24"
done
