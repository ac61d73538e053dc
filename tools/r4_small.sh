#!/bin/bash
# small-grid tile-height sweep (ON THE GPU BOX): steps/s of the reference's default resolution for each tile-height switch
one() { echo -n "$1 | $2: "; env $1 python3 bench.py $2 --steps 3000 --warmup 100 --sweeps 0 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
A="--bc 2 --res 400"
for e in X=1 FS_RBPAIR_RT=6 FS_K34_RT=4 FS_K34_N=4 FS_MAC_RT=4 FS_PAIR_RT=1 FS_PAIR_RT=4 FS_STACK=0 FS_TILE_LIST=0 X=2; do one $e "$A"; done
B="--bc 1 --res 200 --scheme upwind --vc 0 --re 1000 --dt 0.0005"
for e in X=1 FS_RBPAIR_RT=6 FS_MAC_RT=4 FS_STACK=0 FS_TILE_LIST=0 X=2; do one $e "$B"; done
