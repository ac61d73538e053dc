#!/bin/bash
# Run ON THE GPU BOX: BASELINE configs[1] (bc2 res 1600, Jacobi 50) with the register-tile passes and the marching passes
for s in 0 4 6 8; do
  for extra in "" "FS_JM_PF=3"; do
    [ "$s" = 0 ] && [ -n "$extra" ] && continue
    [ "$s" = 8 ] && [ -n "$extra" ] && continue
    echo -n "FS_JACOBI_MARCH=$s $extra: "
    env FS_JACOBI_MARCH=$s $extra python3 bench.py --res 1600 --bc 2 --jacobi 50 --steps 200 --warmup 20 --no-cpu --sweeps 0 2>gpurun_out/jm_$s.err | python3 tools/benchline.py
  done
done
for L in 4 16 28 40; do echo -n "S=4 L=$L: "; FS_JM_L=$L FS_JACOBI_MARCH=4 python3 bench.py --res 1600 --bc 2 --jacobi 50 --steps 200 --warmup 20 --no-cpu --sweeps 0 2>/dev/null | python3 tools/benchline.py; done
for L in 12 24 36; do echo -n "S=6 L=$L: "; FS_JM_L=$L FS_JACOBI_MARCH=6 python3 bench.py --res 1600 --bc 2 --jacobi 50 --steps 200 --warmup 20 --no-cpu --sweeps 0 2>/dev/null | python3 tools/benchline.py; done
