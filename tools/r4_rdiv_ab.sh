#!/bin/bash
# Run ON THE GPU BOX: A/B of the guarded Reynolds-number division (fs_device.h FS_RDIV_FIX = 0 unguarded / 1 Newton step / 2 rare branch)
set -u
L=$PWD/tools/ab
for cfg in "--steps 120 --warmup 20 --no-cpu --sweeps 0" "--res 1600 --bc 2 --steps 300 --warmup 20 --no-cpu --sweeps 0" "--res 400 --bc 2 --steps 2000 --warmup 50 --no-cpu --sweeps 0" "--res 4096 --bc 3 --scheme kk --vc 10 --re 1e8 --steps 120 --warmup 20 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh rdiv "fix0:FS_LIB=$L/libfs_rdiv0.so" "fix1:FS_LIB=$L/libfs_rdiv1.so" "fix2:FS_LIB=$L/libfs_rdiv2.so" "fix0b:FS_LIB=$L/libfs_rdiv0.so" "fix2b:FS_LIB=$L/libfs_rdiv2.so"
done
