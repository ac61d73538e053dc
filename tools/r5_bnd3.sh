#!/bin/bash
# (Historical: FS_FUSE_K2 numbering of the time - 2 = two launches (today's 1), 3 = one launch (today's 2, the default).)
# Run ON THE GPU BOX: fs_cip_step as two launches (FS_FUSE_K2=2) against one launch over both kinds of tile (3), interleaved; the committed build beside them
set -u
mkdir -p gpurun_out
A=$PWD/tools/ab/lib_base.so
for cfg in "--steps 120 --warmup 20 --no-cpu --sweeps 0" "--res 1600 --bc 2 --steps 400 --warmup 40 --no-cpu --sweeps 0" "--res 2048 --bc 5 --steps 300 --warmup 40 --no-cpu --sweeps 0" "--res 8192 --bc 2 --steps 40 --warmup 10 --no-cpu --sweeps 0"; do
  echo "== $cfg"
  BENCH_ARGS="$cfg" bash tools/r3_ab.sh bnd3 "A1:FS_LIB=$A" "B1:FS_FUSE_K2=2" "C1:FS_FUSE_K2=3" "A2:FS_LIB=$A" "B2:FS_FUSE_K2=2" "C2:FS_FUSE_K2=3" "A3:FS_LIB=$A" "B3:FS_FUSE_K2=2" "C3:FS_FUSE_K2=3" | cut -c1-200
  python3 - <<'P'
import json
for n in ("A1","B1","C1"):
    d=json.load(open(f"gpurun_out/ab_bnd3/{n}.json")); print(n, d["kernels"]["cip_step"].get("parts_us"), d["state_checksum"])
P
done 2>&1 | tee gpurun_out/r5_bnd3.txt
