#!/bin/bash
# Build a variant of libfs_hip.so for an A/B on the GPU box (bench.py / the tests take it through FS_LIB): tools/ab_build.sh <name> <TU> "<extra flags>"
# -> tools/ab/lib_<name>.so, made of the tree's objects with <TU>.hip recompiled under the extra flags (tools/ab/ is git-ignored and travels with gpurun)
set -eu
NAME=$1; TU=$2; EXTRA=$3
CS=$(dirname "$0")/../2d-fluid-simulator_amd/csrc
mkdir -p "$(dirname "$0")/ab"
make -C "$CS" -j8 >/dev/null
/opt/rocm/bin/hipcc $EXTRA -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -I/opt/rocm/include -c "$CS/$TU.hip" -o "/tmp/ab_${NAME}_$TU.o"
OBJS=""
for o in fs_core fs_transport fs_pressure fs_comm; do
    if [ "$o" = "$TU" ]; then OBJS="$OBJS /tmp/ab_${NAME}_$TU.o"; else OBJS="$OBJS $CS/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$(dirname "$0")/ab/lib_$NAME.so" $OBJS -ldl
echo "tools/ab/lib_$NAME.so"
