#!/bin/bash
# Run ON THE GPU BOX: differential fuzzing of the final round-6 code (HIP library vs CPU oracle, bit for bit) -> gpurun_out/r6_fuzz.txt
# default launch forms; the large-grid forms forced onto the small fuzz grids (FS_RBPAIR_SPLIT=2: the one-launch red-black pair, the Jacobi two-part
# passes; + FS_SMALL_CELLS=0: 4-row tiles everywhere); grids up to 2.5 M cells; slab cases (2-8 contexts on one GPU) incl. halo 20 with the pair pass
set -u
OUT=gpurun_out/r6_fuzz.txt; : > $OUT
run() { label=$1; shift; echo "== $label" >> $OUT; env "$@" 2>&1 | tail -2 >> $OUT; }
run "default forms, 3000 cases"                              FS_X=0 python3 tools/fuzz_parity.py --cases 3000 --seed 6100000
run "FS_RBPAIR_SPLIT=2, 3000 cases"                          FS_RBPAIR_SPLIT=2 python3 tools/fuzz_parity.py --cases 3000 --seed 6200000
run "FS_RBPAIR_SPLIT=2 FS_SMALL_CELLS=0, 3000 cases"         FS_RBPAIR_SPLIT=2 FS_SMALL_CELLS=0 python3 tools/fuzz_parity.py --cases 3000 --seed 6300000
run "FS_RBPAIR_SPLIT=2 FS_SMALL_CELLS=0 FUZZ_FUSE_P=1, 2000" FS_RBPAIR_SPLIT=2 FS_SMALL_CELLS=0 FUZZ_FUSE_P=1 python3 tools/fuzz_parity.py --cases 2000 --seed 6400000
run "grids to 2.5 M cells, 150 cases"                        FS_X=0 python3 tools/fuzz_parity.py --cases 150 --seed 6500000 --max-cells 2500000
run "slabs, 500 cases"                                       FS_X=0 python3 tools/fuzz_slabs.py --cases 500 --seed 6600000
run "slabs FS_RBPAIR_SPLIT=2 halo 20, 300 cases"             FS_RBPAIR_SPLIT=2 FORCE_HALO=20 python3 tools/fuzz_slabs.py --cases 300 --seed 6700000
cat $OUT
