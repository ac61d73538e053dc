#!/bin/bash
# Run ON THE GPU BOX: parity of the row-streaming Jacobi variants, then the isolated-sweep timings per variant (tools/kbench.py).
OUT=gpurun_out/${1:-jac}; mkdir -p $OUT
for v in 632 416; do
  FS_JACOBI=$v timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_traj.py "tests/test_gpu_fullsize.py::test_fast_paths_equal_reference_literal_kernels" -m gpu -x -q -k "jacobi or cfg2" 2>&1 | tail -3 > $OUT/parity_$v.log
  echo "parity FS_JACOBI=$v: $(tail -1 $OUT/parity_$v.log)"
done
for v in 0 24 416 432 532 616 632 664 832 864; do
  FS_JACOBI=$v python tools/kbench.py --steps 4 --warm 30 --sweeps 200 2>&1 | grep -E "jacobi" | sed "s/^/FS_JACOBI=$v /"
done | tee $OUT/variants.log
