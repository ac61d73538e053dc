for v in 0 23 24; do
  FS_JACOBI=$v python tools/kbench.py --steps 2 --warm 30 --sweeps 100 2>&1 | grep -E "jacobi" | sed "s/^/FS_JACOBI=$v /"
done
FS_JACOBI=23 FS_RCP=33 python tools/kbench.py --steps 2 --warm 30 --sweeps 100 2>&1 | grep -E "jacobi_sweep " | sed "s/^/RT3+rcp /"
FS_JACOBI=23 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_traj.py -m gpu -x -q -k "jacobi" 2>&1 | tail -1
