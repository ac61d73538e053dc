for g in 4 8 16 32 64; do for st in 21 53; do
  FS_XCD_GROUP=$g FS_STACK=$st python tools/kbench.py --steps 2 --warm 30 --sweeps 100 2>&1 | grep -E "jacobi" | sed "s/^/XG=$g STACK=$st /"
done; done
FS_JACOBI=21 python tools/kbench.py --steps 2 --warm 30 --sweeps 100 2>&1 | grep -E "jacobi_sweep " | sed "s/^/RT1 /"
FS_JACOBI=21 FS_XCD_GROUP=16 python tools/kbench.py --steps 2 --warm 30 --sweeps 100 2>&1 | grep -E "jacobi_sweep " | sed "s/^/RT1 XG16 /"
