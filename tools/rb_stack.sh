for s in 0 1 2 4 0 2; do FS_RB_STACK=$s python bench.py --no-cpu --steps 100 --warmup 20 --sweeps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stack',$s, d['value'], d['kernels']['rbsor_iteration']['avg_us'], d['state_checksum']['p'])"; done
