for g in 4 8 16 32; do FS_XCD_GROUP=$g python bench.py --no-cpu --steps 100 --warmup 20 --sweeps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('group',$g, d['value'], ' '.join(f'{n}={k[n][\"avg_us\"]}' for n in ('cip_nonadv','cip_nonadv_grad','cip_advect','vort_confine','rbsor_iteration')))"; done
