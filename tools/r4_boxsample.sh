#!/bin/bash
# ON THE GPU BOX: one line per call - the box's probes (copy, VALU, mixed) and the headline kernels' times of the same process; collected over a round
# to see which probe, if any, predicts the 5-9 % between boxes.  Appends to gpurun_out/boxsamples.txt.
mkdir -p gpurun_out
python3 bench.py --steps 120 --warmup 24 --sweeps 0 --no-cpu 2>/dev/null | python3 -c "
import json,sys,time
d=json.loads(sys.stdin.read()); b=d['box']; k=d['kernels']
print(time.strftime('%H:%M:%S'), 'steps/s', d['value'], 'copy', b['copy_GBps'], 'valu', b.get('valu_ginstr_per_simd'), 'mixed', b.get('mixed_GBps'), 'K3+K4', k['cip_grad_advect_rt']['avg_us'], 'pair', k['rbsor_pair']['avg_us'], 'K2', k['cip_nonadv']['avg_us'], 'VC', k['vort_confine']['avg_us'])" | tee -a gpurun_out/boxsamples.txt
