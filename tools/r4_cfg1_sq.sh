#!/bin/bash
# SQ wave-cycle split + L2 / fabric traffic of configs[1]'s kernels (ON THE GPU BOX)
set -u
OUT=gpurun_out/cfg1_sq; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--bc 2 --res 1600 --jacobi 50 --steps 12 --warmup 6 --no-cpu --no-graph --sweeps 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 bench.py $ARGS > $OUT/bench$i.json 2> $OUT/err$i.txt
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:56]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:7]:
    print(k)
    print("   " + "  ".join(f"{c}={v / max(n[k][c], 1):.0f}" for c, v in sorted(d.items())))
PY
