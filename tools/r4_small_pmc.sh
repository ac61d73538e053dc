#!/bin/bash
# SQ / instruction-cache counters of a small grid's kernels (ON THE GPU BOX), one rocprofv3 PMC pass per counter group
set -u
OUT=gpurun_out/small_pmc; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--bc 1 --res 200 --scheme upwind --vc 0 --re 1000 --dt 0.0005 --steps 40 --warmup 10 --no-cpu --no-graph --sweeps 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD"; do
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
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    print(k)
    print("   " + "  ".join(f"{c}={v / max(n[k][c], 1):.0f}" for c, v in sorted(d.items())))
PY
