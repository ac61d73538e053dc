#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/sq_counters.sh <tag>'): per-kernel SQ wave-cycle split of the headline workload -
# parked on s_waitcnt (memory) / issue-stalled / issuing - from one rocprofv3 PMC pass.  Output: gpurun_out/sq_<tag>/summary.txt
set -u
TAG=${1:-r3}; OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT -o p -- \
    python3 bench.py --steps 6 --warmup 6 --no-cpu --no-graph --sweeps 10 > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:64]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k] += r['Counter_Name'] == 'SQ_WAVES'
print("fractions of SQ_WAVE_CYCLES (quad-cycles a wave is resident): wait_any = parked on s_waitcnt / barrier, wait_inst = ready but not issued, active = issuing")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:16]:
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print(f"{k:66s} launches={n[k]:4d} wait_any={d.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst={d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
          f"active={d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu={d.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} quad-cycles/wave={wc/max(d.get('SQ_WAVES',1),1):8.0f}")
PY
