export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 6 --warmup 6 --no-cpu --no-graph --sweeps 10 > $OUT/bench.json 2> $OUT/err.txt
find $OUT -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:14]:
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print(f"{k:62s} n={cnt[k]:4d} wait_any={d.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst={d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active={d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu={d.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} cyc/wave={wc/max(d.get('SQ_WAVES',1),1):9.0f}")
PY
