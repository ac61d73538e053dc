#!/bin/bash
# Run ON THE GPU BOX: tools/r3_pmc.sh <tag> [ENV=VAL ...] - three rocprofv3 PMC passes of a short bench.py run (SQ wave-cycle split +
# instruction counts; FETCH_SIZE; WRITE_SIZE + L2 hit/miss), one summary line per kernel -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
ARGS=${BENCH_ARGS:---steps 6 --warmup 6 --no-cpu --no-graph --sweeps 0}
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/fetch.err
timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/write -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/write.err
if [ -n "${EXTRA_PMC:-}" ]; then timeout 240 rocprofv3 --pmc $EXTRA_PMC --kernel-trace --output-format csv -d $OUT/extra -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/extra.err; fi
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
for fn in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"^void ", "", r['Kernel_Name']).split('(')[0].replace("fs::", "")[:56]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for fn in glob.glob(out + '/sq/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"^void ", "", r['Kernel_Name']).split('(')[0].replace("fs::", "")[:56]
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':56s} {'n':>4s} {'us(prof)':>8s} {'wait':>5s} {'stall':>5s} {'act':>5s} {'valu':>5s} {'qc/wave':>8s} {'VALU/w':>7s} {'SALU/w':>7s} {'rd MB':>7s} {'wr MB':>7s} {'L2hit':>6s} {'TB/s':>5s}")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:14]:
    wc = d.get('SQ_WAVE_CYCLES', 1); nl = max(n[(k, 'SQ_WAVES')], 1); w = max(d.get('SQ_WAVES', 1), 1)
    us = sum(dur[k]) / max(len(dur[k]), 1)
    rd = 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e6 / max(n[(k, 'FETCH_SIZE')], 1); wr = d.get('WRITE_SIZE', 0) * 1024 / 1e6 / max(n[(k, 'WRITE_SIZE')], 1)
    hit, miss = d.get('TCC_HIT_sum', 0), d.get('TCC_MISS_sum', 0)
    print(f"{k:56s} {nl:4d} {us:8.1f} {d.get('SQ_WAIT_ANY',0)/wc:5.2f} {d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} {d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} {d.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} "
          f"{wc/w:8.0f} {d.get('SQ_INSTS_VALU',0)/w:7.0f} {d.get('SQ_INSTS_SALU',0)/w:7.0f} {rd:7.1f} {wr:7.1f} {hit/max(hit+miss,1):6.2f} {(rd+wr)/max(us,1e-9):5.2f}"
          + "  " + " ".join(f"{c}={v/w:.0f}/w" for c, v in sorted(d.items()) if c not in ('SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_WAVES','SQ_INSTS_VALU','SQ_INSTS_SALU','FETCH_SIZE','WRITE_SIZE','TCC_HIT_sum','TCC_MISS_sum')))
PY
