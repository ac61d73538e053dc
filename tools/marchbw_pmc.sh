#!/bin/bash
# Run ON THE GPU BOX: the marching-stream microbenchmark (tools/marchbw.hip -> tools/marchbw.bin) timed, then under two rocprofv3 PMC passes
set -u
OUT=gpurun_out/marchbw; mkdir -p $OUT; export TMPDIR=/tmp
tools/marchbw.bin 20 short > $OUT/timing.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- tools/marchbw.bin 2 short > /dev/null 2> $OUT/fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/write -o p -- tools/marchbw.bin 2 short > /dev/null 2> $OUT/write.err
python3 - $OUT <<'P' | tee $OUT/summary.txt
import csv, glob, collections, sys, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
print(open(out + '/timing.txt').read())
for k, d in agg.items():
    rd = 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e6 / max(n[(k, 'FETCH_SIZE')], 1); wr = d.get('WRITE_SIZE', 0) * 1024 / 1e6 / max(n[(k, 'WRITE_SIZE')], 1)
    hit, miss = d.get('TCC_HIT_sum', 0), d.get('TCC_MISS_sum', 0)
    print(f"{k:60s} read {rd:7.1f} MB  write {wr:7.1f} MB  L2 hit {hit / max(hit + miss, 1):.2f}")
P
