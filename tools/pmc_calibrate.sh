#!/bin/bash
# Calibrate the rocprofv3 HBM counters on kernels of KNOWN traffic (tools/membw.hip: 537 MB read-only / 537+537 MB copy),
# separate --pmc passes as the MI355X guide prescribes.  Run on the GPU box; writes gpurun_out/pmc_cal/summary.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_cal
mkdir -p $OUT
[ -x tools/membw.bin ] || hipcc -O3 --offload-arch=gfx950 tools/membw.hip -o tools/membw.bin
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- ./tools/membw.bin > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- ./tools/membw.bin > $OUT/write.log 2>&1
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections
def load(d, counter):
    f = glob.glob(f"gpurun_out/pmc_cal/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
true_mb = 536.870912
print("kernel (known traffic: 536.9 MB read; copies also write 536.9 MB)      FETCH_SIZE KiB -> MB   ratio true/reported   WRITE_SIZE KiB -> MB  ratio")
for k in sorted(fe):
    f_mb = fe[k] * 1024 / 1e6
    w_mb = wr.get(k, 0.0) * 1024 / 1e6
    wr_true = true_mb if "copy" in k else 0.0
    print(f"{k[:60]:60s} {f_mb:10.1f}   {true_mb / f_mb if f_mb else float('nan'):6.3f}      {w_mb:10.1f}   {(wr_true / w_mb) if w_mb > 1 else float('nan'):6.3f}")
PY
cat $OUT/summary.txt
