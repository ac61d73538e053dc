#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile.sh <tag>'): rocprofv3 kernel-trace stats + PMC passes for bench.py.
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r6}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 10 --no-cpu --no-graph --sweeps 40"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_valu -o p -- python3 $ARGS > /dev/null 2> $OUT/pmc_valu.err
find $OUT -name "*.csv" | head -20
# the names the bench line quotes (fs_prof_kernels) must be the names of the kernel trace of this very command
python3 tools/check_trace_names.py $OUT/bench_trace.json $(find $OUT/trace -name "*kernel_stats.csv" | head -1) > $OUT/trace_names.txt 2>&1; echo "trace-name check rc=$?" >> $OUT/trace_names.txt
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
