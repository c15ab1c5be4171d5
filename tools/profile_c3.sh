#!/bin/bash
# rocprofv3 kernel statistics of `bench.py --workload C3` (one sweep of 16 steps): profiles/r01_c3_kernel_stats.csv
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_c3; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --workload C3 --steps 16 --warmup 0 --no-cpu-baseline > $OUT/bench.log 2>&1
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/r01_c3_kernel_stats.csv
grep '^{' $OUT/bench.log > $OUT/r01_c3_bench_line_profiled.json
cat $OUT/r01_c3_kernel_stats.csv; tail -c 600 $OUT/r01_c3_bench_line_profiled.json
