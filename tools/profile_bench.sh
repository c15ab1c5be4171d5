#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block (run on the GPU box from the repo root):
#   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
#   profiles/<tag>_pmc_summary.csv    HBM traffic from two separate --pmc passes
#   profiles/traffic.json             what bench.py reports as roofline.traffic
set -e
TAG=${1:-r01_c2_brick}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > $OUT/bench_write.log 2>&1
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats.csv
python3 tools/summarize_pmc.py $TAG 16 $OUT/fetch $OUT/write
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.csv profiles/traffic.json $OUT/
tail -1 $OUT/bench_stats.log
