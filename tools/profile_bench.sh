#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block (run on the GPU box from the repo root):
#   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
#   profiles/<tag>_pmc_summary.csv    HBM traffic from two separate --pmc passes
#   profiles/<tag>_bench_line.json    the bench line of the kernel-trace run
#   profiles/traffic.json             what bench.py reports as roofline.traffic (one entry per workload)
# usage: tools/profile_bench.sh TAG WORKLOAD STEPS [extra bench.py arguments]
set -e
TAG=${1:-r02_c3}
WORK=${2:-C3}
STEPS=${3:-4}
shift 3 || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# the driver runs --steps 20 --warmup 5: its timed steps are 5 .. 24; profiled here without the warm-up launches
ARGS="--workload $WORK --steps $STEPS --warmup 0 --first-step 5 --no-cpu-baseline --no-reference-shape --no-per-kind $@"
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/bench_stats.log 2>&1
timeout -k 10 1000 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout -k 10 1000 rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats.csv
grep '^{' $OUT/bench_stats.log > profiles/${TAG}_bench_line.json
python3 tools/summarize_pmc.py $TAG $STEPS $OUT/fetch $OUT/write $WORK
mkdir -p $OUT/profiles
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.csv profiles/${TAG}_bench_line.json profiles/traffic.json $OUT/profiles/
cat profiles/${TAG}_kernel_stats.csv | cut -c1-200
tail -c 1500 profiles/${TAG}_bench_line.json
