#!/bin/bash
# rocprofv3 kernel statistics + HBM counters of the scattered-light launches run as sweeps of rays (config-4 geometry, batches of 8 launches):
#   profiles/<round>_sca_rays_kernel_stats.csv, _lines.json, _pmc_summary.csv     (run on the GPU box from the repo root)
set -e
R=${1:-r03}; N=${2:-8}
ARGS="--launches $N --cl-global 8388608 --ps-global 2097152"
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_${R}_rays; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/tools/exp_sca.py $ARGS > $OUT/line.json 2> $OUT/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/tools/exp_sca.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/tools/exp_sca.py $ARGS > $OUT/write.log 2>&1
echo "write done"
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) profiles/${R}_sca_rays_kernel_stats.csv
cp $OUT/line.json profiles/${R}_sca_rays_lines.json
python3 - $OUT $R "$ARGS" <<'PY'
import csv, glob, sys, collections, json
out, r, args = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("fetch", "write"):
    for f in glob.glob("%s/%s/**/*_counter_collection.csv" % (out, d), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if "soc_" in name:
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[(name, row["Counter_Name"])] += 1
steps = sum(2 * json.loads(l)["ray_steps"] for l in open("%s/line.json" % out))          # two repetitions of every launch kind per run (the line holds the better one)
with open("profiles/%s_sca_rays_pmc_summary.csv" % r, "w") as fp:
    fp.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of tools/exp_sca.py %s; KiB summed over the dispatches of the run\n" % args)
    fp.write("# the run's rays take about %.4g cell steps (2 repetitions x 3 launch kinds): bytes per step in the last column\nkernel,counter,KiB,dispatches,bytes_per_ray_step\n" % steps)
    for name, v in acc.items():
        for c, x in v.items():
            fp.write("%s,%s,%.6g,%d,%.3f\n" % (name.replace(",", ";"), c, x, n[(name, c)], x * 1024.0 / steps))
PY
mkdir -p $OUT/profiles; cp profiles/${R}_sca_rays_* $OUT/profiles/
cat profiles/${R}_sca_rays_pmc_summary.csv
