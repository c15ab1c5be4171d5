#!/bin/bash
# rocprofv3 kernel statistics + HBM counters of the two other hot kernels (run on the GPU box from the repo root):
#   profiles/<round>_a2e_*  soc_a2e_dosolve_kernel at NE=128 / NFREQ=50 / batch 8192 (config 5)
#   profiles/<round>_sca_*  soc_sca_kernel on the config-4 geometry (256^3-root octree, 3 observers)
set -e
R=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_${R}_kernels; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for K in a2e sca; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${K}_stats -o s -- python3 $ROOT/tools/exp_$K.py > $OUT/${K}_line.json 2> $OUT/${K}_stats.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${K}_fetch -o f -- python3 $ROOT/tools/exp_$K.py > $OUT/${K}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${K}_write -o w -- python3 $ROOT/tools/exp_$K.py > $OUT/${K}_write.log 2>&1
done
cd $ROOT
for K in a2e sca; do
  cp $(find $OUT/${K}_stats -name '*kernel_stats.csv' | head -1) profiles/${R}_${K}_kernel_stats.csv
  cp $OUT/${K}_line.json profiles/${R}_${K}_lines.json
  python3 - $OUT $K $R <<'PY'
import csv, glob, sys, collections
out, k, r = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("fetch", "write"):
    for f in glob.glob("%s/%s_%s/**/*_counter_collection.csv" % (out, k, d), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if "soc_" in name:
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[(name, row["Counter_Name"])] += 1
with open("profiles/%s_%s_pmc_summary.csv" % (r, k), "w") as fp:
    fp.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of tools/exp_%s.py; KiB summed over the dispatches\nkernel,counter,KiB,dispatches\n" % k)
    for name, v in acc.items():
        for c, x in v.items():
            fp.write("%s,%s,%.6g,%d\n" % (name.replace(",", ";"), c, x, n[(name, c)]))
PY
done
mkdir -p $OUT/profiles; cp profiles/${R}_a2e_* profiles/${R}_sca_* $OUT/profiles/
cat profiles/${R}_a2e_lines.json profiles/${R}_sca_lines.json
