#!/bin/bash
# VALU utilisation of the brick walk (run on the GPU box from the repo root): one rocprofv3 --pmc pass over a
# point-source sweep of the config-3 geometry (tools/exp_c3.py), summed per kernel -> profiles/<tag>_valu_summary.csv
# usage: tools/profile_valu.sh TAG
set -e
TAG=${1:-r02_c3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/valu_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/p -o v -- python3 $ROOT/tools/exp_c3.py --kind ps --launches 16 --global0 16777216 --packets 6e7 '{}' > $OUT/run.log 2>&1
cd $ROOT
python3 - "$(find $OUT/p -name '*counter_collection.csv' | head -1)" profiles/${TAG}_valu_summary.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
with open(sys.argv[1]) as fp:
    for row in csv.DictReader(fp):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[k] += 1
names = ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY"]
with open(sys.argv[2], "w") as out:
    out.write("# rocprofv3 --pmc " + " ".join(names) + " -- python3 tools/exp_c3.py --kind ps --launches 16 --global0 16777216 --packets 6e7 '{}'\n")
    out.write("# sums over the dispatches of each kernel (SQ counters in quad-cycles, summed over the SEs' SQs)\n")
    out.write("kernel,dispatches," + ",".join(names) + ",VALU_per_wave_cycle,wait_per_wave_cycle\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        out.write("%s,%d,%s,%.4f,%.4f\n" % (k, calls[k], ",".join("%.6g" % v.get(n, 0.0) for n in names),
                                           v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, v.get("SQ_WAIT_INST_ANY", 0.0) / wc))
print(open(sys.argv[2]).read())
PY
mkdir -p $OUT/profiles && cp profiles/${TAG}_valu_summary.csv $OUT/profiles/
