#!/bin/bash
# Average latencies of the brick-sweep kernels' LDS and vector-memory instructions: SQ_INST_LEVEL_* (instructions in flight, summed over
# cycles) / SQ_INSTS_* of one tools/exp_c3.py run; separate --pmc passes.   usage: tools/prof_levels.sh OUTDIR [exp_c3.py arguments]
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -o a -- python3 $ROOT/tools/exp_c3.py "$@" > $OUT/loga.txt 2>&1 || true
echo "pass a done"
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_IFETCH_LEVEL --output-format csv -d $OUT/b -o b -- python3 $ROOT/tools/exp_c3.py "$@" > $OUT/logb.txt 2>&1 || true
echo "pass b done"
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in "ab":
    for f in glob.glob(out + '/' + d + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as fp:
    for k, v in acc.items():
        if 'brick' not in k and 'lray' not in k:
            continue
        line = k + ' ' + ' '.join('%s=%.4g' % (a, b) for a, b in sorted(v.items()))
        print(line)
        fp.write(line + '\n')
PY
