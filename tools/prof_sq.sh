#!/bin/bash
# SQ counters of the brick-sweep kernels for one tools/exp_c3.py run (three separate --pmc passes).
# usage: tools/prof_sq.sh OUTDIR [exp_c3.py arguments]
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o a -- python3 $ROOT/tools/exp_c3.py "$@" > $OUT/loga.txt 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 $ROOT/tools/exp_c3.py "$@" > $OUT/logb.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/c -o c -- python3 $ROOT/tools/exp_c3.py "$@" > $OUT/logc.txt 2>&1 || true
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in "abc":
    for f in glob.glob(out + '/' + d + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as fp:
    for k, v in acc.items():
        if 'brick' not in k:
            continue
        line = k + ' ' + ' '.join('%s=%.4g' % (a, b) for a, b in sorted(v.items()))
        print(line)
        fp.write(line + '\n')
PY
grep "rep 1" $OUT/loga.txt | tail -2
