#!/bin/bash
# Vector-memory path counters (TA / TCP / address translation) of the brick-sweep kernels for one tools/exp_c3.py run.
# usage: tools/prof_mem.sh OUTDIR [exp_c3.py arguments]      (four separate --pmc passes; run on the GPU box)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; ARGS=("$@"); rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
# (at most four counters of one block per pass: more "exceeds the capabilities of the hardware"; every pass under a time limit)
P() { d=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$d -o $d -- python3 $ROOT/tools/exp_c3.py "${ARGS[@]}" > $OUT/log$d.txt 2>&1 || echo "pass $d failed" ; }
P a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
P b TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum
P c TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
P d TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in "abcd":
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
