set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 800 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --workload C3 --steps 20 --warmup 0 --no-cpu-baseline --no-reference-shape > $OUT/bench.log 2>&1
cd $ROOT
F=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 - "$F" $OUT/passes.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as fp:
    r = csv.DictReader(fp)
    for row in r:
        n = row["Kernel_Name"]
        if "soc_lbrick_pass" in n or "soc_brick_scan" in n or "soc_brick_scatter" in n:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), "P0" if "false, 0" in n else ("P2" if "false, 2" in n else ("scan" if "scan" in n else "scat")), int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0)))
rows.sort()
with open(sys.argv[2], "w") as out:
    t0 = rows[0][0]
    for s, e, k, g in rows:
        out.write("%s %.3f %.3f %d\n" % (k, (s - t0) / 1e6, (e - s) / 1e6, g))
PY
rm -rf $OUT/t
tail -c 600 $OUT/bench.log
