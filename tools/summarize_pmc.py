"""Sum rocprofv3 --pmc counter CSVs over the soc_* dispatches of a bench run and write
profiles/<tag>_pmc_summary.csv + profiles/traffic.json (per bench step).

usage: summarize_pmc.py TAG STEPS FETCH_DIR WRITE_DIR [WORKLOAD]
FETCH_DIR: run with --pmc FETCH_SIZE;  WRITE_DIR: run with --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum
(separate passes, as MI355X_MICROARCH.md's HBM section prescribes).  FETCH_SIZE/WRITE_SIZE are KiB."""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sums(d):
    acc = collections.defaultdict(float)
    nk = collections.Counter()
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'soc_' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value'])
                nk[r['Kernel_Name'].split('(')[0]] += 1
    return dict(acc), dict(nk)


def main():
    tag, steps, dfetch, dwrite = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    workload = sys.argv[5] if len(sys.argv) > 5 else 'C2'
    a, nka = sums(dfetch)
    b, nkb = sums(dwrite)
    fetch = a['FETCH_SIZE'] * 1024.0 / steps
    write = b['WRITE_SIZE'] * 1024.0 / steps
    atom = b.get('TCC_EA0_ATOMIC_sum', 0.0) / steps
    with open(os.path.join(REPO, 'profiles', tag + '_pmc_summary.csv'), 'w') as fp:
        fp.write('# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE TCC_EA0_ATOMIC_sum), '
                 'python bench.py --workload %s --steps %d --warmup 0 --first-step 5 --no-cpu-baseline --no-reference-shape --no-per-kind\n' % (workload, steps))
        fp.write('# sums over all soc_* dispatches of the run / %d steps; dispatch counts: %s\n' % (steps, json.dumps(nka)))
        fp.write('counter,per_step\n')
        fp.write('FETCH_SIZE_KiB,%.6g\nWRITE_SIZE_KiB,%.6g\nTCC_EA0_ATOMIC_sum,%.6g\n' % (fetch / 1024, write / 1024, atom))
    p = os.path.join(REPO, 'profiles', 'traffic.json')
    old = json.load(open(p)) if os.path.exists(p) else {}
    if "hbm_bytes_per_launch" in old:                      # round-1 layout: one workload (C2) at the top level
        old = {"C2": old}
    out = {
        "profile": tag, "workload": workload,
        "kernel": "brick path, %d steps per sweep: %s (per bench step)" % (steps, ", ".join("%s x%d" % (k, v // steps) for k, v in sorted(nka.items()))),
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "atomic_requests": atom,
        "note": "FETCH_SIZE and WRITE_SIZE in KiB x 1024 summed over all dispatches of the run / steps, separate PMC passes. "
                "Raw values (scattered 64-B packet records and 64-B tally rows; the gfx950 x2 FETCH correction applies to "
                "wide coalesced streams only).",
    }
    if "direct_kernel" in old.get(workload, {}):
        out["direct_kernel"] = old[workload]["direct_kernel"]
    old[workload] = out
    json.dump(old, open(p, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
