#!/usr/bin/env python3
"""Make soc_amd/data/c3_dust50.txt: the 50-frequency optical table of config 3.

    python tools/make_c3_dust.py            (this container only: reads the reference's example data)

Input: tmp.dust of the reference's soc_example.zip (44 rows: frequency, asymmetry g, Q_abs, Q_sca of an
`eqdust` file, ASOC_aux.py:557-600).  Output: 50 log-spaced frequencies 1.5e11 .. 2e15 Hz with Q_abs, Q_sca
interpolated log-log and g linearly in log(frequency), clamped to the table's range, in the same `eqdust`
text layout so that soc_amd.files.read_dust reads it.  Data only -- no reference code involved.
"""
import io
import os
import zipfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = zipfile.ZipFile("/root/reference/soc_example.zip")
lines = z.read("tmp.dust").decode().split("\n")
assert lines[0].strip() == "eqdust"
gd, a, n = float(lines[1]), float(lines[2]), int(lines[3])
d = np.loadtxt(io.StringIO("\n".join(lines[4:4 + n])))
f, g, qa, qs = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
F = np.logspace(np.log10(1.5e11), np.log10(2.0e15), 50)
lf, LF = np.log(f), np.log(np.clip(F, f[0], f[-1]))
G = np.interp(LF, lf, g)
QA = np.exp(np.interp(LF, lf, np.log(qa)))
QS = np.exp(np.interp(LF, lf, np.log(qs)))
out = os.path.join(REPO, "soc_amd", "data", "c3_dust50.txt")
with open(out, "w") as fp:
    fp.write("eqdust\n %.5e\n %.5e\n%d\n" % (gd, a, len(F)))
    for i in range(len(F)):
        fp.write(" %.5e   %.5f   %.5e  %.5e\n" % (F[i], G[i], QA[i], QS[i]))
print(out)
