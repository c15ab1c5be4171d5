#!/bin/bash
# libsoc_hip with the wave-level counters of the brick walk and the phase timers of DoSolve compiled in (-DSOC_BRICK_PROF -DSOC_A2E_PROF): soc_amd/libsoc_prof.so
# (git-ignored; travels with gpurun).  Use: SOC_HIP_LIB=soc_amd/libsoc_prof.so python tools/exp_c3.py ...
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
import os, subprocess
from soc_amd import build
cs = build.CSRC
out = os.path.join(build.HERE, "libsoc_prof.so")
subprocess.check_call([build.HIPCC] + build.FLAGS + ["-DSOC_BRICK_PROF", "-DSOC_A2E_PROF", "-shared", "-o", out] + [os.path.join(cs, s) for s in build.SOURCES])
print(out)
PY
