"""Build libsoc_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m soc_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract:
the kernels' fp32 expressions must round exactly like the host build of soc_math.h.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsoc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["soc_kernels.hip", "soc_brick.hip", "soc_a2e.hip", "soc_a2e_pre.hip", "soc_sca.hip", "soc_emit.hip", "soc_map.hip", "soc_capi.hip"]
HEADERS = ["soc_dev.h", "soc_math.h", "soc_rng.h", "soc_walk.h", "soc_ltree.h", "soc_lbricks.h", "soc_octbricks.h", os.path.join("..", "..", "include", "soc_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-munsafe-fp-atomics", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    """Compile if sources are newer than the library.  Returns the library path."""
    if not force and not _stale():
        return LIB
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
