"""Build recipes for the test oracle.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

  build_oracle()  gcc  oracle/soc_oracle.c -> oracle/_build/liborc_soc.so, liborc_libm.so
  build_ref()     clang -x cl on the reference kernels WHERE THEY LIE in /root/reference
                  + oracle/ref_shim.cpp  -> oracle/_ref/ref_<tag>.so   (one per geometry,
                  because the reference bakes geometry in with -D macros, ASOC.py:344-362)

Nothing from /root/reference is copied: the .c/.cl files are only named on the compiler
command line.  oracle/_ref/ is git-ignored and exists only where /root/reference exists
(this container); the GPU box receives the prebuilt .so files with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = "/root/reference"
CLANG = "/opt/rocm/lib/llvm/bin/clang"
BUILD_DIR = os.path.join(HERE, "_build")
REF_DIR = os.path.join(HERE, "_ref")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _link_atomically(cmd, so):
    """run the link with a private output name, then rename: several test processes (pytest -n) may find the same
    library stale at the same time, and none may ever load a half-written file"""
    tmp = "%s.%d.tmp" % (so, os.getpid())
    try:
        subprocess.check_call(cmd + ["-o", tmp])
        os.replace(tmp, so)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build_oracle(force=False):
    """Compile both math modes of the CPU restatement.  Returns {mode: path}."""
    if os.environ.get("SOC_ORACLE_LIB_DIR"):
        # prebuilt variants of the two libraries, e.g. an -fsanitize=address,undefined build (tools/oracle_sanitize.sh)
        return {m: os.path.join(os.environ["SOC_ORACLE_LIB_DIR"], "liborc_%s.so" % m) for m in ("soc", "libm")}
    os.makedirs(BUILD_DIR, exist_ok=True)
    srcs = [os.path.join(HERE, "soc_oracle.c"), os.path.join(HERE, "a2e_oracle.c"),
            os.path.join(HERE, "soc_oracle_index.inc"), os.path.join(REPO, "soc_amd", "csrc", "soc_math.h")]
    out = {}
    for mode, flag in (("soc", []), ("libm", ["-DSOC_ORACLE_LIBM"])):
        so = os.path.join(BUILD_DIR, "liborc_%s.so" % mode)
        out[mode] = so
        if not force and _newer(so, srcs):
            continue
        cmd = ["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
               "-fno-fast-math", "-mfma", "-msse4.1", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] \
            + flag + [srcs[0], srcs[1], "-lm"]
        _link_atomically(cmd, so)
    return out


# --------------------------------------------------------------------------------------
# Reference builds
# --------------------------------------------------------------------------------------

def ref_defs(NX, NY, NZ, LEVELS, CELLS, BINS=2500, PS_METHOD=0, NO_PS=1, WITH_ABU=0,
             USE_EMWEIGHT=0, SAVE_INTENSITY=0, NOABSORBED=1, WITH_MSF=0, NDUST=1, MIRROR=0,
             GL=0.01, HPBG_WEIGHTED=0, WITH_ALI=0, ROI_STEP=0, ROI_NSIDE=16, WITH_ROI_LOAD=0, WITH_ROI_SAVE=0,
             STEP_WEIGHT=-1, SW_A=0.0, SW_B=0.0, DIR_WEIGHT=-1, DW_A=0.0, LEVEL_THRESHOLD=0, CR_HEATING=0.0, ROI_MAP=0,
             MAP_INTERPOLATION=0):
    """The -D list of ASOC.py:344-362 (+ -D NSIDE=128, ASOC.py:396) for one model."""
    AREA = 2 * (NX * NY + NY * NZ + NZ * NX)
    d = dict(NX=NX, NY=NY, NZ=NZ, BINS=BINS, WITH_ALI=WITH_ALI, PS_METHOD=PS_METHOD, FACTOR="1.0000e+20f",
             CELLS=CELLS, AREA=AREA, NO_PS=max(1, NO_PS), WITH_ABU=WITH_ABU, ROI_MAP=ROI_MAP, MAX_SPLIT=4300,
             SELEM=0, ROI_STEP=ROI_STEP, ROI_NSIDE=ROI_NSIDE, WITH_ROI_LOAD=WITH_ROI_LOAD, WITH_ROI_SAVE=WITH_ROI_SAVE,
             AXY="%.5ff" % (NX * NY / AREA), AXZ="%.5ff" % (NX * NZ / AREA), AYZ="%.5ff" % (NY * NZ / AREA),
             LEVELS=LEVELS, LENGTH="%.5ef" % (GL * 3.08567758e18), DO_SPLIT=0, POLSTAT=0,
             SW_A="%.3ef" % SW_A, SW_B="%.3ef" % SW_B, STEP_WEIGHT=STEP_WEIGHT, DIR_WEIGHT=DIR_WEIGHT, DW_A="%.3ef" % DW_A,
             LEVEL_THRESHOLD=LEVEL_THRESHOLD, POLRED=0, p00="0.2000f", MINLOS="-1.000e+00f", MAXLOS="1.000e+10f",
             FFS=1, NODIR=1, USE_EMWEIGHT=USE_EMWEIGHT, SAVE_INTENSITY=SAVE_INTENSITY,
             NOABSORBED=NOABSORBED, INTERPOLATE=0, ADHOC="1.00000e+00f", HPBG_WEIGHTED=HPBG_WEIGHTED,
             WITH_MSF=WITH_MSF, NDUST=NDUST, OPT_IS_HALF=0, POL_RHO_WEIGHT=0, MAP_INTERPOLATION=MAP_INTERPOLATION,
             MIRROR=MIRROR, CR_HEATING=int(CR_HEATING > 0), CR_HEATING_RATE="%.3ef" % CR_HEATING, NVIDIA=0, NSIDE=128)
    return ["-D%s=%s" % (k, v) for k, v in d.items()]


def build_ref(tag, force=False, **model):
    """Compile kernel_ASOC.c for one model + the shim -> oracle/_ref/ref_<tag>.so.
    Returns the path, or None when /root/reference is absent (GPU box)."""
    so = os.path.join(REF_DIR, "ref_%s.so" % tag)
    ksrc = os.path.join(REFERENCE, "kernel_ASOC.c")
    if not os.path.exists(ksrc):
        return so if os.path.exists(so) else None
    os.makedirs(REF_DIR, exist_ok=True)
    shim = os.path.join(HERE, "ref_shim.cpp")
    stamp = so + ".defs"
    defs = ref_defs(**model)
    if (not force and _newer(so, [shim, os.path.join(HERE, "ref_builtins.inc"), os.path.abspath(__file__), ksrc])
            and os.path.exists(stamp)
            and open(stamp).read() == " ".join(defs)):
        return so
    kobj = os.path.join(REF_DIR, "k_%s.%d.o" % (tag, os.getpid()))
    sobj = os.path.join(REF_DIR, "shim_%s.%d.o" % (tag, os.getpid()))
    common = ["-O2", "-fPIC", "-ffp-contract=off", "-target", "x86_64-unknown-linux-gnu"]
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header",
                           "-w", "-I", REFERENCE] + common + defs + ["-c", ksrc, "-o", kobj])
    sdefs = ["-DREF_ROI_LOAD=%d" % int(model.get("WITH_ROI_LOAD", 0)), "-DREF_ROI_SAVE=%d" % int(model.get("WITH_ROI_SAVE", 0))]
    subprocess.check_call([CLANG + "++", "-std=c++17", "-w"] + common + sdefs + ["-c", shim, "-o", sobj])
    _link_atomically([CLANG + "++", "-shared", "-Wl,-z,defs", kobj, sobj, "-lm", "-lpthread"], so)
    os.remove(kobj)
    os.remove(sobj)
    with open(stamp, "w") as fp:
        fp.write(" ".join(defs))
    return so


def build_ref_a2e(tag, NE, NFREQ, LOCAL, CELLS, NIP=5000, force=False):
    """kernel_A2E.c (DoSolve, EqTemperature) for one (NE, NFREQ, LOCAL, CELLS) -> oracle/_ref/refa2e_<tag>.so.
    -D list: A2E.py:283-286."""
    so = os.path.join(REF_DIR, "refa2e_%s.so" % tag)
    ksrc = os.path.join(REFERENCE, "kernel_A2E.c")
    if not os.path.exists(ksrc):
        return so if os.path.exists(so) else None
    os.makedirs(REF_DIR, exist_ok=True)
    drv = os.path.join(HERE, "ref_a2e.cpp")
    defs = ["-DNE=%d" % NE, "-DLOCAL=%d" % LOCAL, "-DNFREQ=%d" % NFREQ, "-DCELLS=%d" % CELLS, "-DNIP=%d" % NIP,
            "-DFACTOR=1.0000e+20f", "-DDUMP_TDUST=0"]
    stamp = so + ".defs"
    if (not force and _newer(so, [drv, os.path.join(HERE, "ref_builtins.inc"), os.path.abspath(__file__), ksrc])
            and os.path.exists(stamp) and open(stamp).read() == " ".join(defs)):
        return so
    kobj = os.path.join(REF_DIR, "ka2e_%s.%d.o" % (tag, os.getpid()))
    sobj = os.path.join(REF_DIR, "da2e_%s.%d.o" % (tag, os.getpid()))
    common = ["-O2", "-fPIC", "-ffp-contract=off", "-target", "x86_64-unknown-linux-gnu"]
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header",
                           "-w", "-I", REFERENCE] + common + defs + ["-c", ksrc, "-o", kobj])
    subprocess.check_call([CLANG + "++", "-std=c++17", "-w"] + common + ["-c", drv, "-o", sobj])
    _link_atomically([CLANG + "++", "-shared", "-Wl,-z,defs", kobj, sobj, "-lm", "-lpthread"], so)
    os.remove(kobj)
    os.remove(sobj)
    with open(stamp, "w") as fp:
        fp.write(" ".join(defs))
    return so


def build_ref_a2e_pre(force=False):
    """kernel_A2E_pre.c (PrepareIntegrationWeightsTrapezoid, PrepareTdown) -> oracle/_ref/refa2epre.so.
    -D list: A2E_pre.py:134 (FACTOR only; NE and NFREQ are kernel arguments)."""
    so = os.path.join(REF_DIR, "refa2epre.so")
    ksrc = os.path.join(REFERENCE, "kernel_A2E_pre.c")
    if not os.path.exists(ksrc):
        return so if os.path.exists(so) else None
    os.makedirs(REF_DIR, exist_ok=True)
    drv = os.path.join(HERE, "ref_a2e_pre.cpp")
    if not force and _newer(so, [drv, os.path.join(HERE, "ref_builtins.inc"), os.path.abspath(__file__), ksrc]):
        return so
    kobj = os.path.join(REF_DIR, "ka2epre.%d.o" % os.getpid())
    sobj = os.path.join(REF_DIR, "da2epre.%d.o" % os.getpid())
    common = ["-O2", "-fPIC", "-ffp-contract=off", "-target", "x86_64-unknown-linux-gnu"]
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header",
                           "-w", "-I", REFERENCE] + common + ["-DFACTOR=1.0000e+20f", "-c", ksrc, "-o", kobj])
    subprocess.check_call([CLANG + "++", "-std=c++17", "-w"] + common + ["-c", drv, "-o", sobj])
    _link_atomically([CLANG + "++", "-shared", "-Wl,-z,defs", kobj, sobj, "-lm", "-lpthread"], so)
    os.remove(kobj)
    os.remove(sobj)
    return so


def sca_defs(NX, NY, NZ, LEVELS, CELLS, BINS=2500, PS_METHOD=0, NO_PS=1, WITH_ABU=0, USE_EMWEIGHT=0, FFS=1, GL=0.01,
             MIRROR=0, HPBG_WEIGHTED=0, WITH_MSF=0, NDUST=1):
    """The -D list of ASOCS.py:133-147 for one model."""
    AREA = 2 * (NX * NY + NY * NZ + NZ * NX)
    d = dict(NX=NX, NY=NY, NZ=NZ, BINS=BINS, WITH_ALI=0, PS_METHOD=PS_METHOD, CELLS=CELLS, AREA=AREA, NO_PS=max(1, NO_PS),
             WITH_ABU=WITH_ABU, FACTOR="1.0000e+20f", AXY="%.5ff" % (NX * NY / AREA), AXZ="%.5ff" % (NX * NZ / AREA),
             AYZ="%.5ff" % (NY * NZ / AREA), LEVELS=LEVELS, LENGTH="%.5ef" % (GL * 3.08567758e18), POLSTAT=0,
             SW_A="0.000e+00f", SW_B="0.000e+00f", STEP_WEIGHT=-1, DIR_WEIGHT=-1, DW_A="0.000e+00f", LEVEL_THRESHOLD=0,
             POLRED=0, WITH_COLDEN=0, MINLOS="-1.000e+00f", MAXLOS="1.000e+10f", FFS=FFS, BG_METHOD=0,
             USE_EMWEIGHT=USE_EMWEIGHT, HPBG_WEIGHTED=HPBG_WEIGHTED, WITH_MSF=WITH_MSF, NDUST=NDUST, OPT_IS_HALF=0, WITH_ROI_LOAD=0, ROI_NSIDE=16,
             MIRROR=MIRROR, NVIDIA=0)
    return ["-D%s=%s" % (k, v) for k, v in d.items()]


def build_ref_sca(tag, force=False, **model):
    """kernel_ASOC_sca.c for one model -> oracle/_ref/refsca_<tag>.so"""
    so = os.path.join(REF_DIR, "refsca_%s.so" % tag)
    ksrc = os.path.join(REFERENCE, "kernel_ASOC_sca.c")
    if not os.path.exists(ksrc):
        return so if os.path.exists(so) else None
    os.makedirs(REF_DIR, exist_ok=True)
    drv = os.path.join(HERE, "ref_sca.cpp")
    defs = sca_defs(**model)
    stamp = so + ".defs"
    if (not force and _newer(so, [drv, os.path.join(HERE, "ref_builtins.inc"), os.path.abspath(__file__), ksrc])
            and os.path.exists(stamp) and open(stamp).read() == " ".join(defs)):
        return so
    kobj = os.path.join(REF_DIR, "ksca_%s.%d.o" % (tag, os.getpid()))
    sobj = os.path.join(REF_DIR, "dsca_%s.%d.o" % (tag, os.getpid()))
    common = ["-O2", "-fPIC", "-ffp-contract=off", "-target", "x86_64-unknown-linux-gnu"]
    # SimRAM_CL reads the local `idust` without ever setting it when WITH_MSF==0
    # (kernel_ASOC_sca.c:1151, used in DSC[idust*BINS+...] :1385): undefined behaviour that GPU
    # compilers resolve to 0.  -ftrivial-auto-var-init=zero gives the x86 build the same value.
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header",
                           "-ftrivial-auto-var-init=zero", "-w", "-I", REFERENCE] + common + defs + ["-c", ksrc, "-o", kobj])
    subprocess.check_call([CLANG + "++", "-std=c++17", "-w"] + common + ["-c", drv, "-o", sobj])
    _link_atomically([CLANG + "++", "-shared", "-Wl,-z,defs", kobj, sobj, "-lm", "-lpthread"], so)
    os.remove(kobj)
    os.remove(sobj)
    with open(stamp, "w") as fp:
        fp.write(" ".join(defs))
    return so


def build_ref_map(tag, force=False, NSIDE=8, **model):
    """kernel_ASOC_map.c for one model -> oracle/_ref/refmap_<tag>.so (the -D list of ASOC.py:344-362 plus
    -D NSIDE=<NPIX.x>, ASOC.py:2934)"""
    so = os.path.join(REF_DIR, "refmap_%s.so" % tag)
    ksrc = os.path.join(REFERENCE, "kernel_ASOC_map.c")
    if not os.path.exists(ksrc):
        return so if os.path.exists(so) else None
    os.makedirs(REF_DIR, exist_ok=True)
    drv = os.path.join(HERE, "ref_map.cpp")
    defs = [d for d in ref_defs(**model) if not d.startswith("-DNSIDE=")] + ["-DNSIDE=%d" % NSIDE]
    stamp = so + ".defs"
    if (not force and _newer(so, [drv, os.path.join(HERE, "ref_builtins.inc"), os.path.abspath(__file__), ksrc])
            and os.path.exists(stamp) and open(stamp).read() == " ".join(defs)):
        return so
    kobj = os.path.join(REF_DIR, "kmap_%s.%d.o" % (tag, os.getpid()))
    sobj = os.path.join(REF_DIR, "dmap_%s.%d.o" % (tag, os.getpid()))
    common = ["-O2", "-fPIC", "-ffp-contract=off", "-target", "x86_64-unknown-linux-gnu"]
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header", "-ftrivial-auto-var-init=zero",
                           "-w", "-I", REFERENCE] + common + defs + ["-c", ksrc, "-o", kobj])
    subprocess.check_call([CLANG + "++", "-std=c++17", "-w"] + common + ["-DROI_MAP=%d" % model.get("ROI_MAP", 0), "-c", drv, "-o", sobj])
    _link_atomically([CLANG + "++", "-shared", "-Wl,-z,defs", kobj, sobj, "-lm", "-lpthread"], so)
    os.remove(kobj)
    os.remove(sobj)
    with open(stamp, "w") as fp:
        fp.write(" ".join(defs))
    return so


def map_ref_models():
    sys.path.insert(0, REPO)
    from soc_amd import synth
    oct8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    oct104 = synth.octree_cloud(104, levels=3, frac=0.002, seed=11)
    return {
        "c8": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512),
        "c8abu": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ABU=1),
        "oct8": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS),
        "oct104": dict(NX=104, NY=104, NZ=104, LEVELS=oct104.LEVELS, CELLS=oct104.CELLS),
        "c208": dict(NX=208, NY=208, NZ=208, LEVELS=1, CELLS=208 ** 3),
        "oct8roi": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, ROI_MAP=1),
        "oct8thr": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, LEVEL_THRESHOLD=1),
        "c8mi1": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, MAP_INTERPOLATION=1),
        "oct8mi1": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, MAP_INTERPOLATION=1),
        "oct8mi2": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, MAP_INTERPOLATION=2),
        "oct104mi2": dict(NX=104, NY=104, NZ=104, LEVELS=oct104.LEVELS, CELLS=oct104.CELLS, MAP_INTERPOLATION=2),
    }


def sca_ref_models():
    sys.path.insert(0, REPO)
    from soc_amd import synth
    oct8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    return {
        "c8": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512),
        "c8noffs": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, FFS=0),
        "c8abu": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ABU=1),
        "c8ps": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NO_PS=2),
        "c8ps1": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NO_PS=2, PS_METHOD=1),
        "c8ps2": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NO_PS=2, PS_METHOD=2),
        "c8ps4": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NO_PS=1, PS_METHOD=4),
        "c8ps5": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NO_PS=2, PS_METHOD=5),
        "c128": dict(NX=128, NY=128, NZ=128, LEVELS=1, CELLS=128 ** 3),
        "c8mir": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, MIRROR=5),
        "c8hpw": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, HPBG_WEIGHTED=1),
        "oct8": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS),
        "oct8emw": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, USE_EMWEIGHT=1),
        "c8msf": dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ABU=1, WITH_MSF=1, NDUST=3, NO_PS=2),
        "oct8msf": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, WITH_ABU=1, WITH_MSF=1, NDUST=3),
    }


def a2e_ref_models():
    return {
        "ne16": dict(NE=16, NFREQ=12, LOCAL=8, CELLS=40, NIP=500),
        "ne64": dict(NE=64, NFREQ=40, LOCAL=8, CELLS=64, NIP=5000),
        "ne128": dict(NE=128, NFREQ=50, LOCAL=8, CELLS=8192, NIP=5000),
    }


def ref_models():
    """Every reference build the tests, the golden generator and bench.py's cpu_baseline use."""
    sys.path.insert(0, REPO)
    from soc_amd import synth
    oct8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    oct104 = synth.octree_cloud(104, levels=3, frac=0.002, seed=11)
    m = {
        "c8":      dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512),
        "c8int":   dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, NOABSORBED=0),
        "c8int2":  dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, SAVE_INTENSITY=2),
        "oct8int2": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, SAVE_INTENSITY=2),
        "c8abu":   dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ABU=1),
        "r654":    dict(NX=6, NY=5, NZ=4, LEVELS=1, CELLS=120),
        "c16":     dict(NX=16, NY=16, NZ=16, LEVELS=1, CELLS=4096),
        "c32":     dict(NX=32, NY=32, NZ=32, LEVELS=1, CELLS=32768),
        "oct4":    dict(NX=4, NY=4, NZ=4, LEVELS=3, CELLS=80),
        "oct8":    dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS),
        "oct8emw": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, USE_EMWEIGHT=1),
        "oct104":  dict(NX=104, NY=104, NZ=104, LEVELS=oct104.LEVELS, CELLS=oct104.CELLS),
        "c128":    dict(NX=128, NY=128, NZ=128, LEVELS=1, CELLS=128 ** 3),
        "c8mir":   dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, MIRROR=25),
        "oct8mir": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, MIRROR=6),
        "oct8emw2": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, USE_EMWEIGHT=2),
        "oct8ali": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, WITH_ALI=1),
        "c8hpw":   dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, HPBG_WEIGHTED=1),
        "oct8hpw": dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, HPBG_WEIGHTED=1, NOABSORBED=0),
    }
    # region of interest: packets entering ROI saved / loaded at the surface (ROI_STEP and ROI_NSIDE are -D constants)
    m["c8roi"] = dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ROI_SAVE=1, ROI_STEP=2, ROI_NSIDE=2)
    m["oct8roi"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, WITH_ROI_SAVE=1, ROI_STEP=1, ROI_NSIDE=4)
    m["c8roil"] = dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, WITH_ROI_LOAD=1, ROI_NSIDE=2)
    m["oct8roils"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, WITH_ROI_LOAD=1, WITH_ROI_SAVE=1,
                          ROI_STEP=2, ROI_NSIDE=2, NOABSORBED=0)
    for k in (1, 2, 4, 5):
        m["c8ps%d" % k] = dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, PS_METHOD=k, NO_PS=2)
    m["c8ps0"] = dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, PS_METHOD=0, NO_PS=2)
    m["oct8cr"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, CR_HEATING=2.5)      # EqTemperature with cosmic-ray heating
    m["oct8ps0"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, PS_METHOD=0, NO_PS=2, NOABSORBED=0)   # point sources inside a hierarchy
    # weighted sampling and per-dust scattering functions (kernel_ASOC.c:516-535,736-799; -D values as ASOC.py:348,357-358 prints them)
    m["c8sw1"] = dict(NX=8, NY=8, NZ=8, LEVELS=1, CELLS=512, STEP_WEIGHT=1, SW_A=0.5, SW_B=0.0)
    m["oct8sw2"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, STEP_WEIGHT=2, SW_A=0.7, SW_B=0.4)
    m["oct8msf"] = dict(NX=8, NY=8, NZ=8, LEVELS=oct8.LEVELS, CELLS=oct8.CELLS, WITH_ABU=1, WITH_MSF=1, NDUST=3)
    # config 3 of BASELINE.json (bench.py's cpu_baseline): synth.octree_cloud(256, levels=4, frac=0.10, seed=1234);
    # the cell count is written out (the cloud takes a second and 400 MB to generate) and checked by bench.py
    m["oct256"] = dict(NX=256, NY=256, NZ=256, LEVELS=4, CELLS=49526352, GL=0.02)
    return m


def build_all_refs(force=False):
    out = {}
    for tag, model in ref_models().items():
        out[tag] = build_ref(tag, force=force, **model)
    for tag, model in map_ref_models().items():
        out["map_" + tag] = build_ref_map(tag, force=force, **model)
    for tag, model in a2e_ref_models().items():
        out["a2e_" + tag] = build_ref_a2e(tag, force=force, **model)
    for tag, model in sca_ref_models().items():
        out["sca_" + tag] = build_ref_sca(tag, force=force, **model)
    out["a2e_pre"] = build_ref_a2e_pre(force=force)
    return out


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_all_refs(force="--force" in sys.argv))
