"""The C-ABI library loads and exports every symbol include/soc_hip.h declares (no compute)."""
import ctypes
import os
import re

from soc_amd import lib as soclib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "soc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(soc_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ("soc_create", "soc_destroy", "soc_set_grid", "soc_sim_pb", "soc_sim_cl", "soc_read_tally",
                 "soc_zero", "soc_last_error", "soc_set_scatter_table", "soc_set_optical"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from soc_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), "libsoc_hip.so does not export %s" % s


def test_python_binding_covers_the_header():
    assert sorted(soclib.API) == declared_symbols()
    soclib.load_library()      # sets restype/argtypes for every entry; raises if one is missing


def test_no_torch_types_in_signatures():
    text = open(os.path.join(REPO, "include", "soc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in text and "at::" not in text and "#include <hip" not in text
