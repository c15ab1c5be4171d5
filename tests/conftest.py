import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_soc():
    from oracle.pyoracle import Oracle
    return Oracle("soc")


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle.pyoracle import Oracle
    return Oracle("libm")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0.  Fails loudly (no fallback) when the library or GPU is missing.
    (soc_amd.lib loads torch's HIP runtime before libsoc_hip.so where torch is installed: two runtimes in one process, the
    one initialised second sees no device -- tests/test_gpu_binding.py::test_library_and_torch_load_in_either_order.)"""
    from soc_amd.lib import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture
def tuned(engine):
    """Set brick-sweep shape parameters (Engine.set_tuning) for one test; the built-in choices come back afterwards."""
    used = set()

    def set_(**kw):
        used.update(kw)
        engine.set_tuning(**kw)
    yield set_
    engine.set_tuning(**{k: 0 for k in used})
