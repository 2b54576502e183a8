import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Build (if stale) and load the native library; needed by CPU and GPU tests alike."""
    from pyfaceanalysis_amd import build, _capi
    build.build()
    return _capi.lib()


_NETS = {}


def get_net(name, **kw):
    """Session cache of trained synthetic hierarchies (training is deterministic)."""
    from pyfaceanalysis_amd import synth
    key = (name, tuple(sorted(kw.items())))
    if key not in _NETS:
        _NETS[key] = synth.build_preset(name, **kw)
    return _NETS[key]


@pytest.fixture(scope="session")
def nets():
    return get_net
