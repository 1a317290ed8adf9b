import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library's MSDA_* test knobs (route forcing, measurement hooks) are honoured only with MSDA_ENABLE_HOOKS=1 and
# are read once (include/msda.h): tests switch them with monkeypatch.setenv, which is wrapped here to re-read them,
# and an autouse fixture re-reads them again once monkeypatch has restored the environment.
os.environ["MSDA_ENABLE_HOOKS"] = "1"


def _reload_knobs():
    """Re-read the knobs -- only if the library is already loaded (a CPU test on the oracle-backed double must not
    dlopen, let alone compile, the HIP library from a fixture teardown; a later load() reads the environment anyway)."""
    from devis_amd import _native
    if _native.is_loaded():
        _native.reload_knobs()


_orig_setenv, _orig_delenv = pytest.MonkeyPatch.setenv, pytest.MonkeyPatch.delenv


def _setenv(self, name, value, prepend=None):
    _orig_setenv(self, name, value, prepend)
    if name.startswith("MSDA_"):
        _reload_knobs()


def _delenv(self, name, raising=True):
    _orig_delenv(self, name, raising)
    if name.startswith("MSDA_"):
        _reload_knobs()


pytest.MonkeyPatch.setenv, pytest.MonkeyPatch.delenv = _setenv, _delenv


@pytest.fixture(autouse=True)
def _knobs_follow_environment():
    """Set up before (hence torn down after) monkeypatch: the knobs are re-read once the environment is restored."""
    yield
    _reload_knobs()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    """Load one committed golden fixture (made from the reference by tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


OP_FIXTURES = golden_names("op_")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; compiled on first use with gcc)."""
    from oracle import msda_oracle
    msda_oracle.build()
    return msda_oracle


@pytest.fixture()
def route_rules_only():
    """For tests that assert which kernel the route RULES (csrc/msda_api.hip) pick: the measured route table
    (devis_amd/routes.json) is taken out for the test and put back afterwards."""
    from devis_amd import _native
    _native.load()
    _native.clear_routes()
    try:
        yield
    finally:
        _native.clear_routes()
        _native._load_shipped_routes()
