import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


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
