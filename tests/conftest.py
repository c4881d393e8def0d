import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hapmap():
    from snprelate_amd.gds import open_gds
    return open_gds(os.path.join(GOLDEN, "hapmap_geno.gds"))


from oracle.synth import synth_geno  # noqa: E402,F401
