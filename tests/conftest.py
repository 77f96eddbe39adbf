import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_data():
    """Seeded synthetic data set: 100 kb genome, 200 reads x 5 kb, 15 % PacBio-like errors."""
    from daccord_amd.synth import SynthData
    import pyoracle
    d = SynthData(100000, 200, 5000, seed=1)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    return d, ovl, piles
