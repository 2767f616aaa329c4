import os
import os.path as osp
import sys

import numpy as np
import pytest

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = osp.join(REPO, "tests", "golden")
# no SMPL-derived base data exists offline: the fixtures and tests run on the synthetic template (an explicit opt-in)
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the in-tree HIP library exists and is current (hipcc cross-compiles without a GPU).  This builds the
    product, it is not a fallback: if hipcc is missing the tests that need the library fail loudly."""
    try:
        from pmce_amd import build
        build.build()
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] could not build libpmce_hip.so: {e}")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(osp.join(GOLDEN, name))
    return load


_SD_CACHE = {}


def cached_state_dict(J, C, depth=3, seed=123):
    """412 MB of deterministic weights; cached per (J,C) for the session."""
    from pmce_amd import synth
    key = (J, C, depth, seed)
    if key not in _SD_CACHE:
        _SD_CACHE.clear()            # keep at most one resident (CPU RAM)
        _SD_CACHE[key] = synth.make_state_dict(synth.pmce_spec(J, C, depth), seed=seed)
    return _SD_CACHE[key]
