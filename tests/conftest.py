import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rank_valid():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "rank_valid.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def rank_bundles():
    from manual_yolo_amd.ckpt import load_bundle
    return {t: load_bundle(os.path.join(GOLDEN, f"rank_{t}.safetensors")) for t in ("best", "last")}
