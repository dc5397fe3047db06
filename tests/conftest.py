import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
SEED0 = 0x48474930   # seeds are SEED0 + config index


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts a throughput floor on the device (deselect with -m 'gpu and not perf' on a shared or throttled box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small():
    return dict(np.load(os.path.join(GOLDEN_DIR, "small_cases.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def lena():
    a = np.fromfile(os.path.join(GOLDEN_DIR, "lena_256.u8"), dtype=np.uint8)
    return a.reshape(256, 256)


@pytest.fixture(scope="session")
def fullhd():
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN_DIR, "fullhd_luma.png")))


@pytest.fixture(scope="session")
def fullhd709():
    """C1's input as image-0.19's to_luma would take it from PIL's RGB decode (truncating BT.709; make_golden.py)."""
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN_DIR, "fullhd_luma709.png")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import hgi_oracle
    hgi_oracle.build()
    return hgi_oracle
