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


def load_gray(path):
    from PIL import Image

    im = Image.open(path)
    assert im.mode == "L"
    return np.asarray(im).copy()


@pytest.fixture(scope="session")
def kitti02():
    return [load_gray(os.path.join(GOLDEN, "kitti02", "%06d.png" % i)) for i in range(4)]


@pytest.fixture(scope="session")
def kitti09():
    return [load_gray(os.path.join(GOLDEN, "kitti09", "%06d.png" % i)) for i in range(2)]


@pytest.fixture(scope="session")
def K02():
    return np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
