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


@pytest.fixture(scope="session")
def butterfly_bgra():
    from PIL import Image
    im = np.array(Image.open(os.path.join(GOLDEN, "butterfly.png")))
    assert im.shape == (340, 512, 4)
    return np.ascontiguousarray(im[..., [2, 1, 0, 3]])      # RGBA -> BGRA (bgra8Unorm texture)


@pytest.fixture(scope="session")
def ipol():
    return np.load(os.path.join(GOLDEN, "butterfly_ipol.npz"))


@pytest.fixture(scope="session")
def butterfly_oracle(butterfly_bgra):
    """Oracle run on butterfly.png with the reference's default 7 octaves."""
    from oracle import pyoracle
    orc = pyoracle.Oracle(512, 340, n_octaves=7, nspo=3)
    res = orc.run(butterfly_bgra, want_float=True)
    return orc, res
