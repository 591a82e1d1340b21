import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def fixture_720p():
    """The reference's own 1280x720 test frames (tests/data/raw_p010_image.p010, raw_yuv420_image.yuv420)."""
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    p010 = np.fromfile(os.path.join(g, "raw_p010_image.p010"), np.uint16)
    yuv = np.fromfile(os.path.join(g, "raw_yuv420_image.yuv420"), np.uint8)
    return p010, yuv, 1280, 720


@pytest.fixture(scope="session")
def hip():
    """The product library, initialised on cuda:0.  Fails (does not skip) when the .so is missing."""
    import torch
    from libultrahdr_dev_amd import api
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    torch.cuda.set_device(0)
    api.init(0)
    return api
