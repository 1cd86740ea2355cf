import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "llama-nuts-and-bolts_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    """a GPU box has the KFD device node and a render node; no library is loaded here (importing torch costs a fresh box one to
    two minutes, and loading a HIP runtime before torch would decide which of the two copies in the image the process uses)"""
    import glob
    return os.path.exists("/dev/kfd") and bool(glob.glob("/dev/dri/renderD*"))


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
