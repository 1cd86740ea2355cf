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
    """a visible HIP device (asked of the HIP runtime directly: importing torch first costs a fresh box one to two minutes)"""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        try:
            import torch
            return torch.cuda.is_available()
        except Exception:
            return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
