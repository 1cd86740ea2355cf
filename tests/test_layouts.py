"""HBM layout functions shared by the re-tilers and the kernels (csrc/lnb_device.h: tiled_index, m16_index, xt_index), checked on the host:
bijections onto their buffers, the 16-byte-unit / 1-KiB-per-wave-load structure DESIGN.md 4 and 5.11 describe, and the k order a matrix-core
lane finds in a unit (k = 128C + 16e + 4m + kk: the reference's k-ascending chain, src/ml/operations_lineartransform.go:46-65)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_index_functions_on_the_host(tmp_path):
    exe = str(tmp_path / "layout_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "native", "layout_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "layout_test: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_header_and_library_agree_on_the_batch_limit():
    """include/lnb.h documents the batch size the library accepts (LNB_BATCH_MAX in csrc/lnb_device.h)"""
    import re
    dev = open(os.path.join(ROOT, "llama-nuts-and-bolts_amd", "csrc", "lnb_device.h")).read()
    n = int(re.search(r"constexpr int LNB_BATCH_MAX = (\d+);", dev).group(1))
    hdr = open(os.path.join(ROOT, "include", "lnb.h")).read()
    assert "1..%d contexts" % n in hdr
    py = open(os.path.join(ROOT, "llama-nuts-and-bolts_amd", "lnb.py")).read()
    assert "1..%d InferenceContexts" % n in py
