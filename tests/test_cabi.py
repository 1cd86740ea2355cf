"""CPU-side checks of the drop-in boundary: liblnb_hip.so builds for gfx950, loads without a GPU, exports every
symbol include/lnb.h declares, and fails LOUDLY (no CPU fallback) when no MI355X is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    return _lnb


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lnb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lnb_[a-z0-9_]+)\s*\(", src)) - {"lnb_layer_cb"})


def test_header_symbols_are_exported(lnb):
    L = C.CDLL(os.path.join(ROOT, "llama-nuts-and-bolts_amd", "liblnb_hip.so"))
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert set(names) == set(lnb.EXPORTS)


def test_no_cpu_fallback(lnb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lnb.LnbError):
        lnb.LlamaTransformer(dim=256, n_layers=1, n_heads=4, n_kv_heads=2, vocab_size=64, multiple_of=64)
    import numpy as np
    with pytest.raises(lnb.LnbError):
        lnb.op_linear(np.zeros((1, 8), dtype=np.uint16), np.zeros((4, 8), dtype=np.uint16))


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "llama-nuts-and-bolts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".go", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in txt.replace("oracle/lnb_oracle.c; spec", "") or f == "lnb_device.h", (dirpath, f)
                assert "import oracle" not in txt and "from oracle" not in txt, (dirpath, f)


def test_ffn_hidden_dim_matches_reference_formula(lnb):
    a = lnb.ModelArgs(**lnb.LLAMA_8B)
    assert lnb.lib().lnb_model_ffn_hidden_dim(C.byref(a)) == 14336            # llamatransformer.go:569-577


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu(lnb):
    """llama-nuts-and-bolts_amd/host/lnb_host.hpp (C++ mirror of the Go API) links against the C ABI; on a box
    without an MI355X NewLlamaTransformer must return the library's error, not fall back."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "tests", "native", "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "llama-nuts-and-bolts_amd"), "-llnb_hip", "-Wl,-rpath," + os.path.join(ROOT, "llama-nuts-and-bolts_amd"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_parity.py")
    r = subprocess.run([exe, "40", "1", "2", "3"], capture_output=True, text=True)
    assert r.returncode == 3 and "error:" in r.stdout


def test_abi_version_is_one_number_everywhere(lnb):
    """include/lnb.h, the library and the ctypes binding agree on LNB_ABI_VERSION (a changed signature must not reach a stale binding);
    lnb.lib() itself refuses a library that reports another number"""
    hdr = open(os.path.join(ROOT, "include", "lnb.h")).read()
    v = int(re.search(r"#define\s+LNB_ABI_VERSION\s+(\d+)", hdr).group(1))
    L = C.CDLL(os.path.join(ROOT, "llama-nuts-and-bolts_amd", "liblnb_hip.so"))
    assert L.lnb_abi_version() == v == lnb.ABI_VERSION
    go = open(os.path.join(ROOT, "llama-nuts-and-bolts_amd", "go", "llamatransformer_hip.go")).read()
    assert "C.lnb_abi_version()" in go and "C.LNB_ABI_VERSION" in go


def test_load_time_queue_default_is_opt_out_and_recorded():
    """the library exports GPU_MAX_HW_QUEUES=16 while it is loaded unless the host set a value or LNB_KEEP_HW_QUEUES=1"""
    import subprocess
    import sys
    so = os.path.join(ROOT, "llama-nuts-and-bolts_amd", "liblnb_hip.so")
    code = "import ctypes, os; ctypes.CDLL(%r); print(os.environ.get('GPU_MAX_HW_QUEUES', 'unset'))" % so
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "LNB_KEEP_HW_QUEUES")}
    # (os.environ is a snapshot taken at interpreter start: ask libc)
    code = ("import ctypes; ctypes.CDLL(%r); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "v = libc.getenv(b'GPU_MAX_HW_QUEUES'); print(v.decode() if v else 'unset')" % so)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "16"
    assert subprocess.run([sys.executable, "-c", code], env=dict(env, LNB_KEEP_HW_QUEUES="1"), capture_output=True, text=True).stdout.strip() == "unset"
    assert subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="8"), capture_output=True, text=True).stdout.strip() == "8"
