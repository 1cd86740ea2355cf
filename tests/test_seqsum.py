"""Fuzz of csrc/lnb_seqsum.h (exact parity-map evaluation of the reference's sequential f32 sum of squares,
src/ml/operations_impl.go:236-251) against the plain sequential loop, on the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "native", "libseqsum_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "native", "seqsum_host.cpp")
    hdr = os.path.join(HERE, "..", "llama-nuts-and-bolts_amd", "csrc", "lnb_seqsum.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", SO])
    L = C.CDLL(SO)
    L.seqsum_ref.restype = C.c_float; L.seqsum_scan.restype = C.c_float
    L.seqsum_ref.argtypes = [C.c_void_p, C.c_int]; L.seqsum_scan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.seqsum_leaf_mismatches.restype = C.c_long
    L.seqsum_items.restype = C.c_float
    L.seqsum_items.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.seqsum_tree.restype = C.c_float
    L.seqsum_tree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def bf16(a):
    return ((a.astype(np.float32).view(np.uint32) >> 16) << 16).view(np.float32)


def test_scan_is_bit_identical_to_the_sequential_sum(lib):
    rng = np.random.default_rng(1)
    fast = []
    for trial in range(3000):
        kind, K = trial % 7, 4096
        if kind == 0: x = bf16(rng.standard_normal(K))
        elif kind == 1: x = bf16(rng.standard_normal(K) * 10 ** rng.uniform(-18, 6))
        elif kind == 2: x = bf16(2.0 ** rng.integers(-12, 4, K))                      # powers of two: ties everywhere
        elif kind == 3: x = bf16(rng.standard_normal(K) * (rng.random(K) < 0.05))      # mostly zeros
        elif kind == 4: x = bf16(np.exp(rng.uniform(-40, 5, K)))                       # subnormal squares .. large
        elif kind == 5: x = bf16(np.full(K, 2.0 ** rng.integers(-8, 8)))
        else: x = bf16(np.abs(rng.standard_normal(K)) * np.where(rng.random(K) < 0.01, 1e4, 1.0))
        p = np.ascontiguousarray((x.astype(np.float64) ** 2).astype(np.float32))
        nf = C.c_int(0)
        r = lib.seqsum_ref(p.ctypes.data, K)
        s = lib.seqsum_scan(p.ctypes.data, K, 64, C.byref(nf))
        assert np.float32(r).view(np.uint32) == np.float32(s).view(np.uint32), (trial, kind)
        fast.append(nf.value)
    assert np.mean(fast) > 40          # most blocks take the verified integer path
    for K, nb in [(256, 64), (8192, 64), (4096, 32), (64, 64), (1024, 16)]:
        for _ in range(200):
            x = bf16(rng.standard_normal(K) * 10 ** rng.uniform(-3, 3))
            p = np.ascontiguousarray((x.astype(np.float64) ** 2).astype(np.float32))
            assert np.float32(lib.seqsum_ref(p.ctypes.data, K)).view(np.uint32) == \
                np.float32(lib.seqsum_scan(p.ctypes.data, K, nb, None)).view(np.uint32)


def _cases(rng, K, trial):
    kind = trial % 8
    if kind == 0: x = bf16(rng.standard_normal(K))
    elif kind == 1: x = bf16(rng.standard_normal(K) * 10 ** rng.uniform(-18, 6))
    elif kind == 2: x = bf16(2.0 ** rng.integers(-12, 4, K))                      # powers of two: ties everywhere
    elif kind == 3: x = bf16(rng.standard_normal(K) * (rng.random(K) < 0.05))      # mostly zeros
    elif kind == 4: x = bf16(np.exp(rng.uniform(-40, 5, K)))                       # subnormal squares .. large
    elif kind == 5: x = bf16(np.full(K, 2.0 ** rng.integers(-8, 8)))
    elif kind == 6: x = bf16(np.abs(rng.standard_normal(K)) * np.where(rng.random(K) < 0.01, 1e4, 1.0))
    else: x = bf16(np.where(np.arange(K) < rng.integers(0, K), 0.0, rng.standard_normal(K)))   # leading zeros
    return np.ascontiguousarray((x.astype(np.float64) ** 2).astype(np.float32))


def test_tree_walk_is_bit_identical_to_the_sequential_sum(lib):
    """The multi-wave form used by the RMSNorm prologue (rms_fold / rms_scale_wide): one leaf per folding lane,
    segmented scan per wave, walker over the item mask."""
    rng = np.random.default_rng(7)
    visits, raws = [], []
    for trial in range(2400):
        K = [4096, 4096, 8192, 256, 512, 1024, 64, 3072, 5120, 8][trial % 10] if trial >= 800 else 4096
        NH = [7, 6, 2][trial % 3]
        p = _cases(rng, K, trial)
        nv, nr = C.c_int(0), C.c_int(0)
        r = lib.seqsum_ref(p.ctypes.data, K)
        s = lib.seqsum_tree(p.ctypes.data, K, NH, 256, C.byref(nv), C.byref(nr))
        assert np.float32(r).view(np.uint32) == np.float32(s).view(np.uint32), (trial, K, NH)
        if K == 4096 and trial % 8 == 0:
            visits.append(nv.value); raws.append(nr.value)
    # gaussian activations: the serial part is a few dozen node visits instead of 4096 dependent adds
    assert np.mean(visits) < 40 and np.mean(raws) < 12, (np.mean(visits), np.mean(raws))
    # the device's two-sums leaf evaluation (seq_leaf) never disagreed with the term-by-term integer evaluation (seq_leaf_steps)
    assert lib.seqsum_leaf_mismatches() == 0
    print("items per row %.1f, replayed leaves per row %.1f" % (np.mean(visits), np.mean(raws)))


def test_item_list_walk_is_bit_identical_and_rarely_falls_back(lib):
    """Round 4: the branch-free walk over the row's item list (runs + split crossing leaves, every check OR-ed, fallback to the old walk).
    Bit-identical to the sequential sum for every input family; for gaussian activations -- what the model feeds it -- the list suffices
    for nearly every row (the fallback is the slow path, not a wrong one)."""
    rng = np.random.default_rng(23)
    fast_gauss, items = [], []
    for trial in range(3200):
        K = [4096, 4096, 8192, 256, 512, 1024, 64, 3072, 5120, 8][trial % 10] if trial >= 1600 else 4096
        NH = [4, 3, 2][trial % 3]              # the device folds on at most four waves (rms_nf)
        p = _cases(rng, K, trial)
        fast, ni = C.c_int(0), C.c_int(0)
        r = lib.seqsum_ref(p.ctypes.data, K)
        s = lib.seqsum_items(p.ctypes.data, K, NH, 256, C.byref(fast), C.byref(ni))
        assert np.float32(r).view(np.uint32) == np.float32(s).view(np.uint32), (trial, K, NH)
        if K == 4096 and trial % 8 == 0 and NH != 2:
            fast_gauss.append(fast.value); items.append(ni.value)
    print("gaussian rows: list sufficient for %.1f %%, %.1f items per row" % (100 * np.mean(fast_gauss), np.mean(items)))
    assert np.mean(fast_gauss) > 0.95 and np.mean(items) < 40
    # outlier channels (one term lifts the sum several binades -- the massive activations of trained Llama checkpoints): still the list
    fast_out = []
    for trial in range(200):
        x = rng.standard_normal(4096).astype(np.float32) * np.float32(0.05)
        for pos in rng.integers(300, 4096, size=1 + trial % 3):
            x[pos] = np.float32(rng.uniform(20, 400))
        p = (x * x).astype(np.float32)
        fast = C.c_int(0)
        r = lib.seqsum_ref(p.ctypes.data, 4096)
        s = lib.seqsum_items(p.ctypes.data, 4096, 4, 256, C.byref(fast), None)
        assert np.float32(r).view(np.uint32) == np.float32(s).view(np.uint32), trial
        fast_out.append(fast.value)
    print("rows with outlier channels: list sufficient for %.1f %%" % (100 * np.mean(fast_out)))
    assert np.mean(fast_out) > 0.9
