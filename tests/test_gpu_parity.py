"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(llama-nuts-and-bolts_amd/liblnb_hip.so), against the CPU oracle on the same seeded inputs.

Bar (north_star): argmax token ids bit-exact, logits within 1e-2; the exact-order kernels are in fact
expected to be BIT-EXACT on every intermediate, so the tests assert equality of the raw bf16 bits and
only fall back to the 1e-2 tolerance where a libm difference (f64 exp / cos / sin last ulp) could
legitimately surface.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TINY = dict(orc.TINY)
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    assert _lnb.device_count() >= 1
    return _lnb


def bf(a):
    return orc.f32_to_bf16(np.asarray(a, dtype=np.float32))


def orc_linear(x, w, nthreads=8):
    y = np.zeros((x.shape[0], w.shape[0]), dtype=np.uint16)
    orc.lib().orc_linear_bf16(orc._p(x), orc._p(w), orc._p(y), x.shape[0], w.shape[0], x.shape[1], nthreads)
    return y


def test_linear_reference_kat_on_gpu(lnb):
    # src/ml/operations_test.go:831-878 (K=3 zero-padded to 8: adding +0 products is exact)
    k = KAT["linear_bf16"]
    w = np.zeros((4, 8), dtype=np.uint16); x = np.zeros((2, 8), dtype=np.uint16)
    w[:, :3] = bf(k["weights"]); x[:, :3] = bf(k["input"])
    for rw in (16, 32, 64):
        y = lnb.op_linear(x, w, rw=rw)
        assert np.abs(orc.bf16_to_f32(y) - np.array(k["expected"], dtype=np.float32)).max() < 1e-3
        assert (y == orc_linear(x, w)).all()


@pytest.mark.parametrize("rows,n,k,rw", [
    (1, 256, 256, 16), (1, 256, 256, 32), (1, 256, 256, 64),
    (3, 100, 896, 16), (3, 100, 896, 32), (3, 100, 896, 64),      # ragged N, K not a multiple of the stage
    (1, 64, 8, 16), (5, 17, 40, 64),                               # tiny / single chunk
    (1, 4096, 4096, 16), (1, 4096, 14336, 16), (1, 6144, 4096, 16), (2, 4096, 4096, 64),
    # rw 4 = the row-broadcast kernel (wo / w2): one chunk, ragged N (not a multiple of 4 / 16), more blocks than CUs, model shapes
    (1, 16, 128, 4), (3, 100, 896, 4), (2, 5000, 256, 4), (1, 4096, 4096, 4), (1, 4096, 14336, 4), (4, 50, 1536, 4),
    (1, 96, 28672, 32), (2, 40, 28672, 16),                       # 70B-like w2 rows: the four-helper configuration for very long K
    # rw 4 with K a multiple of 512 = rowcast_lds_kernel (helper-fed chain waves): one stage, stage counts 2, 3 (the ring's unroll), 4, 5;
    # ragged N; more 16-row blocks than CUs (persistent blocks); several x rows; the longest K it accepts
    (1, 16, 512, 4), (1, 64, 1024, 4), (3, 100, 1536, 4), (1, 37, 2048, 4), (2, 24, 2560, 4), (2, 5000, 512, 4), (1, 8200, 1024, 4), (1, 48, 16384, 4),
    # rw 24 = gemv_quad_kernel (quad-DPP chain waves fed through the LDS; K a multiple of its 256-step stages): one / two / three / many
    # stages, ragged N (the second chain wave's shadow lanes, a partial last block), more blocks than CUs, several x rows, the 8B wq|wk|wv
    (1, 24, 256, 24), (1, 48, 512, 24), (3, 100, 768, 24), (1, 17, 1024, 24), (2, 7000, 512, 24), (1, 6144, 4096, 24), (2, 50, 8192, 24),
])
def test_linear_bit_exact(lnb, rows, n, k, rw):
    rng = np.random.default_rng(rows * 1000003 + n * 101 + k + rw)
    x = bf(rng.standard_normal((rows, k)))
    w = bf(rng.standard_normal((n, k)) * 0.05)
    y = lnb.op_linear(x, w, rw=rw)
    assert (y == orc_linear(x, w)).all()


@pytest.mark.parametrize("rows,n,k,rw", [
    (16, 64, 128, 16), (17, 100, 896, 32), (128, 256, 256, 64), (130, 4096, 4096, 16), (33, 50, 1536, 4), (128, 4096, 4096, 4),
    (64, 300, 8, 64), (200, 96, 14336, 4),
])
def test_prefill_gemm_on_the_matrix_cores_is_bit_exact(lnb, rows, n, k, rw):
    """16 or more rows go through gemm_mfma_kernel: v_mfma_f32_16x16x4_f32 is the same k-ordered f32 chain (tools/mfma_exact.hip),
    so the outputs must still be bit-identical to the oracle -- for every weight layout, ragged M / N, K not a multiple of the
    128-step slab."""
    rng = np.random.default_rng(rows * 7 + n + k + rw)
    x = bf(rng.standard_normal((rows, k)) * 10 ** rng.uniform(-2, 2))
    w = bf(rng.standard_normal((n, k)) * 0.05)
    y = lnb.op_linear(x, w, rw=rw)
    assert (y == orc_linear(x, w)).all()


def test_prefill_rmsnorm_rows_then_gemm_is_bit_exact(lnb):
    rng = np.random.default_rng(3)
    rows, n, k = 40, 192, 4096
    x = bf(rng.standard_normal((rows, k)) * np.exp(rng.uniform(-6, 6, (rows, 1))))
    nw = bf(1 + 0.1 * rng.standard_normal(k))
    w = bf(rng.standard_normal((n, k)) * 0.05)
    y = lnb.op_rmsnorm_linear(x, nw, 1e-5, w, rw=32)
    xn = np.zeros_like(x)
    orc.lib().orc_rmsnorm_bf16(orc._p(x), orc._p(nw), orc._p(xn), rows, k, np.float32(1e-5), None)
    assert (y == orc_linear(xn, w)).all()


def _same_bits_or_both_nan(a, b):
    """NaN-ness must agree; a NaN's sign and payload are left open (the reference does not define them either: Go's float32 arithmetic
    returns whatever the host FPU propagates -- x86 SSE produces the negative default NaN for inf - inf, arm64 and gfx950 the positive one)"""
    fa, fb = orc.bf16_to_f32(a), orc.bf16_to_f32(b)
    return bool(((a == b) | (np.isnan(fa) & np.isnan(fb))).all())


@pytest.mark.parametrize("rows,rw", [(1, 16), (1, 64), (1, 4), (3, 32), (20, 16), (20, 4), (130, 64), (1, 24), (2, 24)])
def test_linear_special_values(lnb, rows, rw):
    """Values outside the comfortable range, through every exact kernel family (chain GEMV, row-broadcast GEMV, f32 matrix-core GEMM):
    signed zeros (the chain starts at +0, so a sum of -0 products is +0), +-inf and NaN in weights and activations (inf - inf, 0 * inf),
    products that overflow f32, products and partial sums in the f32-subnormal range (bf16 shares f32's exponent range: two tiny
    operands give a subnormal, inexactly rounded product -- operations_lineartransform.go:60 multiplies in float32), and sums that
    cancel to exactly zero.  Bits must match the oracle; for NaN results only the NaN-ness.
    The GEMVs multiply, then add, like the reference, and match everywhere.  The f32 matrix-core instruction of the prefill GEMM
    (16 or more rows) FUSES the multiply: same bits whenever every single product is a normal f32 (NOTES.md section 2) -- its rows
    here keep sums that overflow and subnormal partial sums, but no single product outside [2^-126, 2^128)."""
    k, n = 512, 96
    rng = np.random.default_rng(rows * 131 + rw)
    x = bf(rng.standard_normal((rows, k)))
    w = bf(rng.standard_normal((n, k)) * 0.05)
    xf, wf = orc.bf16_to_f32(x).copy(), orc.bf16_to_f32(w).copy()
    wf[0, :] = -0.0                                              # all products -0 (or +0): result +0
    wf[1, :] = 0.0; wf[1, 7] = np.inf                            # one inf product
    wf[2, 3] = np.inf; wf[2, 200] = -np.inf                      # inf - inf -> NaN
    wf[3, 11] = np.nan
    gemm = rows >= 16
    wf[4, :] = 3.0e38                                            # single products overflow (GEMV rows) ...
    if gemm:
        wf[4, :] = 0.0; wf[4, :64] = 1.0e37                      # ... or only the running SUM does (x[:, :64] = 2 below: 64 x 2e37)
    wf[5, :] = 1e-30 * rng.standard_normal(k)                    # products ~1e-30 x 1: tiny but normal
    wf[6, :] = 1e-38 * rng.standard_normal(k)                    # bf16 values at the edge of / inside the subnormal range
    wf[7, :] = np.float32(2.0 ** -100) * rng.integers(1, 128, k)      # with the tiny activations below: subnormal products
    wf[8, 0::2] = 1.0; wf[8, 1::2] = -1.0                        # exact cancellation on duplicated activations
    wf[9, :] = 0.0; wf[9, 100] = np.inf                          # inf * 0 -> NaN (activation 100 is zeroed below)
    wf[10, :] = -wf[5, :]
    xf[:, 100] = 0.0
    xf[:, 1::2] = xf[:, 0::2]
    if gemm:
        xf[:, :64] = 2.0
    if rows > 1:
        # tiny activations: with weight row 7 (2^-100 x small integers) the products are 2^-140 x integers -- exact subnormals in both
        # arithmetics, partial sums subnormal; with weight row 6 (1e-38) single products underflow inexactly: GEMV rows only
        xf[1, :] = np.float32(2.0 ** -40) * rng.integers(-127, 128, k)
        if gemm:
            wf[6, :] = 1e-20 * rng.standard_normal(k)
    if rows > 2:
        xf[2, 17] = np.nan; xf[2, 18] = -np.inf
    x, w = bf(xf), bf(wf)
    y = lnb.op_linear(x, w, rw=rw)
    ref = orc_linear(x, w)
    bad = np.argwhere(~((y == ref) | (np.isnan(orc.bf16_to_f32(y)) & np.isnan(orc.bf16_to_f32(ref)))))
    assert bad.size == 0, "first differences (row, col): %s  got %s  oracle %s" % (bad[:6].tolist(), [hex(int(y[i, j])) for i, j in bad[:6]], [hex(int(ref[i, j])) for i, j in bad[:6]])
    assert y[0, 0] == 0x0000                                     # +0, not -0


def test_linear_lm_head_shape(lnb):
    rng = np.random.default_rng(7)
    x = bf(rng.standard_normal((1, 4096)))
    w = bf(rng.standard_normal((128256, 4096)) * 0.02)
    y = lnb.op_linear(x, w, rw=64)
    assert (y == orc_linear(x, w)).all()


def test_linear_order_sensitivity_guard(lnb):
    # a vector whose sequential f32 sum differs from any pairwise/blocked sum: catches split-K kernels
    k = 4096
    x = bf(np.ones((1, k)))
    wrow = np.full(k, 2.0 ** -12, dtype=np.float32); wrow[0] = 1.0
    w = bf(np.tile(wrow, (64, 1)))
    y = lnb.op_linear(x, w, rw=16)
    assert (y == orc_linear(x, w)).all()


@pytest.mark.parametrize("rows,n,k,rw", [(1, 256, 256, 16), (4, 96, 512, 32), (1, 6144, 4096, 16), (2, 1024, 4096, 64),
                                         (1, 48, 256, 24), (3, 100, 1024, 24), (1, 6144, 4096, 24)])
def test_rmsnorm_linear_bit_exact(lnb, rows, n, k, rw):
    rng = np.random.default_rng(n + k + rw)
    x = bf(rng.standard_normal((rows, k)) * 3.0)
    nw = bf(1 + 0.1 * rng.standard_normal(k))
    w = bf(rng.standard_normal((n, k)) * 0.05)
    y = lnb.op_rmsnorm_linear(x, nw, 1e-5, w, rw=rw)
    xn = np.zeros_like(x)
    orc.lib().orc_rmsnorm_bf16(orc._p(x), orc._p(nw), orc._p(xn), rows, k, np.float32(1e-5), None)
    assert (y == orc_linear(xn, w)).all()


@pytest.mark.parametrize("k,rw", [(4096, 16), (4096, 64), (8192, 32), (512, 16), (3072, 64), (64, 16), (4096, 24), (512, 24)])
def test_rmsnorm_exact_parallel_sum_adversarial(lnb, k, rw):
    """The RMSNorm sum of squares is evaluated by the exact parity-map tree (rms_scale_wide); inputs chosen to stress it:
    huge dynamic range, ties everywhere (powers of two), sparse rows, subnormal squares, outliers, leading zeros."""
    rng = np.random.default_rng(k + rw)
    rows = []
    for kind in range(12):
        if kind == 0: x = rng.standard_normal(k)
        elif kind == 1: x = rng.standard_normal(k) * 1e-15
        elif kind == 2: x = 2.0 ** rng.integers(-12, 4, k)
        elif kind == 3: x = rng.standard_normal(k) * (rng.random(k) < 0.05)
        elif kind == 4: x = np.exp(rng.uniform(-40, 5, k))
        elif kind == 5: x = np.full(k, 2.0 ** rng.integers(-8, 8))
        elif kind == 6: x = np.abs(rng.standard_normal(k)) * np.where(rng.random(k) < 0.01, 1e4, 1.0)
        elif kind == 7: x = np.where(np.arange(k) < k // 3, 0.0, rng.standard_normal(k))
        elif kind == 8: x = np.zeros(k)
        elif kind == 9: x = rng.standard_normal(k) * 1e12
        elif kind == 10: x = np.where(np.arange(k) == k - 1, 3e4, 1e-3 * rng.standard_normal(k))
        else: x = rng.standard_normal(k) * np.exp(rng.uniform(-20, 8, k))
        rows.append(x)
    x = bf(np.stack(rows))
    nw = bf(1 + 0.1 * rng.standard_normal(k))
    w = bf(rng.standard_normal((64, k)) * 0.05)
    y = lnb.op_rmsnorm_linear(x, nw, 1e-5, w, rw=rw)
    xn = np.zeros_like(x)
    orc.lib().orc_rmsnorm_bf16(orc._p(x), orc._p(nw), orc._p(xn), x.shape[0], k, np.float32(1e-5), None)
    assert (y == orc_linear(xn, w)).all()


@pytest.mark.parametrize("k,rw", [(4096, 24), (4096, 16), (2048, 32)])
def test_rmsnorm_item_list_walk_many_rows(lnb, k, rw):
    """Round 4: the norm sum is walked as a list of items (runs of leaves + leaves split at a binade crossing), every check OR-ed, with
    the old record walk as the fallback (a crossing too close to call: ~2.5 % of gaussian rows).  384 rows of varied scale, with and
    without outlier channels, hit both paths; the bits must be the oracle's either way."""
    rng = np.random.default_rng(k * 3 + rw)
    nw = bf(1 + 0.1 * rng.standard_normal(k))
    w = bf(rng.standard_normal((48, k)) * 0.05)
    for chunk in range(48):
        x = rng.standard_normal((8, k)) * np.exp(rng.uniform(-6, 6, (8, 1)))
        if chunk % 3 == 1:
            for r in range(8): x[r, rng.integers(0, k, 1 + r % 4)] *= rng.uniform(50, 3000)
        if chunk % 3 == 2: x *= (rng.random((8, k)) < 0.3)
        x = bf(x)
        y = lnb.op_rmsnorm_linear(x, nw, 1e-5, w, rw=rw)
        xn = np.zeros_like(x)
        orc.lib().orc_rmsnorm_bf16(orc._p(x), orc._p(nw), orc._p(xn), 8, k, np.float32(1e-5), None)
        assert (y == orc_linear(xn, w)).all(), chunk


@pytest.mark.parametrize("k,rw,reps", [(4096, 32, 1), (512, 16, 1), (4096, 64, 1), (4096, 32, 2), (4096, 24, 1), (1024, 24, 2)])
def test_rmsnorm_rows_with_non_finite_and_overflowing_squares(lnb, k, rw, reps):
    """Rows the parity-map evaluation of the norm sum was not designed around: inf / NaN activations, squares that overflow f32
    (|x| > 1.8e19), a running sum that overflows after a few terms, squares that underflow to zero.  The reference just keeps adding
    (sum = inf -> mean = inf -> 1/sqrt = 0 -> x * 0, or NaN): the device must land on the same bits (NaN-ness for NaNs).
    reps = 2: 20 rows, i.e. the prefill form (rmsnorm_rows_kernel + matrix-core GEMM)."""
    rng = np.random.default_rng(k + rw + reps)
    base = rng.standard_normal(k).astype(np.float32)
    rows = []
    for kind in range(10):
        x = base.copy()
        if kind == 0: x[k // 2] = np.inf
        elif kind == 1: x[k // 3] = np.nan
        elif kind == 2: x[5] = -np.inf
        elif kind == 3: x[k // 2: k // 2 + 9] = 1e20                      # squares overflow
        elif kind == 4: x[k - 1] = -3e38
        elif kind == 5: x[:] = 3e38 * np.sign(base)
        elif kind == 6: x[:] = 1e19                                       # squares 1e38: the running sum overflows at the fourth term
        elif kind == 7: x[0] = np.nan
        elif kind == 8: x[:] = 0.0; x[0] = np.inf
        else: x[:] = 1e-30 * base                                         # squares underflow to zero: mean = 0, scale = 1/sqrt(eps)
        rows.append(x)
    x = bf(np.stack(rows * reps))
    nw = bf(1 + 0.1 * rng.standard_normal(k))
    w = bf(rng.standard_normal((48, k)) * 0.05)
    y = lnb.op_rmsnorm_linear(x, nw, 1e-5, w, rw=rw)
    xn = np.zeros_like(x)
    orc.lib().orc_rmsnorm_bf16(orc._p(x), orc._p(nw), orc._p(xn), x.shape[0], k, np.float32(1e-5), None)
    ref = orc_linear(xn, w)
    bad = np.argwhere(~((y == ref) | (np.isnan(orc.bf16_to_f32(y)) & np.isnan(orc.bf16_to_f32(ref)))))
    assert bad.size == 0, "rows that differ: %s; first: got %s oracle %s" % (sorted(set(int(i) for i, _ in bad)), [hex(int(y[i, j])) for i, j in bad[:6]], [hex(int(ref[i, j])) for i, j in bad[:6]])


@pytest.fixture(scope="module")
def tiny_pair(lnb):
    om = orc.Model(**TINY).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize()
    yield om, gm
    gm.close(); om.close()


def test_synthetic_weights_identical(lnb, tiny_pair):
    om, gm = tiny_pair
    for name in om.tensor_names():
        ref = om.get_tensor(name)
        got = gm.get_tensor(name, ref.size)
        assert (ref == got).all(), name


def test_set_tensor_roundtrip(lnb):
    gm = lnb.LlamaTransformer(**TINY)
    rng = np.random.default_rng(3)
    F = gm.ffn_hidden
    shapes = {"layers.1.attention.wk.weight": (128, 256), "layers.0.feed_forward.w3.weight": (F, 256),
              "layers.1.feed_forward.w2.weight": (256, F), "output.weight": (1024, 256), "norm.weight": (256,)}
    for name, shp in shapes.items():
        a = rng.integers(0, 65536, size=shp, dtype=np.uint16)
        gm.set_tensor(name, a)
        assert (gm.get_tensor(name, a.size) == a.reshape(-1)).all(), name
    with pytest.raises(lnb.LnbError):
        gm.set_tensor("output.weight", np.zeros((8, 8), dtype=np.uint16))          # loader.go:183-192 shape check
    with pytest.raises(lnb.LnbError):
        gm.set_tensor("no.such.tensor", np.zeros((8, 8), dtype=np.uint16))
    gm.close()


def test_device_exp_against_the_host_libm_exp_on_every_possible_softmax_input(lnb):
    """SURVEY 8(c) leaves `exp` unpinned between implementations (Go's math.Exp, glibc's, ocml's: each faithful, none the other).  Between the DEVICE (ocml's f64 exp, what every
    attention kernel evaluates or looks up) and the ORACLE (glibc's exp, what oracle/lnb_oracle.c calls -- and what CPython's math.exp calls in this image) the distance can be
    MEASURED exhaustively: a raw score is a bf16, so the softmax numerator exp(float64(trunc_bf16(s / sqrt(hd)))) (llamatransformer.go:464, operations_impl.go:498) is a function
    of 16 bits.  All 65536 inputs, raw f64 bits, divisor 1 (= the device's exp on every bf16 value) and the 8B shape's divisor 11.3125.  Measured on MI355X / ROCm 7.2 against
    glibc 2.35: 231 inputs (0.35 %) differ, every one by exactly ONE f64 ulp, none after narrowing to f32 -- and a numerator only ever leaves f64 as float32(e / Z)
    (operations_impl.go:506), where one f64 ulp moves the f32 rounding with probability ~2^-28: that is why the logits of every parity test, golden and bench run are the
    oracle's bit for bit.  The test pins that characterisation: a toolchain whose exp drifts further (or gets closer) shows up here, not as a one-in-a-billion token flip."""
    import math

    def host(divisor):
        out = np.empty(65536, dtype=np.float64)
        bits = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
        with np.errstate(all="ignore"):
            q = (bits / np.float32(divisor)).astype(np.float32)              # float32 division, round to nearest (DivToScalar on f32 :464)
        s16 = (q.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)   # truncate to bf16
        for i, v in enumerate(s16.astype(np.float64)):
            try:
                out[i] = math.exp(v)
            except OverflowError:
                out[i] = math.inf
        return out

    for divisor in (1.0, 11.3125):
        dev, ref = lnb.op_exp_table(divisor), host(divisor)
        nan_d, nan_r = np.isnan(dev), np.isnan(ref)
        assert (nan_d == nan_r).all()
        ok = ~nan_d
        assert (np.isinf(dev[ok]) == np.isinf(ref[ok])).all() and ((dev[ok] == 0) == (ref[ok] == 0)).all()            # same overflow / underflow points
        ulps = np.abs(dev.view(np.int64)[ok] - ref.view(np.int64)[ok])                                                 # (same sign, finite or both inf: the bit distance is the ulp distance)
        assert ulps.max() <= 1, (divisor, int(ulps.max()))
        assert int((ulps != 0).sum()) <= 400, (divisor, int((ulps != 0).sum()))                                        # measured: 231 at divisor 1
        with np.errstate(all="ignore"):
            assert (dev[ok].astype(np.float32).view(np.uint32) == ref[ok].astype(np.float32).view(np.uint32)).all()   # identical once narrowed to f32
        # ... and against the THIRD implementation, Go's portable math.Exp restated (oracle/go_exp.py; tests/test_exp_implementations.py ties it to the host libm):
        # the device is no further from what the reference calls than the reference's own platforms are from each other
        from oracle.go_exp import go_exp
        bits = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
        with np.errstate(all="ignore"):
            q = (bits / np.float32(divisor)).astype(np.float32)
        s16 = (q.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32).astype(np.float64)
        gox = np.array([go_exp(float(v)) for v in s16])
        ulps_go = np.abs(dev.view(np.int64)[ok] - gox.view(np.int64)[ok])
        assert ulps_go.max() <= 1 and int((ulps_go != 0).sum()) <= 1000, (divisor, int(ulps_go.max()), int((ulps_go != 0).sum()))
        with np.errstate(all="ignore"):
            assert (dev[ok].astype(np.float32).view(np.uint32) == gox[ok].astype(np.float32).view(np.uint32)).all()
        print("exp over 65536 bf16 inputs, divisor %g: device vs host libm %d differ, device vs Go's portable exp %d differ (all by one f64 ulp, none as f32)" % (divisor, int((ulps != 0).sum()), int((ulps_go != 0).sum())))


def test_rope_table_matches_oracle(lnb, tiny_pair):
    om, gm = tiny_pair
    a, b = om.rope_table(), gm.PrecomputedFreqsCis
    assert a.shape == b.shape == (4096, 32, 2)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_tiny_prefill_and_decode_bit_exact(lnb, tiny_pair):
    om, gm = tiny_pair
    toks = orc.synth_tokens(99, 12, TINY["vocab_size"])
    oc = orc.Context(om, 64); gc = lnb.InferenceContext(gm, 64)
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    assert lo.shape == lg.shape == (12, TINY["vocab_size"])           # logits for ALL rows (llamatransformer.go:170)
    assert np.abs(lo - lg).max() <= 1e-2
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all()
    assert ao == ag
    for layer in range(TINY["n_layers"]):
        assert (oc.cache(layer, 0) == gc.CacheK(layer)).all()
        assert (oc.cache(layer, 1) == gc.CacheV(layer)).all()
    tok, pos = ao, 12
    for _ in range(6):                                                 # one-token Forward steps (inference.go:194-202)
        lo, ao = oc.forward([tok], pos)
        lg, ag = gc.Forward([tok], pos)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
        tok, pos = ao, pos + 1
    # chunked prefill allowed by the reference's modulo mask broadcast: S=4 at start_pos=4 (T % S == 0)
    oc2 = orc.Context(om, 64); gc2 = lnb.InferenceContext(gm, 64)
    oc2.forward(toks[:4], 0); gc2.Forward(toks[:4], 0)
    lo, ao = oc2.forward(toks[4:8], 4); lg, ag = gc2.Forward(toks[4:8], 4)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    for c in (oc, oc2):
        c.close()
    for c in (gc, gc2):
        c.close()


def _eq_or_both_nan_f32(a, b):
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("rows,scale,expect_nan", [(12, 44.0, False), (12, 50.0, True), (20, 44.0, False), (20, 47.0, True)])
def test_softmax_without_max_subtraction_overflows_like_the_reference(lnb, rows, scale, expect_nan):
    """ml.Softmax (operations_impl.go:492-508) exponentiates the raw scores in f64 with NO max subtraction: scores beyond 709.78 give
    exp = +inf, Z = +inf, p = inf / inf = NaN for those positions and e / inf = 0 for the others, and the NaN spreads through PV and wo
    to the whole row.  A model whose layer-0 wq / wk are scaled up to the edge must give the same picture on the device: just below it
    (scores of several hundred, e up to 1e300, one-hot probabilities) the same bits; just above it NaN where the oracle has NaN and the
    same argmax (-1: ml.Argmax never selects a NaN).
    rows = 12: one-workgroup-per-head kernel; 20: matrix-core prefill attention; then decode steps through both decode forms (the
    certified softmax denominator of the long-context kernels sees Z = inf)."""
    om = orc.Model(**TINY).fill_synthetic(1234)
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234)
    for name in ("layers.0.attention.wq.weight", "layers.0.attention.wk.weight"):
        wt = orc.bf16_to_f32(om.get_tensor(name).copy()) * np.float32(scale)
        om.set_tensor(name, bf(wt)); gm.set_tensor(name, bf(wt).reshape(-1, TINY["dim"]))
    om.finalize(); gm.finalize()
    toks = orc.synth_tokens(7, rows, TINY["vocab_size"])
    oc = orc.Context(om, 64); gc = lnb.InferenceContext(gm, 64); gl = lnb.InferenceContext(gm, 64).set_attention(0, 0)
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    ll, al = gl.Forward(toks, 0)
    assert bool(np.isnan(lo).any()) == expect_nan, "the scale no longer sits on the intended side of exp's overflow"
    assert _eq_or_both_nan_f32(lo, lg) and _eq_or_both_nan_f32(lo, ll) and ao == ag == al
    for layer in range(TINY["n_layers"]):
        for which in (0, 1):
            ref, got = oc.cache(layer, which)[:rows], (gc.CacheK(layer) if which == 0 else gc.CacheV(layer))[:rows]
            assert _same_bits_or_both_nan(got, ref), (layer, which)
    tok, pos = (ao if ao >= 0 else 5), rows
    for _ in range(5):
        lo, ao = oc.forward([tok], pos)
        lg, ag = gc.Forward([tok], pos)
        ll, al = gl.Forward([tok], pos)
        assert _eq_or_both_nan_f32(lo, lg) and _eq_or_both_nan_f32(lo, ll) and ao == ag == al
        tok, pos = (ao if ao >= 0 else 5), pos + 1
    for c in (oc, gc, gl):
        c.close()
    gm.close(); om.close()


def test_contexts_on_concurrent_host_threads_share_one_model(lnb, tiny_pair):
    """a finalized lnb_model is immutable and shareable, one lnb_ctx per generation and per thread, calls may arrive on
    any OS thread (cgo).  Four host threads, one context each on the same model, run prefill + the captured greedy loop + eager steps
    at the same time (ctypes releases the GIL for the duration of a call); every thread must get the oracle's tokens for ITS prompt,
    and a failing call's message must be the calling thread's own (lnb_last_error is thread-local)."""
    import threading
    om, gm = tiny_pair
    n_threads, P, N = 4, 9, 24
    prompts = [orc.synth_tokens(200 + t, P + t, TINY["vocab_size"]) for t in range(n_threads)]
    refs = []
    for pr in prompts:
        ref, _ = orc.Context(om, 96).generate(pr, N + 1)
        refs.append([int(v) for v in ref])
    results, errors = [None] * n_threads, []
    start = threading.Barrier(n_threads)

    def work(t):
        try:
            pr = prompts[t]
            for rep in range(3):
                gc = lnb.InferenceContext(gm, 96)
                start.wait()
                _, first = gc.Forward(pr, 0, want_logits=(rep == 1))
                got, _ = gc.decode_greedy(first, len(pr), N // 2)
                toks = [first] + [int(v) for v in got]
                tok, pos = toks[-1], len(pr) + N // 2
                for _ in range(N - N // 2):                       # eager one-token steps on the same context
                    _, tok = gc.Forward([tok], pos, want_logits=False)
                    toks.append(int(tok)); pos += 1
                with pytest.raises(lnb.LnbError, match="empty token array"):
                    gc.Forward(np.zeros(0, dtype=np.int32), pos)
                gc.close()
                if results[t] is None:
                    results[t] = toks
                assert results[t] == toks
        except BaseException as e:                                 # surfaces in the main thread below
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for t in range(n_threads):
        assert results[t] == refs[t], "thread %d" % t


def _device_free_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_lifecycle_contexts_models_and_pipes_give_their_device_memory_back(lnb, tiny_pair):
    """A host creates one context per generation (inference.go:174) for as long as it runs: everything a context, a pipe or a model
    allocates -- KV caches, scratch, captured graphs of both attention forms and of the pipeline stage step, events, pinned words --
    must go when it is destroyed.  Also: a context stays usable after a refused call, and reset() replays the same tokens."""
    om, gm = tiny_pair
    prompt = orc.synth_tokens(31, 10, TINY["vocab_size"])
    ref, _ = orc.Context(om, 64).generate(prompt, 13)
    ref = [int(v) for v in ref]

    def one_generation(seq_len=64):
        gc = lnb.InferenceContext(gm, seq_len)
        _, first = gc.Forward(prompt, 0, want_logits=True)
        got, _ = gc.decode_greedy(first, len(prompt), 6)                    # captures the short-attention graph
        gc.set_attention(0, 0)
        got2, _ = gc.decode_greedy(int(got[-1]), len(prompt) + 6, 6)        # and the long-attention one
        toks = [first] + [int(v) for v in got] + [int(v) for v in got2]
        with pytest.raises(lnb.LnbError):                                   # beyond the context: refused, nothing enqueued
            gc.Forward(prompt, seq_len - 3)
        gc.reset(); gc.set_attention(512, 0)
        _, again = gc.Forward(prompt, 0, want_logits=False)
        assert again == first
        pipe = lnb.Pipeline(gm, 0, 1, None)
        pc = lnb.InferenceContext(gm, seq_len)
        slots = [pipe.tick(run=pc, run_rows=len(prompt), run_pos=0, run_tokens=np.ascontiguousarray(prompt, dtype=np.int32))]
        for i in range(4):
            slots.append(pipe.tick(run=pc, run_rows=1, run_pos=len(prompt) + i))
        pipe.sync()
        ptoks = [int(pipe.read_tokens(q, 1)[0]) for q in slots]
        pipe.close(); pc.close(); gc.close()
        return toks, ptoks

    toks, ptoks = one_generation()                                         # warm-up: code objects, allocator pools, the library's one-time tables
    assert toks == ref and ptoks == ref[:5]
    one_generation()
    before = _device_free_bytes()
    for _ in range(25):
        t2, p2 = one_generation()
        assert t2 == ref and p2 == ref[:5]
    mid = _device_free_bytes()
    for _ in range(8):
        m2 = lnb.LlamaTransformer(**TINY).fill_synthetic(7).finalize()
        c2 = lnb.InferenceContext(m2, 32); c2.Forward(prompt, 0, want_logits=False); c2.close(); m2.close()
    after = _device_free_bytes()
    # (the runtime hands memory back in 2 MiB granules: anything below one granule per object kind is pool noise, a leak grows with the count)
    assert before - mid <= 4 << 20 and mid - after <= 4 << 20, "device memory not returned: %.1f MB over 25 generations, %.1f MB over 8 models" % (
        (before - mid) / 1048576.0, (mid - after) / 1048576.0)


def test_tiny_device_greedy_loop_matches_oracle(lnb, tiny_pair):
    om, gm = tiny_pair
    prompt = orc.synth_tokens(5, 9, TINY["vocab_size"])
    ref, _ = orc.Context(om, 80).generate(prompt, 71)
    eng = lnb.InferenceEngine(gm, 80)
    got = eng.GenerateTokens(prompt)
    assert len(got) == 71 and list(ref) == got
    # last-row-only path (logits_out == NULL) returns the same argmax as the full path
    gc = lnb.InferenceContext(gm, 80)
    _, a1 = gc.Forward(prompt, 0, want_logits=False)
    gc.reset()
    lg, a2 = gc.Forward(prompt, 0, want_logits=True)
    assert a1 == a2 == int(orc.lib().orc_argmax_f32(orc._p(lg[-1]), lg.shape[1]))
    gc.close()


def test_tiny_long_prefill_through_the_matrix_cores_bit_exact(lnb, tiny_pair):
    """A 48-token prompt (>= 16 rows: rmsnorm_rows + gemm_mfma with the RoPE/KV, residual and SiLU*up epilogues, the K cache in
    its position-contiguous layout) and a chunked continuation of 48 more rows at start_pos 48: logits, KV caches and the
    following greedy steps must equal the oracle's bit for bit."""
    om, gm = tiny_pair
    oc, gc = orc.Context(om, 128), lnb.InferenceContext(gm, 128)
    toks = orc.synth_tokens(77, 96, TINY["vocab_size"])
    for lo_, hi_ in ((0, 48), (48, 96)):                      # second call: T = 96, S = 48 (T % S == 0, modulo-broadcast mask)
        lo, ao = oc.forward(toks[lo_:hi_], lo_)
        lg, ag = gc.Forward(toks[lo_:hi_], lo_)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    for layer in range(TINY["n_layers"]):
        assert (oc.cache(layer, 0)[:96] == gc.CacheK(layer)[:96]).all() and (oc.cache(layer, 1)[:96] == gc.CacheV(layer)[:96]).all()
    tok = ag
    for i in range(6):
        _, to = oc.forward([tok], 96 + i, want_logits=False)
        _, tg = gc.Forward(np.array([tok], dtype=np.int32), 96 + i, want_logits=False)
        assert to == tg
        tok = to
    gc.close(); oc.close()


@pytest.mark.parametrize("heads,kv_heads,dim,rows", [(4, 2, 512, 37), (4, 4, 256, 83), (2, 1, 256, 16)])
def test_prefill_attention_tiles_ragged_rows_and_head_dims(lnb, heads, kv_heads, dim, rows):
    """attn_mfma_kernel (16 query rows per wave): row counts that are not a multiple of 16 (clamped lanes, a partial diagonal tile),
    head_dim 128 and 64, GQA and MHA, then a ragged chunk at start_pos > 0 (modulo-broadcast mask over T > S) and decode steps that
    read the cache it wrote: bit-exact logits against the oracle."""
    cfg = dict(TINY, n_heads=heads, n_kv_heads=kv_heads, dim=dim, n_layers=2, vocab_size=512)
    om = orc.Model(**cfg).fill_synthetic(31).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(31).finalize()
    toks = orc.synth_tokens(8, 2 * rows, cfg["vocab_size"])
    oc, gc = orc.Context(om, 2 * rows + 8), lnb.InferenceContext(gm, 2 * rows + 8)
    for lo_, hi_ in ((0, rows), (rows, 2 * rows)):
        lo, ao = oc.forward(toks[lo_:hi_], lo_)
        lg, ag = gc.Forward(toks[lo_:hi_], lo_)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    tok = ag
    for i in range(3):
        lo, to = oc.forward([tok], 2 * rows + i)
        lg, tg = gc.Forward(np.array([tok], dtype=np.int32), 2 * rows + i)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and to == tg
        tok = to
    gc.close(); oc.close(); gm.close(); om.close()


def test_rw24_quad_chain_blocks_bit_exact(lnb, tiny_pair, monkeypatch):
    """The 8B wq|wk|wv matrix is stored as 256 blocks of 24 rows (gemv_quad_kernel: four lanes per row, RoPE partner four lanes away,
    the second chain wave half full).  Force the same kernel on the tiny model (dim 256 = one 256-step stage; q + k + v rows not a
    multiple of 24: a partial last block) and compare prefill, decode, the KV cache and the captured greedy loop with the oracle."""
    om, _ = tiny_pair
    monkeypatch.setenv("LNB_RW_QKV", "24")
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize()
    monkeypatch.delenv("LNB_RW_QKV")
    oc, gc = orc.Context(om, 48), lnb.InferenceContext(gm, 48)
    toks = orc.synth_tokens(5, 7, TINY["vocab_size"])
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    tok = ag
    for i in range(8):
        lo, to = oc.forward([tok], 7 + i)
        lg, tg = gc.Forward(np.array([tok], dtype=np.int32), 7 + i)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and to == tg
        tok = to
    for l in range(TINY["n_layers"]):
        assert (oc.cache(l, 0)[:15] == gc.CacheK(l)[:15]).all() and (oc.cache(l, 1)[:15] == gc.CacheV(l)[:15]).all(), l
    more, _ = gc.decode_greedy(tok, 15, 6)
    ref = []
    for i in range(6):
        _, tok = oc.forward([tok], 15 + i, want_logits=False)
        ref.append(tok)
    assert [int(t) for t in more] == [int(t) for t in ref]
    gc.close(); oc.close(); gm.close()


def test_rw56_two_chain_blocks_bit_exact(lnb, tiny_pair, monkeypatch):
    """The 8B gate|up matrix is stored as 256 blocks of 56 rows x 2 chains (one per CU).  Force the same kernel on the tiny
    model (ffn hidden 896 = 16 x 56) and compare prefill + decode with the oracle."""
    om, _ = tiny_pair
    monkeypatch.setenv("LNB_RW_W13", "56")
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize()
    monkeypatch.delenv("LNB_RW_W13")
    assert gm.ffn_hidden % 56 == 0
    oc, gc = orc.Context(om, 48), lnb.InferenceContext(gm, 48)
    toks = orc.synth_tokens(5, 7, TINY["vocab_size"])
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    tok = ag
    for i in range(8):
        lo, to = oc.forward([tok], 7 + i)
        lg, tg = gc.Forward(np.array([tok], dtype=np.int32), 7 + i)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and to == tg
        tok = to
    gc.close(); oc.close(); gm.close()


def test_error_behaviour_matches_reference(lnb, tiny_pair):
    _, gm = tiny_pair
    gc = lnb.InferenceContext(gm, 16)
    with pytest.raises(lnb.LnbError, match="empty token array"):             # llamatransformer.go:146-148
        gc.Forward(np.zeros(0, dtype=np.int32), 0)
    with pytest.raises(lnb.LnbError, match="incompatible locStart"):         # KV/RoPE Slice bounds, tensor.go:275-279
        gc.Forward(np.zeros(17, dtype=np.int32), 0)
    with pytest.raises(lnb.LnbError, match="cannot be broadcasted"):         # mask [S,S] vs scores [H,S,T], tensor.go:414-428
        gc.Forward(np.zeros(3, dtype=np.int32), 4)
    with pytest.raises(lnb.LnbError, match="outside the vocabulary"):
        gc.Forward(np.array([5, 99999], dtype=np.int32), 0)
    gc.close()


def test_gqa_and_unscaled_rope_config(lnb):
    cfg = dict(TINY, n_heads=8, n_kv_heads=8, use_scaled_rope=0, n_layers=1, vocab_size=512, dim=512)
    om = orc.Model(**cfg).fill_synthetic(77).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(77).finalize()
    toks = orc.synth_tokens(1, 7, cfg["vocab_size"])
    lo, ao = orc.Context(om, 32).forward(toks, 0)
    gc = lnb.InferenceContext(gm, 32)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    gc.close(); gm.close(); om.close()


def test_pipeline_stages_on_one_device_equal_whole_model(lnb, tiny_pair):
    """Layer-sharded pipeline developed with logical stages on one GPU: stage 0 = layer 0 +
    embedding, stage 1 = layer 1 + norm + output; the hidden state hand-off is a device copy here (RCCL send/recv
    between processes in bench.py)."""
    import ctypes as C
    om, gm = tiny_pair
    s0 = lnb.LlamaTransformer(layer_begin=0, layer_end=1, **TINY).fill_synthetic(1234).finalize()
    s1 = lnb.LlamaTransformer(layer_begin=1, layer_end=2, **TINY).fill_synthetic(1234).finalize()
    c0, c1 = lnb.InferenceContext(s0, 32), lnb.InferenceContext(s1, 32)
    toks = orc.synth_tokens(3, 6, TINY["vocab_size"])
    L = lnb.lib()
    tok = np.ascontiguousarray(toks, dtype=np.int32)
    lnb._chk(L.lnb_forward_stage(c0.h, lnb._p(tok), 6, 0, None, None))
    nbytes = 6 * TINY["dim"] * 2
    src, dst = L.lnb_ctx_hidden_ptr(c0.h, 1), L.lnb_ctx_hidden_ptr(c1.h, 0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst, src, nbytes, 3) == 0                             # hipMemcpyDeviceToDevice
    assert hip.hipDeviceSynchronize() == 0      # a device-to-device hipMemcpy need not have finished when it returns, and the contexts' streams are NON-blocking ones:
                                                # nothing else orders the copy in front of the next stage's launches (seen once the library spread the streams over 16 hardware queues)
    logits = np.empty((6, TINY["vocab_size"]), dtype=np.float32)
    am = C.c_int32(-2)
    lnb._chk(L.lnb_forward_stage(c1.h, None, 6, 0, lnb._p(logits), C.byref(am)))
    ref, ra = orc.Context(om, 32).forward(toks, 0)
    assert (ref.view(np.uint32) == logits.view(np.uint32)).all() and ra == am.value
    for c in (c0, c1):
        c.close()
    s0.close(); s1.close()


@pytest.mark.parametrize("cuts,rows", [((0, 1, 4, 6), 6), ((0, 2, 6), 6), ((0, 1, 2, 3, 4, 5, 6), 24), ((0, 3, 5, 6), 1), ((0, 5, 6), 20)])
def test_pipeline_stages_cut_inside_a_block(lnb, tiny_pair, cuts, rows):
    """lnb_model_create_parts: stages that start or end inside a block -- after its attention part (hand-off: the [S, dim] vector) or
    after its gate/up part (hand-off: that vector + the [S, ffn_hidden] activations).  Prefill (GEMV rows and the matrix-core path)
    and decode steps through every chain of stages must give the oracle's logits bit for bit; a stage holds exactly its parts' tensors."""
    import ctypes as C
    om, _ = tiny_pair
    L = lnb.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **TINY).fill_synthetic(1234).finalize() for a, b in zip(cuts[:-1], cuts[1:])]
    ctxs = [lnb.InferenceContext(s, 64) for s in stages]
    oc = orc.Context(om, 64)
    toks = orc.synth_tokens(3, rows + 2, TINY["vocab_size"])
    F = stages[0].ffn_hidden

    def through(tok, pos):
        t = np.ascontiguousarray(tok, dtype=np.int32)
        logits = np.empty((len(t), TINY["vocab_size"]), dtype=np.float32)
        am = C.c_int32(-2)
        for q, c in enumerate(ctxs):
            lastq = q == len(ctxs) - 1
            lnb._chk(L.lnb_forward_stage(c.h, lnb._p(t) if q == 0 else None, len(t), pos, lnb._p(logits) if lastq else None, C.byref(am) if lastq else None))
            if not lastq:
                assert hip.hipMemcpy(L.lnb_ctx_hidden_ptr(ctxs[q + 1].h, 0), L.lnb_ctx_hidden_ptr(c.h, 1), len(t) * TINY["dim"] * 2, 3) == 0
                if cuts[q + 1] % 3 == 2:                       # cut between gate/up and down: the activations travel too
                    assert hip.hipMemcpy(L.lnb_ctx_hidden_ptr(ctxs[q + 1].h, 2), L.lnb_ctx_hidden_ptr(c.h, 2), len(t) * F * 2, 3) == 0
                assert hip.hipDeviceSynchronize() == 0         # (a D2D hipMemcpy may return before it has run; the contexts' streams do not wait for the null stream)
        return logits, am.value

    for lo_, hi_ in ((0, rows), (rows, rows + 1), (rows + 1, rows + 2)):
        ref, ra = oc.forward(toks[lo_:hi_], lo_)
        got, ga = through(toks[lo_:hi_], lo_)
        assert (ref.view(np.uint32) == got.view(np.uint32)).all() and ra == ga
    for s, (a, b) in zip(stages, zip(cuts[:-1], cuts[1:])):
        held = {n for n, _ in s.tensor_infos()}
        for l in range(TINY["n_layers"]):
            assert ("layers.%d.attention.wq.weight" % l in held) == (a <= 3 * l < b)
            assert ("layers.%d.feed_forward.w3.weight" % l in held) == (a <= 3 * l + 1 < b) == ("layers.%d.ffn_norm.weight" % l in held)
            assert ("layers.%d.feed_forward.w2.weight" % l in held) == (a <= 3 * l + 2 < b)
    with pytest.raises(lnb.LnbError):
        lnb.LlamaTransformer(part_begin=3, part_end=3, **TINY)
    oc.close()
    for c in ctxs:
        c.close()
    for s in stages:
        s.close()


def test_forward_stage_begin_end_equals_forward_stage(lnb, tiny_pair):
    """the non-blocking halves of lnb_forward_stage (what a pipeline rank overlaps its exchange with): same token, one begin per end"""
    import ctypes as C
    _, gm = tiny_pair
    L = lnb.lib()
    toks = np.ascontiguousarray(orc.synth_tokens(11, 20, TINY["vocab_size"]), dtype=np.int32)
    a, b = lnb.InferenceContext(gm, 40), lnb.InferenceContext(gm, 40)
    ref, got = C.c_int32(-2), C.c_int32(-3)
    lnb._chk(L.lnb_forward_stage(a.h, lnb._p(toks), 20, 0, None, C.byref(ref)))
    with pytest.raises(lnb.LnbError, match="without a begin"):
        lnb._chk(L.lnb_forward_stage_end(b.h, C.byref(got)))
    lnb._chk(L.lnb_forward_stage_begin(b.h, lnb._p(toks), 20, 0, 1))
    with pytest.raises(lnb.LnbError, match="has not been ended"):
        lnb._chk(L.lnb_forward_stage_begin(b.h, lnb._p(toks), 20, 0, 1))
    lnb._chk(L.lnb_forward_stage_end(b.h, C.byref(got)))
    assert ref.value == got.value
    one = np.array([ref.value], dtype=np.int32)
    lnb._chk(L.lnb_forward_stage(a.h, lnb._p(one), 1, 20, None, C.byref(ref)))
    lnb._chk(L.lnb_forward_stage_begin(b.h, lnb._p(one), 1, 20, 1)); lnb._chk(L.lnb_forward_stage_end(b.h, C.byref(got)))
    assert ref.value == got.value
    bad = np.array([TINY["vocab_size"] + 5], dtype=np.int32)
    lnb._chk(L.lnb_forward_stage_begin(b.h, lnb._p(bad), 1, 21, 0))
    with pytest.raises(lnb.LnbError, match="outside the vocabulary"):
        lnb._chk(L.lnb_forward_stage_end(b.h, None))
    a.close(); b.close()


@pytest.mark.parametrize("mult,cut", [(2, 0), (1, 0), (2, 5), (1, 2), (2, 1)])
def test_pipeline_two_processes_share_one_gpu(lnb, mult, cut):
    """bench.py --gpus 2 in miniature: two torch.distributed ranks (gloo), each with half of the blocks on cuda:0, exchanging the
    hidden state and the token ring through pipeline.run_ticks; every generated token is checked against the oracle."""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    script = os.path.join(os.path.dirname(__file__), "native", "pipeline_two_rank.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, LNB_TEST_MULT=str(mult), LNB_TEST_CUT=str(cut)))     # cut 5 / 2: the activations travel too
    assert r.returncode == 0 and "PIPELINE_TWO_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("nb,copy", [(3, "1"), (5, "0")])
def test_pipeline_two_processes_batched_ticks_through_torch_distributed(lnb, nb, copy):
    """the torch.distributed fallback's BATCHED tick: groups of nb sequences through two ranks sharing this GPU over gloo -- a pipe
    without a transport (lnb_pipeline_init_host) runs the stage steps, pipeline.run_ticks_batched_torch moves lnb_batch_boundary_ptr's buffers --
    with and without the second weight copy; every sequence of every group equals the oracle's greedy continuation."""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    script = os.path.join(os.path.dirname(__file__), "native", "pipeline_two_rank.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, LNB_TEST_MULT="2", LNB_TEST_CUT="0", LNB_TEST_BATCH=str(nb), LNB_TEST_BATCH_COPY=copy))
    assert r.returncode == 0 and "PIPELINE_TWO_RANK_BATCHED_OK %d" % (4 * nb) in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_bench_two_ranks_prints_exactly_one_json_line(lnb, overlap):
    """The driver's N > 1 launch line (torch.distributed.run ... bench.py --gpus 2) end to end -- timed windows, barrier, max over
    ranks, rank-0 JSON -- with the two ranks sharing this GPU over gloo (LNB_PIPELINE_BACKEND; RCCL needs one GPU per rank)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--model", "tiny", "--prompt-len", "20"], capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, LNB_PIPELINE_BACKEND="gloo", LNB_PIPELINE_OVERLAP=overlap, LNB_PIPELINE_BATCH="3"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["sequences_in_flight"] == (4 if overlap == "1" else 2)
    assert abs(d["value"] - 6 * d["config"]["sequences_in_flight"] / (d["ms_per_step"] * 6 / 1e3)) / d["value"] < 1e-3
    bt = d["config"]["batched"]                               # the fallback's batched tick, next to the unbatched value as in the native line
    assert bt["groups"] == 4 and bt["batch"] == 3 and bt["tokens_per_s"] == d["config"]["value_batched"] > 0 and len(bt["tokens_seq0"]) == 8


def test_pipeline_stage_hidden_views_are_zero_copy_torch_tensors(lnb, tiny_pair):
    """pipeline.LnbStage (what `bench.py --gpus N` runs on every rank): the hidden state that RCCL sends/receives is a torch
    view of the library's own device buffer (CUDA array interface).  Two logical stages on this GPU, the hand-off done
    through those views (torch copy_ standing in for isend/irecv); tokens must equal the oracle's greedy continuation."""
    import torch
    import pipeline
    om, _ = tiny_pair
    P, steps = 6, 5
    st0 = pipeline.LnbStage(lnb, torch, TINY, 0, 2, 1, 32, 0)
    st1 = pipeline.LnbStage(lnb, torch, TINY, 1, 2, 1, 32, 0)
    prompt = orc.synth_tokens(11, P, TINY["vocab_size"])
    exp, _ = orc.Context(om, 32).generate(prompt, steps)
    got = []
    toks, rows, pos = np.ascontiguousarray(prompt, dtype=np.int32), P, 0
    for _ in range(steps):
        st0.run(0, rows, pos, toks); st0.synchronize()
        a, b = st0.hidden_buffer(0, rows), st1.hidden_buffer(0, rows)
        assert a.is_cuda and a.dtype == torch.int16 and tuple(a.shape) == (rows, TINY["dim"])
        assert a.data_ptr() == lnb.lib().lnb_ctx_hidden_ptr(st0.ctx[0].h, 0)            # a view, not a copy
        b.copy_(a); torch.cuda.synchronize()
        tok = st1.run(0, rows, pos, None); st1.synchronize()
        got.append(int(tok))
        pos += rows; rows = 1; toks = np.array([tok], dtype=np.int32)
    assert got == [int(t) for t in exp[:steps]]
    st0.close(); st1.close()


def test_model_loaded_from_a_torch_checkpoint_matches_the_oracle(lnb, tmp_path):
    """Weight ingestion end to end (SURVEY.md 8f "next" #2): torch.save writes a real zip checkpoint with Meta's key names,
    the library mmaps it, unpickles it and binds the tensors (host -> HBM, re-tiled); logits must be bit-identical to the
    oracle fed the same tensors.  Also the reference's getTensor errors (loader.go:183-192)."""
    torch = pytest.importorskip("torch")
    cfg = dict(TINY)
    gm = lnb.LlamaTransformer(**cfg)
    infos = gm.tensor_infos()
    assert len(infos) == 3 + 9 * cfg["n_layers"]
    g = torch.Generator().manual_seed(5)
    sd = {}
    for name, shape in infos:
        t = torch.randn(*shape, generator=g) * 0.05
        if "norm" in name:
            t = t + 1
        sd[name] = t.to(torch.bfloat16)
    path = str(tmp_path / "consolidated.00.pth")
    torch.save(sd, path)
    ck = lnb.Checkpoint(path)
    gm.load_checkpoint(ck).finalize()
    om = orc.Model(**cfg)
    for name, _ in infos:
        om.set_tensor(name, sd[name].contiguous().view(torch.int16).numpy().view(np.uint16).ravel())
    om.finalize()
    toks = orc.synth_tokens(21, 9, cfg["vocab_size"])
    lo, ao = orc.Context(om, 32).forward(toks, 0)
    gc = lnb.InferenceContext(gm, 32)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    gc.close(); om.close()
    # getTensor errors
    del sd["layers.1.ffn_norm.weight"]
    torch.save(sd, path)
    ck2 = lnb.Checkpoint(path)
    with pytest.raises(lnb.LnbError, match=r'tensor "layers\.1\.ffn_norm\.weight" not found'):
        gm.load_checkpoint(ck2)
    ck2.close()
    sd["layers.1.ffn_norm.weight"] = torch.ones(cfg["dim"] + 1).to(torch.bfloat16)
    torch.save(sd, path)
    ck3 = lnb.Checkpoint(path)
    with pytest.raises(lnb.LnbError, match=r"has incorrect shape; expected \[256\], got \[257\]"):
        gm.load_checkpoint(ck3)
    ck3.close(); ck.close(); gm.close()


def test_cpp_host_mirror_generates_the_oracle_tokens(lnb, tiny_pair):
    """The C++ mirror of the Go API (host/lnb_host.hpp: NewLlamaTransformer, InferenceEngine.GenerateTokens with the
    per-layer Logf hook) drives the same C ABI: its greedy continuation equals the oracle's."""
    import subprocess
    om, _ = tiny_pair
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "llama-nuts-and-bolts_amd"), "-llnb_hip", "-Wl,-rpath," + os.path.join(root, "llama-nuts-and-bolts_amd"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    prompt = [7, 99, 512, 3, 64]
    r = subprocess.run([exe, "48"] + [str(t) for t in prompt], capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("tokens:")][0]
    got = [int(t) for t in line.split()[1:] if t.lstrip("-").isdigit()]
    ref, _ = orc.Context(om, 48).generate(np.array(prompt, dtype=np.int32), 43)
    assert got == [int(t) for t in ref]
    assert "state=3" in line                                         # GSFinishedByReachingSeqLen (inference.go:240-246)
    logged = int([l for l in r.stdout.splitlines() if l.startswith("layers_logged:")][0].split()[1])
    assert logged == 2                                               # Logf fired once per layer of the prefill Forward


def test_stop_ids_are_checked_on_the_device(lnb, tiny_pair):
    """inference.go:233-252: generation ends with the first stop token, which is emitted.  lnb_ctx_set_stop_ids + lnb_decode_greedy_until:
    the same tokens whether the run is enqueued one step, seven steps or all at once per call; the frozen context can be continued by hand
    from where it stopped; lnb_batch_decode_until with a different stop id per sequence; and the C++ mirror's GenerateTokens with chunk 1 / 32."""
    import subprocess
    om, gm = tiny_pair
    prompt = orc.synth_tokens(77, 9, TINY["vocab_size"])
    ref, _ = orc.Context(om, 64).generate(prompt, 40)
    ref = [int(t) for t in ref]
    k = next(i for i in range(6, 30) if ref[i] not in ref[:i])      # a token that first appears at step k: the stop id
    stop = ref[k]
    for chunk in (1, 7, 64):
        gc = lnb.InferenceContext(gm, 64).set_stop_ids([1023 if stop != 1023 else 1022, stop])
        _, first = gc.Forward(prompt, 0, want_logits=False)
        got, pos, tok, fin = [first], len(prompt), first, (first == stop)
        while not fin and pos < 50:
            out, fin, _ = gc.decode_greedy_until(tok, pos, min(chunk, 50 - pos))
            got += [int(t) for t in out]; pos += len(out); tok = got[-1]
        assert got == ref[:k + 1] and fin, (chunk, got, ref[:k + 1])
        # the stopped context goes on by hand exactly where the reference would be after emitting the stop token
        gc.set_stop_ids([])
        more, _ = gc.decode_greedy(tok, pos, 5)
        assert [int(t) for t in more] == ref[k + 1:k + 6]
        gc.close()
    # batch: every sequence its own stop id (or none); finished sequences keep their caches and positions
    gb = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize().enable_batch()
    n, steps = 5, 20
    prompts = [orc.synth_tokens(300 + s, 6 + s, TINY["vocab_size"]) for s in range(n)]
    refs = [[int(t) for t in orc.Context(om, 64).generate(prompts[s], steps + 2)[0]] for s in range(n)]
    ctxs = [lnb.InferenceContext(gb, 64) for _ in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    ks = []
    for s in range(n):
        ks.append(None if s == 3 else next((i for i in range(2 + 3 * s, steps) if refs[s][i] not in refs[s][:i]), None))
        ctxs[s].set_stop_ids([] if ks[s] is None else [refs[s][ks[s]]])
    b = lnb.Batch(ctxs)
    outs, _ = b.decode_until(firsts, [len(p) for p in prompts], steps)
    for s in range(n):
        want = refs[s][1:1 + steps] if ks[s] is None else refs[s][1:ks[s] + 1]
        assert [int(t) for t in outs[s]] == want, (s, ks[s])
    b.close()
    for c in ctxs:
        c.close()
    gb.close()
    # the C++ mirror of generateTokensInternal: chunk sizes 1 and 32 end at the same stop token with GSFinishedByReachingEOS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "llama-nuts-and-bolts_amd"), "-llnb_hip", "-Wl,-rpath," + os.path.join(root, "llama-nuts-and-bolts_amd"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    lines = []
    for chunk in ("1", "32"):
        r = subprocess.run([exe, "64"] + [str(int(t)) for t in prompt], capture_output=True, text=True, check=True, env=dict(os.environ, LNB_STOP_IDS=str(stop), LNB_CHUNK=chunk))
        lines.append([l for l in r.stdout.splitlines() if l.startswith("tokens:")][0])
    assert lines[0] == lines[1] and "state=2" in lines[0]              # GSFinishedByReachingEOS (inference.go:13-17)
    assert [int(t) for t in lines[0].split()[1:] if t.lstrip("-").isdigit()] == ref[:k + 1]


def test_cpp_host_mirror_loadmodel_from_a_model_directory(lnb, tmp_path):
    """lnb::LoadModel(dir) = model.LoadModel (src/model/loader.go:18-70) without the tokenizer: consolidated.00.pth written by
    torch.save + params.json -> NewLlamaTransformer -> GenerateTokens; tokens must equal the oracle's with the same tensors."""
    import json
    import subprocess
    torch = pytest.importorskip("torch")
    cfg = dict(TINY)
    gm = lnb.LlamaTransformer(**cfg)
    infos = gm.tensor_infos()
    gm.close()
    g = torch.Generator().manual_seed(9)
    sd = {name: ((torch.randn(*shape, generator=g) * 0.05) + (1 if "norm" in name else 0)).to(torch.bfloat16) for name, shape in infos}
    torch.save(sd, str(tmp_path / "consolidated.00.pth"))
    (tmp_path / "params.json").write_text(json.dumps({"dim": cfg["dim"], "n_layers": cfg["n_layers"], "n_heads": cfg["n_heads"],
                                                       "n_kv_heads": cfg["n_kv_heads"], "multiple_of": cfg["multiple_of"],
                                                       "ffn_dim_multiplier": cfg["ffn_dim_multiplier"], "norm_eps": 1e-05,
                                                       "rope_theta": 500000.0, "use_scaled_rope": True}))   # no vocab_size: from tok_embeddings
    om = orc.Model(**cfg)
    for name, _ in infos:
        om.set_tensor(name, sd[name].contiguous().view(torch.int16).numpy().view(np.uint16).ravel())
    om.finalize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "llama-nuts-and-bolts_amd"), "-llnb_hip", "-Wl,-rpath," + os.path.join(root, "llama-nuts-and-bolts_amd"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    prompt = [5, 17, 300, 2]
    r = subprocess.run([exe, "24"] + [str(t) for t in prompt], capture_output=True, text=True, check=True,
                       env=dict(os.environ, LNB_MODEL_DIR=str(tmp_path)))
    line = [l for l in r.stdout.splitlines() if l.startswith("tokens:")][0]
    got = [int(t) for t in line.split()[1:] if t.lstrip("-").isdigit()]
    ref, _ = orc.Context(om, 24).generate(np.array(prompt, dtype=np.int32), 20)
    assert got == [int(t) for t in ref]
    om.close()
