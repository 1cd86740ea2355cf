"""Pin the CPU oracle against every weight-free known-answer test the reference ships
(SURVEY.md section 8c).  Values live in tests/golden/reference_kat.json (transcribed from the
reference's *_test.go files and docs; each entry cites file:line)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))
L = orc.lib()
P = orc._p
THRESHOLD_F32 = 1e-3   # src/common/utils.go:15


def bf(a):
    return orc.f32_to_bf16(np.asarray(a, dtype=np.float32))


def test_bf16_truncation_kat():
    for x, expected in KAT["bf16_truncation"]["cases"]:
        b = L.orc_f32_to_bf16(np.float32(x))
        assert L.orc_bf16_to_f32(b) == np.float32(expected)
    # truncation, not round-to-nearest-even: 1.53 would round UP to 1.53125 under RNE
    assert L.orc_bf16_to_f32(L.orc_f32_to_bf16(np.float32(1.53))) != np.float32(1.53125)


def test_bf16_le_decode_kat():
    for raw, bits, f32 in KAT["bf16_le_decode"]["cases"]:
        got = int(np.frombuffer(bytes(raw), dtype="<u2")[0])
        assert got == bits
        assert L.orc_bf16_to_f32(bits) == np.float32(f32)


def test_numpy_helpers_match_c():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1000).astype(np.float32) * 37.0
    b = orc.f32_to_bf16(x)
    for i in range(0, 1000, 37):
        assert b[i] == L.orc_f32_to_bf16(x[i])
        assert orc.bf16_to_f32(b[i:i + 1])[0] == np.float32(L.orc_bf16_to_f32(int(b[i])))


def test_linear_f32_kat():
    k = KAT["linear_f32"]
    w = np.array(k["weights"], dtype=np.float32); x = np.array(k["input"], dtype=np.float32)
    yf = np.zeros((2, 4), dtype=np.float32); yb = np.zeros((2, 4), dtype=np.uint16)
    L.orc_linear_f32(P(x), P(w), P(yf), P(yb), 2, 4, 3)
    assert np.abs(orc.bf16_to_f32(yb) - np.array(k["expected"], dtype=np.float32)).max() < THRESHOLD_F32


def test_linear_bf16_kat():
    k = KAT["linear_bf16"]
    w = bf(k["weights"]); x = bf(k["input"])
    y = np.zeros((2, 4), dtype=np.uint16)
    L.orc_linear_bf16(P(x), P(w), P(y), 2, 4, 3, 1)
    got = orc.bf16_to_f32(y)
    assert np.abs(got - np.array(k["expected"], dtype=np.float32)).max() < THRESHOLD_F32
    # exact restatement: sequential f32 accumulation then truncation
    xf, wf = orc.bf16_to_f32(x), orc.bf16_to_f32(w)
    for m in range(2):
        for n in range(4):
            acc = np.float32(0)
            for kk in range(3):
                acc = np.float32(acc + np.float32(xf[m, kk] * wf[n, kk]))
            assert y[m, n] == orc.f32_to_bf16(np.array([acc]))[0]


def test_matmul_bf16_kat_is_exact_truncation():
    k = KAT["matmul_bf16"]
    a = np.stack([bf(k["input"])] * k["batch"]); b = np.stack([bf(k["other"])] * k["batch"])
    c = np.zeros((k["batch"], 2, 4), dtype=np.uint16)
    L.orc_matmul_bf16(P(a), P(b), P(c), k["batch"], 2, 3, 4)
    got = orc.bf16_to_f32(c)
    exp = np.array(k["expected"], dtype=np.float32)
    for bi in range(k["batch"]):
        # the reference prints 5 significant digits of an exactly-truncated bf16: match to the print precision
        assert np.allclose(got[bi], exp, rtol=6e-5, atol=0), (got[bi], exp)
    # round-to-nearest-even would NOT reproduce these (SURVEY.md section 4 take-away 1)
    import torch
    rne = (torch.tensor(orc.bf16_to_f32(a[0])) @ torch.tensor(orc.bf16_to_f32(b[0]))).to(torch.bfloat16).float().numpy()
    assert not np.allclose(rne, exp, rtol=6e-5, atol=0)


def test_arange_kat():
    for start, end, step, expected in KAT["arange_bf16"]["cases"]:
        out = np.zeros(64, dtype=np.uint16)
        n = L.orc_arange_bf16(start, end, step, P(out))
        assert n == len(expected)
        assert list(orc.bf16_to_f32(out[:n])) == [float(v) for v in expected]
    assert L.orc_arange_bf16(5, 5, 1, P(np.zeros(4, dtype=np.uint16))) == -1   # start >= end is an error (impl:12-14)


def test_outer_kat():
    k = KAT["outer"]
    v1, v2 = bf(k["v1"]), bf(k["v2"])
    out = np.zeros((4, 3), dtype=np.uint16)
    L.orc_outer_bf16(P(v1), 4, P(v2), 3, P(out))
    assert (orc.bf16_to_f32(out) == np.array(k["expected"], dtype=np.float32)).all()


def test_polar_kat():
    k = KAT["polar"]
    ab = np.array(k["abs"], dtype=np.float32)
    an = np.array([np.float32(math.pi / d) if d else np.float32(0) for d in k["angle_pi_div"]], dtype=np.float32)
    out = np.zeros((5, 2), dtype=np.float32)
    L.orc_polar_f32(P(ab), P(an), P(out), 5)
    assert np.abs(out - np.array(k["expected"], dtype=np.float32)).max() < THRESHOLD_F32


def test_triu_kat():
    k = KAT["triu_square"]
    inp = bf(np.full((k["rows"], k["cols"]), k["fill"]))
    for diag, expected in k["cases"].items():
        out = np.full_like(inp, 0xFFFF)
        L.orc_triu_bf16(P(inp), P(out), k["rows"], k["cols"], int(diag))
        assert (orc.bf16_to_f32(out) == np.array(expected, dtype=np.float32)).all()


def test_causal_mask_like_prepare():
    # llamatransformer.go:128-136: Full(-inf bf16) then triu(diagonal=1); goldens :35-53 are 0 / -inf
    S = 15
    inp = bf(np.full((S, S), -np.inf))
    out = np.zeros_like(inp)
    L.orc_triu_bf16(P(inp), P(out), S, S, 1)
    m = orc.bf16_to_f32(out)
    for i in range(S):
        for j in range(S):
            assert m[i, j] == (-np.inf if j > i else 0.0)


def test_pow_mean_kat():
    k = KAT["pow_bf16"]
    out = np.zeros(5, dtype=np.float32)
    L.orc_pow_bf16(P(bf(k["input"])), P(out), 5, float(k["power"]))
    assert list(out) == [float(v) for v in k["expected"]]
    k = KAT["mean_3d"]
    g, last = k["shape"][0] * k["shape"][1], k["shape"][2]
    inp = np.arange(1, g * last + 1, dtype=np.float32)
    out = np.zeros(g, dtype=np.float32)
    L.orc_mean_f32(P(inp), P(out), g, last)
    assert list(out) == [float(v) for v in k["expected"]]


def test_rope_freqs_match_docs_table():
    k = KAT["rope_freqs_scaled"]
    fr = np.zeros(64, dtype=np.uint16)
    L.orc_rope_freqs(128, 500000.0, 1, P(fr))
    got = orc.bf16_to_f32(fr)
    exp = np.array(k["values"], dtype=np.float32)
    # printed with 5 significant digits; a bf16 has < 3, so this identifies every bf16 value uniquely
    assert np.allclose(got, exp, rtol=6e-5, atol=0), np.abs(got / exp - 1).max()


def test_rope_bf16_position_quirk():
    rows = 4096
    cis = np.zeros((rows, 64, 2), dtype=np.float32)
    ang = np.zeros((rows, 64), dtype=np.uint16)
    L.orc_rope_table(128, rows, 500000.0, 1, P(cis), P(ang))
    a = orc.bf16_to_f32(ang)
    for p, i, expected in KAT["rope_angles_bf16_positions"]["cases"]:
        assert np.isclose(a[p, i], np.float32(expected), rtol=6e-5, atol=0), (p, i, a[p, i], expected)
    # positions >= 256 lose low bits because t is a bf16 tensor (llamatransformer.go:725)
    assert (a[257] == a[256]).all() and (a[4095] == a[4080]).all()
    # cis = (cos, sin) of the bf16 angle, computed in f64 (operations_impl.go:127-133)
    assert np.allclose(cis[..., 0], np.cos(a.astype(np.float64)).astype(np.float32), atol=1e-7)
    assert np.allclose(cis[..., 1], np.sin(a.astype(np.float64)).astype(np.float32), atol=1e-7)


def test_ffn_hidden_dim():
    k = KAT["ffn_hidden_dim"]
    a = orc.make_args(dim=k["dim"], multiple_of=k["multiple_of"], ffn_dim_multiplier=k["ffn_dim_multiplier"])
    import ctypes
    assert L.orc_ffn_hidden_dim(ctypes.byref(a)) == k["expected"]


def test_silu_table_and_softmax_and_argmax():
    import ctypes
    t = np.ctypeslib.as_array(L.orc_silu_table(), shape=(65536,))
    for bits in (0x0000, 0x3F80, 0xBF80, 0x4040, 0x7F80):
        x = float(orc.bf16_to_f32(np.array([bits], dtype=np.uint16))[0])
        expected = np.float32(x / (1.0 + math.exp(-x))) if math.isfinite(x) else np.float32(x)
        assert t[bits] == expected
    x = np.array([[0.5, -1.0, 2.0, 0.0]], dtype=np.float32)
    out = np.zeros_like(x)
    L.orc_softmax_f32(P(x), P(out), 1, 4)
    e = np.exp(x.astype(np.float64)); ref = (e / e.sum()).astype(np.float32)
    assert np.abs(out - ref).max() < 1e-7
    # first-max-wins, NaN never selected, all -inf -> -1   (operations_impl.go:529-541)
    assert L.orc_argmax_f32(P(np.array([1, 3, 3, 2], dtype=np.float32)), 4) == 1
    assert L.orc_argmax_f32(P(np.array([np.nan, 1, np.nan], dtype=np.float32)), 3) == 1
    assert L.orc_argmax_f32(P(np.array([-np.inf, -np.inf], dtype=np.float32)), 2) == -1


def test_rmsnorm_restatement():
    rng = np.random.default_rng(1)
    dim = 64
    x = bf(rng.standard_normal((3, dim))); w = bf(1 + 0.1 * rng.standard_normal(dim))
    y = np.zeros((3, dim), dtype=np.uint16); pre = np.zeros((3, dim), dtype=np.uint16)
    L.orc_rmsnorm_bf16(P(x), P(w), P(y), 3, dim, np.float32(1e-5), P(pre))
    xf, wf = orc.bf16_to_f32(x), orc.bf16_to_f32(w)
    for r in range(3):
        s = np.float32(0)
        for kk in range(dim):
            s = np.float32(s + np.float32(np.float64(xf[r, kk]) ** 2))
        m = np.float32(np.float32(s / np.float32(dim)) + np.float32(1e-5))
        rs = np.float32(1.0 / math.sqrt(float(m)))
        h = orc.f32_to_bf16((xf[r] * rs).astype(np.float32))
        assert (h == pre[r]).all()
        assert (orc.f32_to_bf16(orc.bf16_to_f32(h) * wf) == y[r]).all()
