"""Three faithful float64 `exp` implementations over the 65536 bf16 inputs a softmax can see (src/ml/operations_impl.go:498: math.Exp(float64(bf16 score))):
Go's portable algorithm restated (oracle/go_exp.py: fdlibm's e_exp, what src/math/exp.go ports), the host libm (what the C oracle calls), and -- in the GPU suite,
tests/test_gpu_parity.py -- the device's ocml.  None is the other at the last ulp; all coincide once narrowed to float32, which is the only form in which a softmax
numerator leaves float64 (impl:506: float32(e / Z))."""
import math

import numpy as np

from oracle.go_exp import go_exp


def _inputs():
    with np.errstate(all="ignore"):                         # (signalling NaN patterns among the 65536)
        return (np.arange(65536, dtype=np.uint32) << 16).view(np.float32).astype(np.float64)


def _libm(v):
    try:
        return math.exp(v)
    except OverflowError:
        return math.inf


def test_gos_portable_exp_restated_is_within_one_ulp_of_the_host_libm_on_every_bf16_input():
    differ, worst, f32_differ = 0, 0, 0
    for v in _inputs():
        v = float(v)
        if v != v:
            assert go_exp(v) != go_exp(v)                    # NaN in, NaN out
            continue
        g, h = go_exp(v), _libm(v)
        assert math.isinf(g) == math.isinf(h) and (g == 0.0) == (h == 0.0), v      # same overflow / underflow points on this input set
        if g != h:
            differ += 1
            worst = max(worst, abs(int(np.float64(g).view(np.int64)) - int(np.float64(h).view(np.int64))))
        with np.errstate(all="ignore"):
            f32_differ += int(np.float32(g).view(np.uint32) != np.float32(h).view(np.uint32))
    # measured (glibc 2.35): 491 inputs differ, each by one ulp -- a mistyped constant in the restatement would be off by thousands of ulps on most inputs
    assert worst <= 1 and differ <= 1000 and f32_differ == 0, (differ, worst, f32_differ)


def test_special_cases_of_exp_go():
    assert go_exp(0.0) == 1.0 and go_exp(-0.0) == 1.0 and go_exp(math.inf) == math.inf and go_exp(-math.inf) == 0.0
    assert go_exp(710.0) == math.inf and go_exp(-746.0) == 0.0 and go_exp(1e-10) == 1.0 + 1e-10
    assert abs(int(np.float64(go_exp(1.0)).view(np.int64)) - int(np.float64(math.e).view(np.int64))) <= 1      # (fdlibm's exp(1) is the float64 above math.E)


def test_the_c_restatement_in_the_oracle_is_the_python_one_bit_for_bit():
    """oracle/lnb_oracle.c: exp_go (behind orc_set_exp_impl(1)) against oracle/go_exp.py on all 65536 inputs"""
    from oracle import oracle as orc
    L = orc.lib()
    L.orc_set_exp_impl(1)
    try:
        for v in _inputs():
            v = float(v)
            a, b = L.orc_exp_f64(v), go_exp(v)
            assert (a != a and b != b) or np.float64(a).view(np.int64) == np.float64(b).view(np.int64), v
    finally:
        L.orc_set_exp_impl(0)
    assert L.orc_exp_f64(1.0) == math.exp(1.0)


def test_not_one_logit_bit_of_a_model_run_depends_on_which_exp_the_oracle_calls():
    """The whole point of measuring the implementations against each other: run the ORACLE with the host libm's exp and again with Go's portable math.Exp restated
    (softmax numerators and denominators, operations_impl.go:498/506, and the SiLU table, activations.go:24) -- every logit of every row of the prompt's Forward and of the
    following greedy steps must have the same bits, on several shapes and seeds (~1.5 million softmax elements and the whole 65536-entry SiLU table go through the other exp)."""
    from oracle import oracle as orc
    L = orc.lib()
    shapes = [dict(orc.TINY), dict(orc.TINY, dim=512, n_heads=8, n_kv_heads=2, n_layers=4, vocab_size=2048), dict(orc.TINY, dim=384, n_heads=6, n_kv_heads=6, n_layers=3)]
    try:
        for k, cfg in enumerate(shapes):
            runs = []
            for impl in (0, 1):
                L.orc_set_exp_impl(impl)
                om = orc.Model(**cfg).fill_synthetic(70 + k).finalize()
                oc = orc.Context(om, 160)
                lg, tok = oc.forward(orc.synth_tokens(5 + k, 96, cfg["vocab_size"]), 0)
                rows, toks = [lg.view(np.uint32).copy()], [tok]
                for i in range(24):
                    lg, tok = oc.forward(np.array([toks[-1]], dtype=np.int32), 96 + i)
                    rows.append(lg.view(np.uint32).copy()); toks.append(tok)
                runs.append((rows, toks))
                oc.close(); om.close()
            assert runs[0][1] == runs[1][1], k
            assert all((a == b).all() for a, b in zip(runs[0][0], runs[1][0])), k
    finally:
        L.orc_set_exp_impl(0)
