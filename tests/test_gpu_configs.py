"""GPU parity at the geometries of BASELINE.json's other configs (run with -m gpu on an MI355X), HIP path vs the CPU oracle, bit for bit:

* configs[2] (long context): contexts that cross every branch of the decode attention (second K register set above 512 cached
  positions, third pass above 1024, more than eight PV chunks) and of the prefill attention (hundreds of 16-position tiles), with
  RoPE rows beyond the reference's 4096-row table and bf16-quantised positions (llamatransformer.go:409-514, :109);
* configs[4] (70B-like shape cut to two layers): dim 8192, 64/8 heads, FFN 28672 as a whole Forward (the four-helper long-K w2
  kernel, the dim-8192 norm, 64-head attention);
* ml.Argmax edge cases through the kernel of the device greedy loop (operations_impl.go:513-548);
* decode -> multi-row Forward -> decode on ONE context (the captured decode graph must not outlive the logits buffer it was
  captured with).
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    assert _lnb.device_count() >= 1
    return _lnb


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# head_dim 64 (4 query heads on 2 KV heads) and head_dim 128 (2 on 1); max_seq_len 2304 -> a 4608-row RoPE table
LONG_CFGS = {
    "hd64": dict(orc.TINY, max_seq_len=2304),
    "hd128": dict(orc.TINY, n_heads=2, n_kv_heads=1, max_seq_len=2304),
}


@pytest.fixture(scope="module", params=sorted(LONG_CFGS))
def long_pair(lnb, request):
    cfg = LONG_CFGS[request.param]
    om = orc.Model(**cfg).fill_synthetic(4321).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(4321).finalize()
    assert gm.PrecomputedFreqsCis.shape[0] == 4608
    assert (_bits(om.rope_table()) == _bits(gm.PrecomputedFreqsCis)).all()
    yield cfg, om, gm
    gm.close(); om.close()


@pytest.mark.parametrize("chunks,steps", [
    ((510,), 4),            # decode at T = 511, 512, 513, 514: the scores loop's second K register set switches on at 513
    ((1023,), 3),           # T = 1024, 1025, 1026: third scores pass, 17 PV chunks
    ((1535,), 3),           # T = 1536, 1537, 1538
    ((4094,), 3),           # T = 4095, 4096, 4097: positions quantised to bf16 (4095 -> 4080), RoPE row 4096 = first row past the reference's table
    ((2080, 2080), 3),      # chunked prefill, second chunk at start_pos 2080 (T % S == 0: modulo-broadcast mask), then T = 4161..4163
])
def test_long_context_prefill_and_decode_bit_exact(lnb, long_pair, chunks, steps):
    cfg, om, gm = long_pair
    total = sum(chunks)
    seq_len = total + steps + 1
    toks = orc.synth_tokens(1000 + total, total, cfg["vocab_size"])
    oc, gc = orc.Context(om, seq_len), lnb.InferenceContext(gm, seq_len)
    pos = 0
    for n in chunks:
        lo, ao = oc.forward(toks[pos:pos + n], pos)
        lg, ag = gc.Forward(toks[pos:pos + n], pos)
        assert np.abs(lo - lg).max() <= 1e-2                                   # north_star tolerance
        assert (_bits(lo) == _bits(lg)).all() and ao == ag                     # expected: bit-identical
        pos += n
    tok = ag
    for i in range(steps):                                                     # S = 1 kernels (attn_exact_kernel) over the long cache
        lo, ao = oc.forward([tok], pos)
        lg, ag = gc.Forward(np.array([tok], dtype=np.int32), pos)
        assert (_bits(lo) == _bits(lg)).all() and ao == ag, "decode step at T = %d" % (pos + 1)
        tok, pos = ao, pos + 1
    for layer in range(cfg["n_layers"]):
        assert (oc.cache(layer, 0)[:pos] == gc.CacheK(layer)[:pos]).all()
        assert (oc.cache(layer, 1)[:pos] == gc.CacheV(layer)[:pos]).all()
    # the device greedy loop (hipGraph replays) continues from the same cache and gives the tokens of the step-by-step path
    gc2 = lnb.InferenceContext(gm, seq_len)
    p2 = 0
    for n in chunks:
        _, first = gc2.Forward(toks[p2:p2 + n], p2, want_logits=False)
        p2 += n
    got, _ = gc2.decode_greedy(first, total, steps)
    oc2 = orc.Context(om, seq_len)
    p2 = 0
    for n in chunks:
        _, t = oc2.forward(toks[p2:p2 + n], p2, want_logits=False)
        p2 += n
    ref = []
    for i in range(steps):
        _, t = oc2.forward([t], total + i, want_logits=False)
        ref.append(t)
    assert [int(x) for x in got] == ref
    for c in (oc, oc2):
        c.close()
    for c in (gc, gc2):
        c.close()


@pytest.mark.parametrize("P", [3, 70, 300, 700])
def test_long_context_attention_kernels_equal_the_oracle_at_every_context(lnb, long_pair, P):
    """attn_long_scores_kernel + attn_long_pv_kernel forced on at every T (threshold 0) and the one-workgroup-per-head kernel (threshold
    off), each with the certified tree estimate of the softmax denominator and with the serial f64 sum forced (force_zseq), against
    the oracle: identical logits bits, identical KV; no certification fallback unless forced."""
    cfg, om, gm = long_pair
    toks = orc.synth_tokens(5000 + P, P, cfg["vocab_size"])
    oc = orc.Context(om, P + 8)
    _, tok0 = oc.forward(toks, 0, want_logits=False)
    ref, tok = [], tok0
    for i in range(4):
        lo, tok_n = oc.forward([tok], P + i)
        ref.append((lo, tok_n)); tok = tok_n
    for thr, zseq in ((10 ** 9, 0), (10 ** 9, 1), (0, 0), (0, 1)):
        gc = lnb.InferenceContext(gm, P + 8).set_attention(thr, zseq)
        _, t0 = gc.Forward(toks, 0, want_logits=False)
        assert t0 == tok0
        tok = t0
        for i in range(4):
            lg, tg = gc.Forward(np.array([tok], dtype=np.int32), P + i)
            assert (_bits(ref[i][0]) == _bits(lg)).all() and tg == ref[i][1], (thr, zseq, i)
            tok = tg
        n = gc.zseq_count()
        # rows that took the serial walk: none unless forced; forced: the 4 decode steps (either kernel form) plus the prompt's rows when
        # the prompt is short enough to run on the one-workgroup kernel (fewer than 16 rows), x layers x heads
        rows = 4 + (P if P < 16 else 0)
        assert n == (rows * cfg["n_layers"] * cfg["n_heads"] if zseq else 0), (thr, zseq, n)
        for layer in range(cfg["n_layers"]):
            assert (oc.cache(layer, 0)[:P + 4] == gc.CacheK(layer)[:P + 4]).all() and (oc.cache(layer, 1)[:P + 4] == gc.CacheV(layer)[:P + 4]).all()
        gc.close()
    oc.close()


def test_context_beyond_the_reach_of_the_one_workgroup_per_head_kernel(lnb):
    """head_dim 128, 8192+ positions: more than the 12-bytes-per-position LDS staging of attn_exact_kernel holds.  The prefill runs on the
    matrix-core attention, one-token steps on the long-context kernels (whatever the crossover says); a call of 2..15 rows there is refused."""
    cfg = dict(LONG_CFGS["hd128"], max_seq_len=4200)
    om = orc.Model(**cfg).fill_synthetic(77).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(77).finalize()
    P, steps = 8188, 3
    toks = orc.synth_tokens(9, P, cfg["vocab_size"])
    oc, gc = orc.Context(om, P + 16), lnb.InferenceContext(gm, P + 16).set_attention(10 ** 9, 0)      # crossover "never": the reach decides
    lo, ao = oc.forward(toks, 0, want_logits=False)
    lg, ag = gc.Forward(toks, 0, want_logits=False)
    assert ao == ag
    tok = ao
    for i in range(steps):
        lo, ao = oc.forward([tok], P + i)
        lg, ag = gc.Forward(np.array([tok], dtype=np.int32), P + i)
        assert (_bits(lo) == _bits(lg)).all() and ao == ag, i
        tok = ao
    got, _ = gc.decode_greedy(tok, P + steps, 2)
    ref = []
    for i in range(2):
        _, tok = oc.forward([tok], P + steps + i, want_logits=False)
        ref.append(tok)
    assert [int(t) for t in got] == ref
    with pytest.raises(lnb.LnbError, match="2..15"):
        gc.Forward(np.zeros(2, dtype=np.int32), P + steps + 2 + 1)              # start 8194, T = 8196 = 2 * 4098
    with pytest.raises(lnb.LnbError, match="too long"):
        lnb.InferenceContext(gm, 40000)
    gc.close(); oc.close(); gm.close(); om.close()


def test_the_longest_context_the_library_accepts(lnb):
    """22 000 positions, head_dim 128: close to the limit lnb_ctx_create enforces (the long-context PV kernel keeps p_j for the whole
    context in the LDS).  The CPU oracle would need minutes for the 22 000-row prefill, so this is a device-side consistency check at
    full length: the prefill (chunks on the matrix cores; RoPE rows and bf16 positions far beyond the reference's table) feeds decode
    steps whose attention runs with the CERTIFIED softmax denominator and, on a second context with identical history, with the
    reference's serial f64 sum -- logits and tokens must agree bit for bit, no certification fallback may have been needed, and the
    captured loop must continue the eager steps."""
    cfg = dict(LONG_CFGS["hd128"], max_seq_len=11008)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(5).finalize()
    P, chunk = 21984, 2748                                        # 8 chunks: every call keeps T % S == 0 (the reference's mask rule)
    toks = orc.synth_tokens(11, P, cfg["vocab_size"])
    ctxs = [lnb.InferenceContext(gm, P + 24).set_attention(512, z) for z in (0, 1)]
    firsts = []
    for gc in ctxs:
        for c0 in range(0, P, chunk):
            _, tok = gc.Forward(toks[c0:c0 + chunk], c0, want_logits=False)
        firsts.append(tok)
    assert firsts[0] == firsts[1]
    tok = firsts[0]
    for i in range(4):
        la, ta = ctxs[0].Forward(np.array([tok], dtype=np.int32), P + i)
        lb, tb = ctxs[1].Forward(np.array([tok], dtype=np.int32), P + i)
        assert (_bits(la) == _bits(lb)).all() and ta == tb, i
        tok = ta
    assert ctxs[0].zseq_count() == 0 and ctxs[1].zseq_count() > 0
    if (os.cpu_count() or 1) >= 32 or os.environ.get("LNB_TEST_FORCE_ORACLE") == "1":
        # with enough host cores (about a minute of them) the oracle walks the same 22 000 positions: same chunks, then two decode steps
        om = orc.Model(**cfg).fill_synthetic(5).finalize()
        oc = orc.Context(om, P + 24)
        for c0 in range(0, P, chunk):
            _, otok = oc.forward(toks[c0:c0 + chunk], c0, want_logits=False)
        assert otok == firsts[0]
        t2 = otok
        gc2 = lnb.InferenceContext(gm, P + 24)
        for c0 in range(0, P, chunk):
            gc2.Forward(toks[c0:c0 + chunk], c0, want_logits=False)
        for i in range(2):
            lo, ao = oc.forward([t2], P + i)
            lg, ag = gc2.Forward(np.array([t2], dtype=np.int32), P + i)
            assert (_bits(lo) == _bits(lg)).all() and ao == ag, i
            t2 = ao
        gc2.close(); oc.close(); om.close()
    ga, _ = ctxs[0].decode_greedy(tok, P + 4, 6)
    gb, _ = ctxs[1].decode_greedy(tok, P + 4, 6)
    assert [int(t) for t in ga] == [int(t) for t in gb]
    for gc in ctxs:
        gc.close()
    gm.close()


def test_greedy_loop_switches_graphs_at_the_attention_crossover(lnb, long_pair):
    """lnb_decode_greedy replays the short-attention graph up to the crossover context and the long-attention graph beyond it: a run
    that starts below and ends above (default crossover 512, and a crossover in the middle of a short run) equals the oracle's tokens."""
    cfg, om, gm = long_pair
    for P, steps, thr in ((500, 30, -1), (20, 24, 30)):
        toks = orc.synth_tokens(7000 + P, P, cfg["vocab_size"])
        ref, _ = orc.Context(om, P + steps + 2).generate(toks, steps + 1)
        gc = lnb.InferenceContext(gm, P + steps + 2).set_attention(thr, 0)
        _, first = gc.Forward(toks, 0, want_logits=False)
        got, _ = gc.decode_greedy(first, P, steps)
        assert [first] + [int(t) for t in got] == [int(t) for t in ref]
        gc.close()


def test_llama70b_like_geometry_two_layers_bit_exact(lnb):
    """configs[4]'s shape (dim 8192, 64 query heads on 8 KV heads, FFN 28672) cut to two layers and a 2048-token vocabulary:
    a 24-row prefill (matrix-core path) and 32 one-token steps, logits and KV cache against the oracle."""
    cfg = dict(orc.LLAMA_8B, dim=8192, n_layers=2, n_heads=64, n_kv_heads=8, vocab_size=2048, multiple_of=4096)
    om = orc.Model(**cfg).fill_synthetic(70).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(70).finalize()
    assert gm.ffn_hidden == om.ffn_hidden == 28672 and gm.head_dim == 128
    P, N = 24, 32
    toks = orc.synth_tokens(170, P, cfg["vocab_size"])
    oc, gc = orc.Context(om, P + N + 1), lnb.InferenceContext(gm, P + N + 1)
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    assert np.abs(lo - lg).max() <= 1e-2
    assert (_bits(lo) == _bits(lg)).all() and ao == ag
    tok = ao
    for i in range(N):
        lo, ao = oc.forward([tok], P + i)
        lg, ag = gc.Forward(np.array([tok], dtype=np.int32), P + i)
        assert (_bits(lo) == _bits(lg)).all() and ao == ag, "decode step %d" % i
        tok = ao
    for layer in range(2):
        assert (oc.cache(layer, 0)[:P + N] == gc.CacheK(layer)[:P + N]).all()
        assert (oc.cache(layer, 1)[:P + N] == gc.CacheV(layer)[:P + N]).all()
    # and the hipGraph-replayed loop
    gc2 = lnb.InferenceContext(gm, P + N + 1)
    _, first = gc2.Forward(toks, 0, want_logits=False)
    got, _ = gc2.decode_greedy(first, P, N)
    ref, _ = orc.Context(om, P + N + 1).generate(toks, N + 1)
    assert [first] + [int(t) for t in got] == [int(t) for t in ref]
    gc.close(); gc2.close(); oc.close(); gm.close(); om.close()


def _argmax_ref(u16):
    f = orc.bf16_to_f32(np.asarray(u16, dtype=np.uint16))
    return int(orc.lib().orc_argmax_f32(orc._p(f), f.size))


def test_argmax_kernel_edge_cases(lnb):
    """ml.Argmax (operations_impl.go:529-541): strict '<' from -MaxFloat32: first maximum wins, NaN / -inf never selected."""
    NAN, NINF, PINF = 0x7FC0, 0xFF80, 0x7F80
    bf = orc.f32_to_bf16
    rng = np.random.default_rng(5)
    cases = []
    a = bf(rng.standard_normal(128256).astype(np.float32)); a[[100000, 7, 60000]] = bf(np.float32(9.0)); cases.append(("ties at distant indices", a, 7))
    a = bf(rng.standard_normal(128256).astype(np.float32)); a[[128255, 128248]] = bf(np.float32(8.5)); cases.append(("tie inside the last 16 B unit", a, 128248))
    a = bf(rng.standard_normal(4099).astype(np.float32)); a[4098:] = bf(np.float32(50.0)); cases.append(("maximum in the scalar tail (V % 8 != 0)", a, 4098))
    a = bf(rng.standard_normal(13).astype(np.float32)); cases.append(("V smaller than one 16 B unit x threads", a, None))
    a = bf(rng.standard_normal(1001).astype(np.float32)); a[::3] = NAN; cases.append(("NaNs are skipped", a, None))
    a = np.full(1024, NAN, dtype=np.uint16); a[777:778] = bf(np.float32(-3.0)); cases.append(("one number among NaNs", a, 777))
    cases.append(("all NaN", np.full(2048, NAN, dtype=np.uint16), -1))
    cases.append(("all -inf", np.full(128256, NINF, dtype=np.uint16), -1))
    a = np.full(5000, NINF, dtype=np.uint16); a[4321] = 0xFF7F; cases.append(("most negative finite bf16 beats -inf", a, 4321))
    a = bf(rng.standard_normal(9000).astype(np.float32)); a[[8000, 1234]] = PINF; cases.append(("+inf, first one", a, 1234))
    a = np.zeros(777, dtype=np.uint16); a[5] = 0x8000; cases.append(("+0 / -0 tie: index 0", a, 0))
    a = bf(rng.standard_normal(128256).astype(np.float32)); a[:] = np.minimum(a.view(np.int16), 0x3F00).view(np.uint16); cases.append(("saturated plateau", a, None))
    for name, arr, want in cases:
        ref = _argmax_ref(arr)
        if want is not None:
            assert ref == want, name                                    # the oracle itself on the constructed case
        assert lnb.op_argmax(arr) == ref, name
    for V in (8, 9, 1023, 1024, 4096 * 8, 4096 * 8 + 1, 128256):      # random rows with heavy ties (few distinct values)
        for rep in range(3):
            arr = bf(rng.integers(-3, 4, V).astype(np.float32))
            assert lnb.op_argmax(arr) == _argmax_ref(arr), (V, rep)


def test_decode_graph_survives_a_multi_row_forward_on_the_same_context(lnb):
    """multi-turn use of ONE InferenceContext: greedy decode (captures the hipGraph with the 1-row logits buffer), then a Forward
    that asks for the logits of 8 rows (the buffer is re-allocated), then greedy decode again -- every token and the 8 logits rows
    against the oracle."""
    cfg = dict(orc.TINY)
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    oc, gc = orc.Context(om, 64), lnb.InferenceContext(gm, 64)
    p1 = orc.synth_tokens(41, 8, cfg["vocab_size"])
    p2 = orc.synth_tokens(42, 8, cfg["vocab_size"])
    _, t0 = gc.Forward(p1, 0, want_logits=False)
    got1, _ = gc.decode_greedy(t0, 8, 8)                              # positions 8..15
    lg, t1 = gc.Forward(p2, 16, want_logits=True)                     # T = 24, S = 8
    got2, _ = gc.decode_greedy(t1, 24, 6)
    _, r0 = oc.forward(p1, 0, want_logits=False)
    ref1, tok = [], r0
    for i in range(8):
        _, tok = oc.forward([tok], 8 + i, want_logits=False)
        ref1.append(tok)
    lo, r1 = oc.forward(p2, 16)
    ref2, tok = [], r1
    for i in range(6):
        _, tok = oc.forward([tok], 24 + i, want_logits=False)
        ref2.append(tok)
    assert t0 == r0 and [int(t) for t in got1] == ref1
    assert (_bits(lo) == _bits(lg)).all() and t1 == r1
    assert [int(t) for t in got2] == ref2
    gc.close(); oc.close(); gm.close(); om.close()


def test_create_rejects_degenerate_arguments_without_crashing(lnb):
    for bad in (dict(n_kv_heads=0), dict(vocab_size=-1), dict(vocab_size=0), dict(n_layers=0), dict(multiple_of=0), dict(dim=0), dict(n_heads=0)):
        with pytest.raises(lnb.LnbError):
            lnb.LlamaTransformer(**dict(orc.TINY, **bad))


def test_calls_with_bad_arguments_fail_with_a_message_and_leave_the_handles_usable(lnb):
    """Every refusal goes through status < 0 + lnb_last_error() (the Go side turns it into an error value): none may crash, hang or
    poison the context.  Raw ctypes calls, so that nothing is filtered by the Python wrapper."""
    import ctypes as C
    L = lnb.lib()
    cfg = dict(orc.TINY)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(3).finalize()
    om = orc.Model(**cfg).fill_synthetic(3).finalize()
    out = C.c_void_p()
    assert L.lnb_ctx_create(gm.h, 10 ** 9, C.byref(out)) != 0 and b"too long" in L.lnb_last_error()
    for seq_len in (0, -5):                                                  # SequenceLength <= 0 means the model's MaxSequenceLength (inferencecontext.go:22-26)
        assert L.lnb_ctx_create(gm.h, seq_len, C.byref(out)) == 0 and out.value
        kv = np.zeros((cfg["max_seq_len"], cfg["n_kv_heads"] * (cfg["dim"] // cfg["n_heads"])), dtype=np.uint16)
        assert L.lnb_ctx_read_kv(out, 0, 1, kv.ctypes.data_as(C.c_void_p)) == 0 and not kv.any()      # max_seq_len rows of zeros (ml.Zeros)
        L.lnb_ctx_destroy(out)
    assert L.lnb_ctx_create(None, 16, C.byref(out)) != 0
    gc = lnb.InferenceContext(gm, 24)
    toks = np.ascontiguousarray(orc.synth_tokens(1, 8, cfg["vocab_size"]), dtype=np.int32)
    tp = toks.ctypes.data_as(C.c_void_p)
    am = C.c_int32(0)
    bad_calls = [
        lambda: L.lnb_forward(gc.h, tp, 0, 0, None, C.byref(am)),            # empty token array
        lambda: L.lnb_forward(gc.h, tp, -3, 0, None, C.byref(am)),
        lambda: L.lnb_forward(gc.h, None, 8, 0, None, C.byref(am)),          # null tokens
        lambda: L.lnb_forward(gc.h, tp, 8, -1, None, C.byref(am)),           # negative start
        lambda: L.lnb_forward(gc.h, tp, 8, 20, None, C.byref(am)),           # beyond the context
        lambda: L.lnb_forward(gc.h, tp, 8, 2 ** 31 - 4, None, C.byref(am)),  # start_pos + seq overflows int
        lambda: L.lnb_forward(None, tp, 8, 0, None, C.byref(am)),
        lambda: L.lnb_decode_greedy(gc.h, 5, 0, 0, tp, None),                # zero steps
        lambda: L.lnb_decode_greedy(gc.h, 5, 0, -2, tp, None),
        lambda: L.lnb_decode_greedy(gc.h, 5, 20, 8, tp, None),               # runs past the context
        lambda: L.lnb_decode_greedy(gc.h, -1, 0, 2, tp, None),               # token outside the vocabulary
        lambda: L.lnb_decode_greedy(gc.h, cfg["vocab_size"], 0, 2, tp, None),
        lambda: L.lnb_decode_greedy(gc.h, 5, 0, 2, None, None),              # no output buffer
        lambda: L.lnb_ctx_read_kv(gc.h, 99, 0, tp),
        lambda: L.lnb_ctx_read_kv(gc.h, -1, 0, tp),
        lambda: L.lnb_ctx_read_kv(gc.h, 0, 0, None),
        lambda: L.lnb_ctx_set_mode(gc.h, 7),
        lambda: L.lnb_profile_kernel(gc.h, 99, 0, 4, C.byref(C.c_float(0))),
        lambda: L.lnb_profile_kernel(gc.h, 1, 0, 0, C.byref(C.c_float(0))),
        lambda: L.lnb_forward_stage_end(gc.h, None),                         # end without begin
    ]
    for i, call in enumerate(bad_calls):
        assert call() != 0, i
        assert len(L.lnb_last_error()) > 0, i
    bad_tok = toks.copy(); bad_tok[3] = cfg["vocab_size"] + 7                 # a token id outside the vocabulary: reported with its index
    assert L.lnb_forward(gc.h, bad_tok.ctypes.data_as(C.c_void_p), 8, 0, None, C.byref(am)) != 0 and b"index 3" in L.lnb_last_error()
    # ... and the context still generates the oracle's tokens
    gc.reset()
    _, first = gc.Forward(toks, 0, want_logits=False)
    got, _ = gc.decode_greedy(first, 8, 6)
    ref, _ = orc.Context(om, 24).generate(toks, 7)
    assert [first] + [int(t) for t in got] == [int(t) for t in ref]
    gc.close(); gm.close(); om.close()


def test_model_and_pipe_api_misuse_is_refused(lnb):
    import ctypes as C
    L = lnb.lib()
    cfg = dict(orc.TINY)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(3)
    out = C.c_void_p()
    assert L.lnb_ctx_create(gm.h, 16, C.byref(out)) != 0 and b"not finalized" in L.lnb_last_error()
    pipe = C.c_void_p()
    assert L.lnb_pipeline_init(gm.h, 0, 1, None, C.byref(pipe)) != 0 and b"not finalized" in L.lnb_last_error()
    gm.finalize()
    w = np.zeros((cfg["vocab_size"], cfg["dim"]), dtype=np.uint16)
    shape = (C.c_int64 * 2)(cfg["vocab_size"], cfg["dim"])
    assert L.lnb_model_set_tensor(gm.h, b"output.weight", None, shape, 2) != 0                                  # null data
    assert L.lnb_model_set_tensor(gm.h, b"output.weight", w.ctypes.data_as(C.c_void_p), shape, 7) != 0         # absurd rank
    assert L.lnb_model_get_tensor(gm.h, b"output.weight", w.ctypes.data_as(C.c_void_p), 5) != 0               # buffer too small
    assert L.lnb_model_get_tensor(gm.h, b"no.such", w.ctypes.data_as(C.c_void_p), w.size) != 0
    nm, shp, rk = C.c_char_p(), (C.c_int64 * 2)(), C.c_int(0)
    assert L.lnb_model_tensor_info(gm.h, -1, C.byref(nm), shp, C.byref(rk)) != 0
    assert L.lnb_model_tensor_info(gm.h, 10 ** 6, C.byref(nm), shp, C.byref(rk)) != 0
    # pipe: ranks, foreign contexts, tokens on the wrong stage, a multi-row step without tokens
    assert L.lnb_pipeline_init(gm.h, 1, 1, None, C.byref(pipe)) != 0 and L.lnb_pipeline_init(gm.h, 0, 0, None, C.byref(pipe)) != 0
    assert L.lnb_pipeline_init(gm.h, 0, 2, None, C.byref(pipe)) != 0                                            # world 2 on a whole model
    p1 = lnb.Pipeline(gm, 0, 1)
    other = lnb.LlamaTransformer(**cfg).fill_synthetic(4).finalize()
    oc, gc = lnb.InferenceContext(other, 16), lnb.InferenceContext(gm, 16)
    slot = C.c_int(0)
    toks = np.ascontiguousarray(orc.synth_tokens(1, 4, cfg["vocab_size"]), dtype=np.int32)
    tp = toks.ctypes.data_as(C.c_void_p)
    assert L.lnb_pipeline_tick(p1.h, oc.h, 4, 0, tp, None, 0, None, 0, C.byref(slot)) != 0 and b"another model" in L.lnb_last_error()
    assert L.lnb_pipeline_tick(p1.h, gc.h, 4, 0, None, None, 0, None, 0, C.byref(slot)) != 0                    # multi-row step without tokens
    assert L.lnb_pipeline_tick(p1.h, gc.h, 4, 14, tp, None, 0, None, 0, C.byref(slot)) != 0                     # beyond the context
    assert L.lnb_pipeline_tick(p1.h, gc.h, 0, 0, tp, None, 0, None, 0, C.byref(slot)) != 0                      # no rows
    assert L.lnb_pipeline_read_tokens(p1.h, 0, 1, tp) != 0                                                      # nothing logged yet
    assert L.lnb_pipeline_tick(p1.h, gc.h, 4, 0, tp, None, 0, None, 0, C.byref(slot)) == 0 and slot.value == 0  # ... and a good tick still works
    assert L.lnb_pipeline_sync(p1.h) == 0
    om = orc.Model(**cfg).fill_synthetic(3).finalize()
    _, ref = orc.Context(om, 16).forward(toks, 0, want_logits=False)
    assert int(p1.read_tokens(0, 1)[0]) == ref
    p1.close(); oc.close(); gc.close(); other.close(); gm.close(); om.close()
