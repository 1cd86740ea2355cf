"""Batched EXACT decode (lnb_batch_*, -m gpu): 1..16 independent sequences per pass over the weights, each the column of an f32 matrix-core
product.  Parity: every sequence's tokens and KV-cache bits equal the CPU oracle's single-sequence run (tiny shapes) and the device's own
oracle-verified single-sequence path (8B shape; sequence 0 also against the committed configs[1] golden).
Reference: one context per generation (src/inference/inference.go:174), W shared across rows (src/ml/operations_lineartransform.go:173-193)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    assert _lnb.device_count() >= 1
    return _lnb


CFGS = {
    "tiny_hd64": dict(orc.TINY),                                                  # dim 256, 4/2 heads, FFN 896 (7 chunks of 128: not a multiple of the ring depth)
    "tiny_hd128": dict(orc.TINY, n_heads=2, n_kv_heads=1),
    "h8kv2_hd128": dict(orc.TINY, dim=1024, n_heads=8, n_kv_heads=2),             # (n_heads & 7) == 0: the XCD-aware attention grid
    "odd_tiles": dict(orc.TINY, dim=384, n_heads=3, n_kv_heads=1, vocab_size=1000, multiple_of=128),   # 24 / 40 / 63 tiles: ragged jobs, ragged vocabulary
}


@pytest.mark.parametrize("name", sorted(CFGS))
def test_every_sequence_of_a_batch_equals_its_own_oracle_run(lnb, name):
    cfg = CFGS[name]
    om = orc.Model(**cfg).fill_synthetic(606).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(606).finalize().enable_batch()
    assert gm.batch_bytes() > 0
    steps = 11
    for n in (1, 2, 5, 16, 17, 40):                                                # up to 16: columns of one matrix instruction; more: rows of the streaming product
        plens = [4 + 3 * s + (17 if s % 4 == 1 else 0) for s in range(n)]           # different prompt lengths: different positions per column
        prompts = [orc.synth_tokens(100 * n + s, plens[s], cfg["vocab_size"]) for s in range(n)]
        refs, ocs = [], []
        for s in range(n):
            oc = orc.Context(om, plens[s] + steps + 6)
            r, _ = oc.generate(prompts[s], steps + 4)
            refs.append([int(t) for t in r]); ocs.append(oc)
        ctxs = [lnb.InferenceContext(gm, plens[s] + steps + 6) for s in range(n)]
        firsts = []
        for s in range(n):
            _, t = ctxs[s].Forward(prompts[s], 0, want_logits=False)
            firsts.append(t)
        assert firsts == [r[0] for r in refs]
        b = lnb.Batch(ctxs)
        got, ms = b.decode(firsts, plens, steps)
        assert ms > 0
        for s in range(n):
            assert [int(t) for t in got[s]] == refs[s][1:1 + steps], (name, n, s)
            T = plens[s] + steps
            for layer in range(cfg["n_layers"]):
                assert (ocs[s].cache(layer, 0)[:T] == ctxs[s].CacheK(layer)[:T]).all(), (name, n, s, layer)
                assert (ocs[s].cache(layer, 1)[:T] == ctxs[s].CacheV(layer)[:T]).all(), (name, n, s, layer)
        # a second call on the same batch continues where the first stopped; then one context goes on alone
        more, _ = b.decode([refs[s][steps] for s in range(n)], [plens[s] + steps for s in range(n)], 2)
        for s in range(n):
            assert [int(t) for t in more[s]] == refs[s][steps + 1:steps + 3], (name, n, s)
        solo, _ = ctxs[n - 1].decode_greedy(refs[n - 1][steps + 2], plens[n - 1] + steps + 2, 1)
        assert int(solo[0]) == refs[n - 1][steps + 3]
        b.close()
        for c in ctxs:
            c.close()
        for oc in ocs:
            oc.close()
    gm.close(); om.close()


def test_batch_decode_at_the_8b_shape_equals_single_sequence_runs_and_the_golden(lnb):
    """Llama-3.1-8B shape, 16 prompts of 128 tokens in flight: each sequence's tokens equal its own run through lnb_decode_greedy (the
    path tests/test_gpu_full_8b.py checks against the oracle); sequence 0 is the configs[1] prompt: equal to the oracle's golden tokens"""
    cfg = dict(lnb.LLAMA_8B)
    P, steps, n = 128, 24, 16
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize().enable_batch()
    prompts = [lnb.synth_tokens(99 + s, P, cfg["vocab_size"]) for s in range(n)]
    ctxs = [lnb.InferenceContext(gm, P + steps + 8) for _ in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    b = lnb.Batch(ctxs)
    got, ms = b.decode(firsts, [P] * n, steps)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs1_tokens.json")))["tokens"]
    assert [firsts[0]] + [int(t) for t in got[0]] == gold[:steps + 1]
    solo = lnb.InferenceContext(gm, P + steps + 8)
    for s in (1, 7, 15):
        solo.reset()
        _, f = solo.Forward(prompts[s], 0, want_logits=False)
        assert f == firsts[s]
        ref, _ = solo.decode_greedy(f, P, steps)
        assert [int(t) for t in got[s]] == [int(t) for t in ref], s
        for layer in (0, 31):
            assert (solo.CacheK(layer)[:P + steps] == ctxs[s].CacheK(layer)[:P + steps]).all() and (solo.CacheV(layer)[:P + steps] == ctxs[s].CacheV(layer)[:P + steps]).all()
    print("8B shape, 16 sequences: %.3f ms per step = %.0f tokens/s aggregate" % (ms / steps, 1e3 * n * steps / ms))
    b.close(); solo.close()
    for c in ctxs:
        c.close()
    gm.close()


def test_wide_batch_decode_at_the_8b_shape_equals_single_sequence_runs_and_the_golden(lnb):
    """64 prompts in flight at the Llama-3.1-8B shape (rows of gemm_stream_kernel, 4 tiles of 16 sequences per wave where the grid allows):
    sequences 1, 30, 63 equal their own runs through lnb_decode_greedy, sequence 0 the configs[1] golden; different prompt lengths"""
    cfg = dict(lnb.LLAMA_8B)
    steps, n = 12, 64
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize().enable_batch()
    plens = [128 if s == 0 else 24 + (s * 7) % 50 for s in range(n)]
    prompts = [lnb.synth_tokens(99 + s, plens[s], cfg["vocab_size"]) for s in range(n)]
    ctxs = [lnb.InferenceContext(gm, plens[s] + steps + 8) for s in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    b = lnb.Batch(ctxs)
    got, ms = b.decode(firsts, plens, steps)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs1_tokens.json")))["tokens"]
    assert [firsts[0]] + [int(t) for t in got[0]] == gold[:steps + 1]
    for s in (1, 30, 63):
        solo = lnb.InferenceContext(gm, plens[s] + steps + 8)
        _, f = solo.Forward(prompts[s], 0, want_logits=False)
        assert f == firsts[s]
        ref, _ = solo.decode_greedy(f, plens[s], steps)
        assert [int(t) for t in got[s]] == [int(t) for t in ref], s
        T = plens[s] + steps
        for layer in (0, 31):
            assert (solo.CacheK(layer)[:T] == ctxs[s].CacheK(layer)[:T]).all() and (solo.CacheV(layer)[:T] == ctxs[s].CacheV(layer)[:T]).all()
        solo.close()
    print("8B shape, 64 sequences: %.3f ms per step = %.0f tokens/s aggregate" % (ms / steps, 1e3 * n * steps / ms))
    b.close()
    for c in ctxs:
        c.close()
    gm.close()


def test_column_group_batch_of_17_to_32_sequences_at_the_8b_shape(lnb):
    """Round 4: 17 .. 32 sequences run the thin matrices (wq|wk|wv, wo, w2) as two column groups of mfma_pair_kernel, the fat ones as rows of
    gemm_stream_kernel; the activations change layout at their producers (norm, attention, SiLU*up epilogue).  25 sequences (a ragged
    second group) of different prompt lengths at the 8B shape: sequence 0 the configs[1] golden, sequences 1, 15, 16, 24 their own
    single-sequence runs, KV caches included."""
    cfg = dict(lnb.LLAMA_8B)
    steps, n = 10, 25
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize().enable_batch()
    plens = [128 if s == 0 else 20 + (s * 11) % 60 for s in range(n)]
    prompts = [lnb.synth_tokens(99 + s, plens[s], cfg["vocab_size"]) for s in range(n)]
    ctxs = [lnb.InferenceContext(gm, plens[s] + steps + 8) for s in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    b = lnb.Batch(ctxs)
    got, ms = b.decode(firsts, plens, steps)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs1_tokens.json")))["tokens"]
    assert [firsts[0]] + [int(t) for t in got[0]] == gold[:steps + 1]
    for s in (1, 15, 16, 24):
        solo = lnb.InferenceContext(gm, plens[s] + steps + 8)
        _, f = solo.Forward(prompts[s], 0, want_logits=False)
        assert f == firsts[s]
        ref, _ = solo.decode_greedy(f, plens[s], steps)
        assert [int(t) for t in got[s]] == [int(t) for t in ref], s
        T = plens[s] + steps
        for layer in (0, 31):
            assert (solo.CacheK(layer)[:T] == ctxs[s].CacheK(layer)[:T]).all() and (solo.CacheV(layer)[:T] == ctxs[s].CacheV(layer)[:T]).all()
        solo.close()
    b.close()
    for c in ctxs:
        c.close()
    gm.close()


def test_batch_argument_checks(lnb):
    cfg = dict(orc.TINY)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1).finalize()
    c0, c1 = lnb.InferenceContext(gm, 32), lnb.InferenceContext(gm, 32)
    lnb.Batch([c0, c1]).close()                              # (round 5: without the second copy a batch runs its products as rows on the resident layouts)
    gm.enable_batch()
    with pytest.raises(lnb.LnbError, match="twice"):
        lnb.Batch([c0, c0])
    with pytest.raises(lnb.LnbError, match="1..128"):
        lnb.Batch([lnb.InferenceContext(gm, 16) for _ in range(129)])
    with pytest.raises(lnb.LnbError, match="enable_batch"):
        gm.fill_synthetic(2)                                 # the second copy would go stale
    c1.set_mode("fast")
    with pytest.raises(lnb.LnbError, match="exact-order only"):
        lnb.Batch([c0, c1])
    c1.set_mode("exact")
    b = lnb.Batch([c0, c1])
    with pytest.raises(lnb.LnbError, match="beyond the KV cache"):
        b.decode([1, 2], [30, 3], 4)
    with pytest.raises(lnb.LnbError, match="outside the vocabulary"):
        b.decode([1, cfg["vocab_size"]], [0, 0], 2)
    other = lnb.LlamaTransformer(**cfg).fill_synthetic(1).finalize().enable_batch()
    co = lnb.InferenceContext(other, 32)
    with pytest.raises(lnb.LnbError, match="another model"):
        lnb.Batch([c0, co])
    odd = lnb.LlamaTransformer(**dict(cfg, dim=192, n_heads=3, n_kv_heads=3)).fill_synthetic(1).finalize()
    with pytest.raises(lnb.LnbError, match="multiples of 128"):
        odd.enable_batch()
    for x in (b, c0, c1, co, other, odd, gm):
        x.close()


def _bf(a):
    return orc.f32_to_bf16(np.asarray(a, dtype=np.float32))


def _orc_linear(x, w):
    y = np.empty((x.shape[0], w.shape[0]), dtype=np.uint16)
    orc.lib().orc_linear_bf16(orc._p(np.ascontiguousarray(x)), orc._p(np.ascontiguousarray(w)), orc._p(y), x.shape[0], w.shape[0], x.shape[1], 0)
    return y


@pytest.mark.parametrize("rows,n,k,rw", [(16, 48, 128, 16), (17, 100, 512, 32), (40, 64, 4096, 64), (100, 272, 1024, 4), (128, 96, 256, 16),
                                         (130, 33, 384, 32), (300, 130, 640, 4), (64, 1000, 128, 64)])
def test_prefill_product_on_the_streaming_matrix_core_feed_is_bit_exact(lnb, rows, n, k, rw):
    """gemm_stream_kernel (weights from their M16 copy straight into the A operand, activations as f32 rows in the LDS, 1..8 batch tiles per
    wave, 1 / 2 / 4 waves per weight tile) against the oracle's k-ordered loop: ragged row counts, ragged tile counts, several row groups,
    every source layout the copy is made from"""
    rng = np.random.default_rng(rows * 31 + n + k + rw)
    x = _bf(rng.standard_normal((rows, k)) * 10 ** rng.uniform(-2, 2))
    w = _bf(rng.standard_normal((n, k)) * 0.05)
    os.environ["LNB_OP_STREAM"] = "1"
    try:
        y = lnb.op_linear(x, w, rw=rw)
    finally:
        del os.environ["LNB_OP_STREAM"]
    assert (y == _orc_linear(x, w)).all()


@pytest.mark.parametrize("name", ["tiny_hd64", "odd_tiles", "h8kv2_hd128"])
def test_prefill_of_a_batch_enabled_model_equals_the_oracle(lnb, name):
    """with the matrix-core copy present, Forward of 16 or more rows runs every product on gemm_stream_kernel: logits of all rows, KV cache
    and a chunked continuation (start_pos > 0) bit for bit against the oracle; the same model without the copy gives the same bits"""
    cfg = CFGS[name]
    om = orc.Model(**cfg).fill_synthetic(808).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(808).finalize().enable_batch()
    g0 = lnb.LlamaTransformer(**cfg).fill_synthetic(808).finalize()
    for chunks in ((16,), (37,), (150,), (48, 48), (20, 20, 20)):
        total = sum(chunks)
        toks = orc.synth_tokens(4000 + total, total, cfg["vocab_size"])
        oc, gc, c0 = orc.Context(om, total + 4), lnb.InferenceContext(gm, total + 4), lnb.InferenceContext(g0, total + 4)
        pos = 0
        for nrows in chunks:
            lo, ao = oc.forward(toks[pos:pos + nrows], pos)
            lg, ag = gc.Forward(toks[pos:pos + nrows], pos)
            l0, a0 = c0.Forward(toks[pos:pos + nrows], pos)
            assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag, (name, chunks, pos)
            assert (l0.view(np.uint32) == lg.view(np.uint32)).all() and a0 == ag
            pos += nrows
        for layer in range(cfg["n_layers"]):
            assert (oc.cache(layer, 0)[:total] == gc.CacheK(layer)[:total]).all() and (oc.cache(layer, 1)[:total] == gc.CacheV(layer)[:total]).all()
        got, _ = gc.decode_greedy(ag, total, 3)
        ref = []
        t = ao
        for i in range(3):
            _, t = oc.forward([t], total + i, want_logits=False)
            ref.append(t)
        assert [int(v) for v in got] == ref
        oc.close(); gc.close(); c0.close()
    gm.close(); g0.close(); om.close()


def test_prefill_at_the_8b_shape_on_the_streaming_feed_reproduces_the_golden(lnb):
    """configs[1]'s 128-token prompt through a batch-enabled 8B model (every prefill product on gemm_stream_kernel), then the greedy loop:
    the oracle's golden continuation; prefill time printed next to the LDS-tiled kernel's"""
    import time
    cfg = dict(lnb.LLAMA_8B)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs1_tokens.json")))["tokens"]
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    prompt = lnb.synth_tokens(99, 128, cfg["vocab_size"])
    c = lnb.InferenceContext(gm, 192)
    times = {}
    for label in ("LDS-tiled gemm_mfma_kernel", "streaming gemm_stream_kernel"):
        if label.startswith("streaming"):
            gm.enable_batch()
        for rep in range(2):
            c.reset()
            t0 = time.perf_counter()
            _, first = c.Forward(prompt, 0, want_logits=False)
            times[label] = round(1e3 * (time.perf_counter() - t0), 2)
        assert first == gold[0], label
        got, _ = c.decode_greedy(first, 128, 40)
        assert [first] + [int(t) for t in got] == gold[:41], label
    print("128-row prefill of the 8B shape, ms:", times)
    c.close(); gm.close()


def test_dense_batch_attention_beyond_one_pass_of_512_positions(lnb, monkeypatch):
    """attn_exact_kernel<HD, DENSE> (chosen for a batch when heads x sequences > 256) reloads its K rows per 512-position pass; the
    other batch tests stop at ~150 positions.  Here: 70 sequences x 4 heads on the tiny shape with prompts of 480..1100 positions -- some
    cross 512 and 1024 INSIDE the decode window -- against each sequence's own single-sequence run (lnb_decode_greedy: the oracle-checked
    path), against the oracle itself for three of them, and against the non-DENSE form (LNB_ATTN_BATCH_DENSE=0)."""
    cfg = dict(orc.TINY)
    n, steps = 70, 10
    plens = [480 + (s * 97) % 620 for s in range(n)]
    plens[0], plens[1], plens[2], plens[3] = 507, 1019, 511, 1100          # cross 512 / 1024 within the window; exactly at the edge; the longest
    seqmax = max(plens) + steps + 6
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(606).finalize(rope_rows=seqmax + 8).enable_batch()
    om = orc.Model(**cfg).fill_synthetic(606).finalize()
    prompts = [orc.synth_tokens(7000 + s, plens[s], cfg["vocab_size"]) for s in range(n)]

    def run(dense):
        if dense is not None:
            monkeypatch.setenv("LNB_ATTN_BATCH_DENSE", dense)
        ctxs = [lnb.InferenceContext(gm, seqmax) for _ in range(n)]
        firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
        b = lnb.Batch(ctxs)
        got, _ = b.decode(firsts, plens, steps)
        toks = [[firsts[s]] + [int(t) for t in got[s]] for s in range(n)]
        kv = [ctxs[s].CacheK(1)[:plens[s] + steps].copy() for s in (0, 1, 3)]
        b.close()
        for c in ctxs:
            c.close()
        if dense is not None:
            monkeypatch.delenv("LNB_ATTN_BATCH_DENSE")
        return toks, kv

    toks, kv = run(None)
    toks0, kv0 = run("0")
    assert toks == toks0 and all((a == b_).all() for a, b_ in zip(kv, kv0))
    solo = lnb.InferenceContext(gm, seqmax)
    for s in (0, 1, 2, 3, 17, 69):
        solo.reset()
        _, f = solo.Forward(prompts[s], 0, want_logits=False)
        ref, _ = solo.decode_greedy(f, plens[s], steps)
        assert toks[s] == [f] + [int(t) for t in ref], s
    solo.close()
    rope_ok = seqmax + 8 <= 2 * cfg["max_seq_len"]                       # the oracle's RoPE table has the reference's 2 x max_seq_len rows
    if rope_ok:
        for s in (0, 2):
            oc = orc.Context(om, seqmax)
            r, _ = oc.generate(prompts[s], steps + 1)
            assert toks[s] == [int(t) for t in r][:steps + 1], s
            oc.close()
    om.close(); gm.close()


@pytest.mark.parametrize("heads,kvh,force_z", [(4, 1, "0"), (8, 2, "0"), (4, 1, "1")])
def test_gqa_aware_batch_attention_equals_the_oracle_per_sequence(lnb, monkeypatch, heads, kvh, force_z):
    """Round 4: attn_gqa_kernel -- one workgroup per (KV head, sequence) serves the four query heads that share the KV head (K row loaded
    once for four score chains, V staged once, one PV chain per thread).  Forced on (LNB_ATTN_GQA=1; by default it takes over from
    KV heads x sequences >= 256) for narrow (<= 16) and wide batches, prompts that cross one 512-position scores pass and several 64-position
    PV chunks, and with the serial softmax denominator forced; every sequence against its own CPU-oracle run, KV caches included."""
    cfg = dict(orc.TINY, dim=128 * heads, n_heads=heads, n_kv_heads=kvh)
    monkeypatch.setenv("LNB_ATTN_GQA", "1")
    monkeypatch.setenv("LNB_ATTN_GQA_FORCE_ZSEQ", force_z)
    steps = 6
    om = orc.Model(**cfg).fill_synthetic(707).finalize()
    for n in ((3, 17) if force_z == "0" else (5,)):
        plens = [5 + 61 * s for s in range(n)]
        plens[0] = 509                                                              # crosses 512 inside the decode window
        if n > 4: plens[4] = 63                                                     # crosses a PV chunk edge
        seqmax = max(plens) + steps + 6
        gm = lnb.LlamaTransformer(**cfg).fill_synthetic(707).finalize(rope_rows=seqmax + 8).enable_batch()
        prompts = [orc.synth_tokens(31 * n + s, plens[s], cfg["vocab_size"]) for s in range(n)]
        ctxs = [lnb.InferenceContext(gm, seqmax) for _ in range(n)]
        firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
        b = lnb.Batch(ctxs)
        got, _ = b.decode(firsts, plens, steps)
        for s in range(n):
            oc = orc.Context(om, seqmax)
            r, _ = oc.generate(prompts[s], steps + 1)
            assert [firsts[s]] + [int(t) for t in got[s]] == [int(t) for t in r][:steps + 1], (n, s)
            T = plens[s] + steps
            assert (oc.cache(1, 1)[:T] == ctxs[s].CacheV(1)[:T]).all(), (n, s)
            oc.close()
        b.close()
        for c in ctxs:
            c.close()
        gm.close()
    om.close()


def test_a_member_context_cannot_be_destroyed_under_a_live_batch(lnb):
    """a batch bakes its members' device pointers into its tables and graphs -- lnb_ctx_destroy on a member must fail until the
    batch is gone (then succeed); the context and the batch stay usable after the refused call"""
    cfg = dict(orc.TINY)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(5).finalize().enable_batch()
    ctxs = [lnb.InferenceContext(gm, 40) for _ in range(3)]
    firsts = [c.Forward(orc.synth_tokens(s, 5, cfg["vocab_size"]), 0, want_logits=False)[1] for s, c in enumerate(ctxs)]
    b = lnb.Batch(ctxs)
    L = lnb.lib()
    assert L.lnb_ctx_destroy(ctxs[1].h) != 0 and b"live batch" in L.lnb_last_error()
    got, _ = b.decode(firsts, [5, 5, 5], 3)                              # still works
    assert got.shape == (3, 3)
    b.check_error()
    b.close()
    for c in ctxs:
        c.close()                                                        # now fine
    gm.close()
