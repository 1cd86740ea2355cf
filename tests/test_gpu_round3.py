"""GPU parity added in round 3 (run with -m gpu on an MI355X), HIP path vs the CPU oracle through the C ABI.

* The long-context decode attention (attn_long_scores_kernel + attn_long_pv_kernel) THROUGH THE XCD-REMAP BRANCH of xcd_head_block
  ((n_heads & 7) == 0: what the 8B shape's 32 heads take).  Every earlier oracle test of those kernels ran 4 or 2 heads, i.e. the
  plain (blockIdx.x, blockIdx.y) branch.  Head layouts 8/2, 16/4 and 8/8 at head_dim 64 and 128, prompts 300 ... 4100, the crossover
  forced off / on, both forms of the softmax denominator (llamatransformer.go:409-514).
* The 8B HEAD GEOMETRY itself (dim 4096, 32 query / 8 KV heads, head_dim 128, FFN 14336) cut to two layers with a 4096-token prompt
  + 64 greedy tokens: the configs[2] workload, against the oracle and against the committed golden of bench.py's `--model llama8b-2l`.
* Entry points mixed on one context (pipeline ticks, lnb_forward, lnb_decode_greedy, lnb_profile_kernel): the tick's captured graph
  must never replay at a position another entry point has overwritten.
* A multi-row call at head_dim 32 beyond the row-per-workgroup attention kernel's LDS reach is refused (it used to be admitted).
"""
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


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _cfg(dim, n_heads, n_kv_heads, **kw):
    return dict(orc.TINY, dim=dim, n_heads=n_heads, n_kv_heads=n_kv_heads, max_seq_len=2304, **kw)


# (n_heads & 7) == 0 in every one of them: the attention grids take the XCD-aware (head, block) mapping
XCD_CFGS = {
    "h8kv2_hd64": _cfg(512, 8, 2), "h16kv4_hd64": _cfg(1024, 16, 4), "h8kv8_hd64": _cfg(512, 8, 8),
    "h8kv2_hd128": _cfg(1024, 8, 2), "h16kv4_hd128": _cfg(2048, 16, 4), "h8kv8_hd128": _cfg(1024, 8, 8),
}
XCD_PROMPTS = {"h8kv2_hd64": (300, 700, 1100, 4100), "h8kv2_hd128": (300, 700, 1100, 4100)}


@pytest.mark.parametrize("name", sorted(XCD_CFGS))
def test_long_context_attention_through_the_xcd_remap_branch(lnb, name):
    cfg = XCD_CFGS[name]
    assert cfg["n_heads"] % 8 == 0
    om = orc.Model(**cfg).fill_synthetic(2024).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(2024).finalize()
    assert (_bits(om.rope_table()) == _bits(gm.PrecomputedFreqsCis)).all()
    for P in XCD_PROMPTS.get(name, (300, 700, 1100)):
        toks = orc.synth_tokens(31000 + P, P, cfg["vocab_size"])
        oc = orc.Context(om, P + 8)
        _, tok0 = oc.forward(toks, 0, want_logits=False)
        ref, tok = [], tok0
        for i in range(4):
            lo, tok_n = oc.forward([tok], P + i)
            ref.append((lo, tok_n)); tok = tok_n
        # crossover: never (the one-workgroup-per-head kernel), always (the long-context pair), the default 512; serial-Z forced or not
        for thr, zseq in ((10 ** 9, 0), (0, 0), (0, 1), (-1, 0)):
            gc = lnb.InferenceContext(gm, P + 8).set_attention(thr, zseq)
            _, t0 = gc.Forward(toks, 0, want_logits=False)           # 16+ rows: attn_mfma_kernel (bmajor form of the same remap)
            assert t0 == tok0, (name, P, thr, zseq)
            tok = t0
            for i in range(4):
                lg, tg = gc.Forward(np.array([tok], dtype=np.int32), P + i)
                assert np.abs(ref[i][0] - lg).max() <= 1e-2                                      # north_star tolerance
                assert (_bits(ref[i][0]) == _bits(lg)).all() and tg == ref[i][1], (name, P, thr, zseq, i)   # expected: bit-identical
                tok = tg
            assert gc.zseq_count() == (4 * cfg["n_layers"] * cfg["n_heads"] if zseq else 0), (name, P, thr, zseq)
            for layer in range(cfg["n_layers"]):
                assert (oc.cache(layer, 0)[:P + 4] == gc.CacheK(layer)[:P + 4]).all(), (name, P, thr, layer)
                assert (oc.cache(layer, 1)[:P + 4] == gc.CacheV(layer)[:P + 4]).all(), (name, P, thr, layer)
            gc.close()
        # the captured greedy loop (graph replays of the long-context form) continues the same history
        gc = lnb.InferenceContext(gm, P + 8).set_attention(0, 0)
        _, t0 = gc.Forward(toks, 0, want_logits=False)
        got, _ = gc.decode_greedy(t0, P, 4)
        assert [int(t) for t in got] == [r[1] for r in ref], (name, P)
        gc.close(); oc.close()
    gm.close(); om.close()


CFG2_2L = dict(orc.LLAMA_8B, n_layers=2, max_seq_len=2304)


@pytest.mark.skipif((os.cpu_count() or 1) < 32 and os.environ.get("LNB_TEST_FORCE_ORACLE") != "1", reason="the oracle needs ~2.5 T MAC for the 4096-row prefill")
def test_configs2_workload_at_the_8b_head_geometry_two_layers(lnb):
    """dim 4096, 32/8 heads, head_dim 128, FFN 14336, vocab 128256, two layers: 4096-token prompt (one call, matrix cores, 4096 RoPE rows =
    the reference's whole table) + 64 greedy tokens at T = 4097 ... 4160 through the long-context kernels; oracle run HERE, compared with
    the committed golden too (bench.py's check for `--model llama8b-2l --prompt-len 4096`)."""
    cfg, P, N = CFG2_2L, 4096, 65
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    prompt = orc.synth_tokens(99, P, cfg["vocab_size"])
    oc = orc.Context(om, P + N + 1)
    ref, _ = oc.generate(prompt, N)
    ref = [int(t) for t in ref]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs2_2layer_tokens.json")))["tokens"]
    assert ref == gold[:N], "the oracle run here disagrees with the committed golden"
    for thr, zseq in ((-1, 0), (-1, 1), (10 ** 9, 0)):           # default crossover (long kernels), serial Z forced, one-workgroup kernel
        gc = lnb.InferenceContext(gm, P + N + 1).set_attention(thr, zseq)
        _, first = gc.Forward(prompt, 0, want_logits=False)
        got, _ = gc.decode_greedy(first, P, N - 1)
        assert [first] + [int(t) for t in got] == ref, (thr, zseq)
        if thr == -1 and zseq == 0:
            assert gc.zseq_count() == 0
            for layer in range(2):
                assert (oc.cache(layer, 0)[:P + N - 1] == gc.CacheK(layer)[:P + N - 1]).all()
                assert (oc.cache(layer, 1)[:P + N - 1] == gc.CacheV(layer)[:P + N - 1]).all()
        gc.close()
    # one decode step's logits, bit for bit, at T = 4097 through the long kernels
    gc = lnb.InferenceContext(gm, P + 4)
    _, first = gc.Forward(prompt, 0, want_logits=False)
    oc2 = orc.Context(om, P + 4)
    oc2.forward(prompt, 0, want_logits=False)
    lo, ao = oc2.forward([first], P)
    lg, ag = gc.Forward(np.array([first], dtype=np.int32), P)
    assert (_bits(lo) == _bits(lg)).all() and ao == ag
    gc.close(); oc2.close(); oc.close(); gm.close(); om.close()


def test_committed_configs2_golden_is_reproduced_by_the_device(lnb):
    """no oracle run (any host): the device continuation of the two-layer configs[2] workload equals the committed oracle golden"""
    cfg, P = CFG2_2L, 4096
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs2_2layer_tokens.json")))
    assert gold["prompt_len"] == P and gold["n_layers"] == 2
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    prompt = lnb.synth_tokens(gold["prompt_seed"], P, cfg["vocab_size"])
    n = len(gold["tokens"])
    gc = lnb.InferenceContext(gm, P + n + 1)
    _, first = gc.Forward(prompt, 0, want_logits=False)
    got, _ = gc.decode_greedy(first, P, n - 1)
    assert [first] + [int(t) for t in got] == gold["tokens"]
    gc.close(); gm.close()


def test_entry_points_mixed_on_one_context_keep_the_device_position_right(lnb):
    """lnb_pipeline_tick skipped its set_state launch when the host-side dev_pos said the captured graph had left the right
    position behind -- but lnb_profile_kernel / lnb_forward / lnb_decode_greedy rewrote the device state without telling it.  Ticks
    interleaved with the other entry points on ONE context must keep producing the oracle's tokens."""
    cfg = dict(orc.TINY)
    om = orc.Model(**cfg).fill_synthetic(77).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(77).finalize()
    P, N = 12, 13
    prompt = orc.synth_tokens(5, P, cfg["vocab_size"])
    ref, _ = orc.Context(om, 64).generate(prompt, N)
    ref = [int(t) for t in ref]
    pipe = lnb.Pipeline(gm, 0, 1, None)
    gc = lnb.InferenceContext(gm, 64)
    got = {}

    def tick(i, pos, tokens=None, rows=1):                   # the tick that produces ref[i]
        got[i] = pipe.tick(run=gc, run_rows=rows, run_pos=pos, run_tokens=tokens)

    tick(0, 0, prompt, P)                                    # prefill through the tick path; the one-stage ring feeds each token back
    tick(1, P); tick(2, P + 1)                               # graph replays; the graph leaves position P + 2 on the device
    pipe.sync()
    # (a) lnb_profile_kernel: StepState.pos <- 40 (its KV writes land in row 40, beyond this run); the token ring is untouched.
    #     Before the fix the next tick skipped set_state (host-side dev_pos still said P + 2) and replayed the graph at position 40.
    gc.profile_kernel(0, 40, 2)
    tick(3, P + 2); tick(4, P + 3)
    pipe.sync()
    assert [int(pipe.read_tokens(got[i], 1)[0]) for i in range(5)] == ref[:5]
    # (b) the device greedy loop on the same context (it advances the position on the device by itself), then ticks again
    more, _ = gc.decode_greedy(ref[4], P + 4, 3)
    assert [int(t) for t in more] == ref[5:8]
    tick(8, P + 7, np.array([ref[7]], dtype=np.int32))       # a host-token tick re-seeds the ring
    tick(9, P + 8)
    pipe.sync()
    # (c) an eager Forward on the same context, then ticks
    _, t10 = gc.Forward(np.array([ref[9]], dtype=np.int32), P + 9, want_logits=False)
    assert t10 == ref[10]
    tick(11, P + 10, np.array([ref[10]], dtype=np.int32))
    tick(12, P + 11)
    pipe.sync()
    assert [int(pipe.read_tokens(got[i], 1)[0]) for i in (8, 9, 11, 12)] == [ref[8], ref[9], ref[11], ref[12]]
    pipe.close(); gc.close(); gm.close(); om.close()


def test_multi_row_call_at_head_dim_32_beyond_the_lds_reach_is_refused(lnb):
    """calls of 16+ rows at head_dim 32 do not run on the matrix-core attention (64 / 128 only) but on the row-per-workgroup
    kernel, whose LDS arrays hold ~12 K positions; the guard used to look only at the row count."""
    cfg = dict(orc.TINY, dim=128, n_heads=4, n_kv_heads=2, max_seq_len=8192)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1).finalize()
    gc = lnb.InferenceContext(gm, 14000)
    S = 16
    with pytest.raises(lnb.LnbError, match="row-per-workgroup"):
        gc.Forward(np.zeros(S, dtype=np.int32), 12800 - S)                        # T = 12800 = 800 * 16 > ~12.2 K
    lg, _ = gc.Forward(np.zeros(S, dtype=np.int32), 0)                            # the same call inside the reach still works
    assert np.isfinite(lg).all()
    gc.close(); gm.close()


def test_loopback_transport_refuses_out_of_order_runs(lnb):
    """in the in-process transport, running a sequence whose receive is posted but whose sender has not arrived is an
    error instead of a run on stale input."""
    cfg = dict(orc.TINY)
    g0 = lnb.LlamaTransformer(part_begin=0, part_end=3, **cfg).fill_synthetic(3).finalize()
    g1 = lnb.LlamaTransformer(part_begin=3, part_end=6, **cfg).fill_synthetic(3).finalize()
    p0, p1 = lnb.Pipeline(g0, 0, 2, loopback_group="ooo"), lnb.Pipeline(g1, 1, 2, loopback_group="ooo")
    c0, c1 = lnb.InferenceContext(g0, 32), lnb.InferenceContext(g1, 32)
    prompt = orc.synth_tokens(2, 4, cfg["vocab_size"])
    p1.tick(recv=c1, recv_rows=4)                                                 # rank 1 asks for its input first ...
    with pytest.raises(lnb.LnbError, match="lock-step"):
        p1.tick(run=c1, run_rows=4, run_pos=0)                                    # ... and must not run before rank 0 has posted it
    p0.tick(run=c0, run_rows=4, run_pos=0, run_tokens=prompt)
    p0.tick(send=c0, send_rows=4)
    p1.tick(run=c1, run_rows=4, run_pos=0)                                        # now it may
    p1.sync()
    for x in (p0, p1, c0, c1, g0, g1):
        x.close()
