"""Round 5 (-m gpu, through the C ABI):
* the committed configs[2] oracle goldens at depth -- the 8-layer cut and the FULL 32-layer model (4096-token prompt + greedy tokens at
  T > 4096) -- replayed device-only (the driver's GPU suite must see them; the goldens are made once on host cores by
  tests/golden/make_configs2_cut_tokens.py);
* the throughput schedule (lnb_ctx_set_schedule): other FORMS of the one-token kernels, the same bits;
* stop ids: lnb_decode_greedy / lnb_batch_decode ignore them, the _until forms report finished flags, chunked batches keep ended sequences
  frozen, multi-stage batched pipelines refuse them;
* lnb_ctx_destroy refused under a live batch is reported by the Python wrapper; the cycle-stamp ABI bench.py builds its measured model on."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(orc.TINY)


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    assert _lnb.device_count() >= 1
    return _lnb


def _golden(n_layers):
    return os.path.join(ROOT, "tests", "golden", "configs2_%dlayer_tokens.json" % n_layers)


@pytest.mark.parametrize("n_layers", [8, 32])
def test_committed_configs2_goldens_at_depth_are_reproduced_by_the_device(lnb, n_layers):
    """llamatransformer.go:409-514 over T > 4096 rows, operations_impl.go:478-511, after 8 / 32 residual blocks: the 4096-token prompt in ONE
    Forward (f32 matrix cores) + the greedy continuation through the long-context attention kernels = the CPU oracle's tokens
    (tests/test_golden_files.py checks on the CPU side that the files are there: a missing golden fails THAT suite instead of skipping here)."""
    if not os.path.exists(_golden(n_layers)):
        pytest.skip("tests/golden/configs2_%dlayer_tokens.json not generated yet" % n_layers)
    gold = json.load(open(_golden(n_layers)))
    P = gold["prompt_len"]
    assert P == 4096 and gold["n_layers"] == n_layers
    cfg = dict(orc.LLAMA_8B, n_layers=n_layers, max_seq_len=2304)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    prompt = lnb.synth_tokens(gold["prompt_seed"], P, cfg["vocab_size"])
    n = len(gold["tokens"])
    gc = lnb.InferenceContext(gm, P + n + 1)
    _, first = gc.Forward(prompt, 0, want_logits=False)
    got, _ = gc.decode_greedy(first, P, n - 1)
    assert [first] + [int(t) for t in got] == gold["tokens"]
    assert gc.zseq_count() == 0                              # every softmax row certified its denominator (no serial walk)
    if n_layers == 8:                                        # ... and through the other forms: serial f64 denominator forced, throughput schedule
        # (ONE Forward again: the reference's mask is [seq, seq] broadcast over the context, llamatransformer.go:55-64 -- a prompt fed in chunks is a
        # different computation there, and here)
        g2 = lnb.InferenceContext(gm, P + 12).set_attention(-1, 1).set_schedule("throughput")
        _, f2 = g2.Forward(prompt, 0, want_logits=False)
        more, _ = g2.decode_greedy(f2, P, 8)
        assert [f2] + [int(t) for t in more] == gold["tokens"][:9]
        g2.close()
    gc.close(); gm.close()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint16 if a.dtype == np.uint16 else np.uint32)


@pytest.mark.parametrize("shape", ["tiny", "8b-2l"])
def test_throughput_schedule_is_bit_identical(lnb, shape):
    """lnb_ctx_set_schedule: wq|wk|wv on 128-step stages, wo / w2 on the self-feeding row-broadcast kernel -- tokens, logits bits and KV bits of the
    latency and the throughput forms agree with each other and (tiny) with the oracle; switching on one context mid-run keeps the history."""
    if shape == "tiny":
        cfg, P, N = TINY, 12, 12
    else:
        cfg, P, N = dict(orc.LLAMA_8B, n_layers=2), 40, 10      # the 8B block shape: gemv_quad_kernel<24, ...> and rowcast_lds_kernel are what `latency` runs
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    prompt = lnb.synth_tokens(5, P, cfg["vocab_size"])
    runs = {}
    for sched in ("latency", "throughput"):
        gc = lnb.InferenceContext(gm, P + N + 4).set_schedule(sched)
        _, first = gc.Forward(prompt, 0, want_logits=False)
        got, _ = gc.decode_greedy(first, P, N)
        lg, am = gc.Forward(np.array([int(got[-1])], dtype=np.int32), P + N)      # one eager step: its logits, bit for bit
        runs[sched] = ([first] + [int(t) for t in got], _bits(lg).copy(), am, [gc.CacheK(l).copy() for l in range(cfg["n_layers"])], [gc.CacheV(l).copy() for l in range(cfg["n_layers"])])
        gc.close()
    a, b = runs["latency"], runs["throughput"]
    assert a[0] == b[0] and (a[1] == b[1]).all() and a[2] == b[2]
    for l in range(cfg["n_layers"]):
        assert (a[3][l][:P + N + 1] == b[3][l][:P + N + 1]).all() and (a[4][l][:P + N + 1] == b[4][l][:P + N + 1]).all()
    if shape == "tiny":
        om = orc.Model(**cfg).fill_synthetic(1234).finalize()
        ref, _ = orc.Context(om, P + N + 4).generate(prompt, N + 1)
        assert a[0] == [int(t) for t in ref]
        om.close()
    # one context, switched between the two mid-generation
    gc = lnb.InferenceContext(gm, P + N + 4)
    _, first = gc.Forward(prompt, 0, want_logits=False)
    h1, _ = gc.decode_greedy(first, P, 4)
    gc.set_schedule("throughput")
    h2, _ = gc.decode_greedy(int(h1[-1]), P + 4, 4)
    gc.set_schedule("latency")
    h3, _ = gc.decode_greedy(int(h2[-1]), P + 8, N - 8)
    assert [first] + [int(t) for t in h1] + [int(t) for t in h2] + [int(t) for t in h3] == a[0]
    gc.close(); gm.close()


def test_stop_ids_only_count_for_the_until_entry_points(lnb):
    """lnb_decode_greedy / lnb_batch_decode promise n_steps tokens -- with stop ids on the context they used to freeze
    silently and return stale log entries.  They now ignore the ids; the _until forms honour them and say so per sequence; a chunked batch
    keeps an ended sequence frozen (start_pos < 0) instead of restarting it from its stop token."""
    om = orc.Model(**TINY).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize().enable_batch()
    n, steps = 4, 25                                          # (five chunks of five below)
    prompts = [orc.synth_tokens(300 + s, 6 + s, TINY["vocab_size"]) for s in range(n)]
    refs = [[int(t) for t in orc.Context(om, 64).generate(prompts[s], steps + 2)[0]] for s in range(n)]
    # single sequence: the stop id is on the context, lnb_decode_greedy runs through it and every token is the oracle's
    k = next(i for i in range(4, 16) if refs[0][i] not in refs[0][:i])
    gc = lnb.InferenceContext(gm, 64).set_stop_ids([refs[0][k]])
    _, first = gc.Forward(prompts[0], 0, want_logits=False)
    got, _ = gc.decode_greedy(first, len(prompts[0]), steps)
    assert [first] + [int(t) for t in got] == refs[0][:steps + 1]
    gc.reset()
    _, first = gc.Forward(prompts[0], 0, want_logits=False)
    out, fin, _ = gc.decode_greedy_until(first, len(prompts[0]), steps)
    assert fin and [first] + [int(t) for t in out] == refs[0][:k + 1]
    gc.close()
    # batch
    ctxs = [lnb.InferenceContext(gm, 64) for _ in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    ks = [next((i for i in range(3 + 2 * s, steps - 4) if refs[s][i] not in refs[s][:i]), None) if s != 2 else None for s in range(n)]
    for s in range(n):
        ctxs[s].set_stop_ids([] if ks[s] is None else [refs[s][ks[s]]])
    b = lnb.Batch(ctxs)
    plain, _ = b.decode(firsts, [len(p) for p in prompts], steps)                  # ignores the ids
    for s in range(n):
        assert [int(t) for t in plain[s]] == refs[s][1:steps + 1], s
    for c, p in zip(ctxs, prompts):                                               # again from the prompts, now in chunks of 5 with the ids honoured
        c.reset(); c.Forward(p, 0, want_logits=False)
    toks, pos, done = list(firsts), [len(p) for p in prompts], [False] * n
    outs = [[] for _ in range(n)]
    for _ in range(0, steps, 5):
        chunk, _ = b.decode_until(toks, [-1 if done[s] else pos[s] for s in range(n)], 5)
        for s in range(n):
            if done[s]:
                assert len(chunk[s]) == 0 and b.finished[s]                      # frozen: nothing generated, still reported as finished
                continue
            outs[s] += [int(t) for t in chunk[s]]
            pos[s] += len(chunk[s]); toks[s] = outs[s][-1]; done[s] = b.finished[s]
    for s in range(n):
        want = refs[s][1:steps + 1] if ks[s] is None else refs[s][1:ks[s] + 1]
        assert outs[s] == want and done[s] == (ks[s] is not None), (s, ks[s])
    # an ended sequence's context is exactly where the reference would be after emitting the stop token: it goes on alone from there
    s0 = next(s for s in range(n) if ks[s] is not None)
    with pytest.raises(lnb.LnbError, match="live batch"):
        ctxs[s0].close()                                                          # (the refusal is reported, the handle kept)
    b.close()
    ctxs[s0].set_stop_ids([])
    more, _ = ctxs[s0].decode_greedy(outs[s0][-1], len(prompts[s0]) + len(outs[s0]), 3)
    assert [int(t) for t in more] == refs[s0][ks[s0] + 1:ks[s0] + 4]
    for c in ctxs:
        c.close()
    gm.close(); om.close()


def test_batched_ticks_of_a_multi_stage_pipeline_refuse_stop_ids(lnb):
    """under lnb_pipeline_tick_batch only the last stage sees the token; the other stages would keep advancing.  The set-up call
    of a stage batch refuses contexts that carry stop ids; a one-stage pipe (whole model) honours them."""
    g0 = lnb.LlamaTransformer(part_begin=0, part_end=3, **TINY).fill_synthetic(3).finalize().enable_batch()
    c0 = [lnb.InferenceContext(g0, 32) for _ in range(2)]
    c0[1].set_stop_ids([7])
    b0 = lnb.Batch(c0)
    with pytest.raises(lnb.LnbError, match="stop ids"):
        b0.set_state([1, 2], [0, 0])
    c0[1].set_stop_ids([])
    b0.set_state([1, 2], [0, 0])
    b0.close()
    for c in c0:
        c.close()
    g0.close()


def test_cycle_stamps_of_a_gemv_launch(lnb):
    """lnb_profile_kernel_stamps (bench.py: roofline.measured_model): a chain wave is marked by stamp 6 (main-loop start), stamp 7 is the launch on
    the wall clock; the shader clock derived from the two lies where an MI355X can run"""
    cfg = dict(orc.LLAMA_8B, n_layers=2)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    gc = lnb.InferenceContext(gm, 64)
    gc.Forward(lnb.synth_tokens(1, 16, cfg["vocab_size"]), 0, want_logits=False)
    for which in (0, 2, 3, 4, 5):
        v, khz = gc.profile_kernel_stamps(which, 20)
        assert khz > 0
        cw = next(w for w in range(8) if v[w][0] > 0 and v[w][12] > 0)
        assert v[cw][0] >= 200                                   # the 8B shapes: one workgroup per CU (the output product: 256 persistent ones)
        ghz = v[cw][1] / (v[cw][13] / khz * 1e3) * 1e-3
        assert 1.0 < ghz < 2.6, (which, ghz)
        assert 0 < v[cw][12] < v[cw][1] <= v[cw][2]
    gc.close(); gm.close()


def test_prefill_and_large_batches_run_from_the_resident_layouts(lnb):
    """Round 5: gemm_stream_kernel reads the RESIDENT weight layouts (row-broadcast units = M16 units in another order; chain-layout units transposed
    over the wave's rows with v_permlane16/32_swap), so a prompt and a batch of more than 32 sequences need no second copy of the weights.
    Tiny model: prefill logits bits = oracle; 40 / 20 / 5 sequences batched WITHOUT lnb_model_enable_batch (every product as rows of the streaming kernel) =
    their oracle runs, and the same contexts go on with the column forms once the copy exists."""
    import subprocess, sys
    om = orc.Model(**TINY).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize()
    assert gm.batch_bytes() == 0
    for rows in (16, 37, 64):
        toks = orc.synth_tokens(7, rows, TINY["vocab_size"])
        gc = lnb.InferenceContext(gm, 128)
        lg, am = gc.Forward(toks, 0)
        lo, ao = orc.Context(om, 128).forward(toks, 0)
        assert (_bits(lg) == _bits(lo)).all() and am == ao, rows
        gc.close()
    n, steps = 40, 6
    prompts = [orc.synth_tokens(900 + s, 5 + s % 7, TINY["vocab_size"]) for s in range(n)]
    ctxs = [lnb.InferenceContext(gm, 32) for _ in range(n)]
    firsts = [ctxs[s].Forward(prompts[s], 0, want_logits=False)[1] for s in range(n)]
    refs = {}
    for cnt in (40, 20, 5):                                          # (more than 32: rows anyway; 17..32 and <= 16: the rows form INSTEAD of the column forms)
        for c, p in zip(ctxs[:cnt], prompts):
            c.reset(); c.Forward(p, 0, want_logits=False)
        b = lnb.Batch(ctxs[:cnt])
        assert gm.batch_bytes() == 0
        got, _ = b.decode(firsts[:cnt], [len(p) for p in prompts[:cnt]], steps)
        for s in range(0, cnt, 3):
            if s not in refs:
                refs[s] = [int(t) for t in orc.Context(om, 32).generate(prompts[s], steps + 3)[0]]
            assert [firsts[s]] + [int(t) for t in got[s]] == refs[s][:steps + 1], (cnt, s)
        b.close()
    gm.enable_batch()                                                # ... and the column forms pick the same caches up
    b = lnb.Batch(ctxs[:5])
    more, _ = b.decode([int(got[s][-1]) for s in range(5)], [len(prompts[s]) + steps for s in range(5)], 2)      # (got: the five-sequence run just above)
    for s in (0, 3):
        assert [int(t) for t in more[s]] == refs[s][steps + 1:steps + 3], s
    b.close()
    for c in ctxs:
        c.close()
    gm.close(); om.close()
