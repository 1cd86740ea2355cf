"""Round 6 (-m gpu, through the C ABI):
* configs[4] at depth: the committed oracle golden of a 10-layer cut of the 70B-like shape (= ONE stage of the 8-GPU pipeline) replayed device-only
  (VERDICT r5 task 4a: oracle evidence used to stop at 2 of 80 layers);
* configs[3] literally, on one GPU: 8 stages x 4 whole blocks of the 8B shape through the in-process transport, 512-token prompts (4 MiB hand-off per
  hop) + decode ticks for 2N = 16 sequences in flight, every sequence against its single-process device run (which the golden suite ties to the oracle)
  (VERDICT r5 task 5: the multi-stage tests ran the tiny shape only);
* lnb_runtime_info: the hardware queues the HIP runtime really gives the library's streams (VERDICT r5 task 7)."""
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


def test_committed_configs4_ten_layer_golden_is_reproduced_by_the_device(lnb):
    """BASELINE.json configs[4] (dim 8192, 64 / 8 heads, FFN 28672), first 10 layers + norm + output: 16-token prompt in one Forward (f32 matrix cores,
    resident layouts) + 16 greedy tokens through the one-token kernels at dim 8192 = the CPU oracle's tokens (tests/golden/make_configs4_cut_tokens.py;
    tests/test_golden_files.py checks on the CPU side that the file is there).  Also through the throughput kernel forms."""
    path = os.path.join(ROOT, "tests", "golden", "configs4_10layer_tokens.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/configs4_10layer_tokens.json not generated yet")
    gold = json.load(open(path))
    cfg = dict(orc.LLAMA_8B, **{k: gold["model"][k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")})
    assert cfg["dim"] == 8192 and cfg["n_layers"] == 10
    P, n = gold["prompt_len"], len(gold["tokens"])
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    assert gm.ffn_hidden == 28672
    prompt = lnb.synth_tokens(gold["prompt_seed"], P, cfg["vocab_size"])
    for sched in ("latency", "throughput"):
        gc = lnb.InferenceContext(gm, P + n + 1).set_schedule(sched)
        _, first = gc.Forward(prompt, 0, want_logits=False)
        got, _ = gc.decode_greedy(first, P, n - 1)
        assert [first] + [int(t) for t in got] == gold["tokens"], sched
        gc.close()
    gm.close()


def test_committed_configs4_eighty_layer_golden_is_reproduced_by_the_device(lnb):
    """BASELINE.json configs[4] at FULL depth: dim 8192 x 80 layers, 141 GB of synthetic weights on the one GPU, 16-token prompt + the CPU oracle's greedy continuation
    (tests/golden/configs4_80layer_tokens.json: made by make_configs4_cut_tokens.py 80 on the GPU box's host cores -- the oracle holds the whole model in memory).  Latency
    and throughput kernel forms, and the forced serial softmax denominator (llamatransformer.go:145-180 at the shape the 8-GPU pipeline shards)."""
    path = os.path.join(ROOT, "tests", "golden", "configs4_80layer_tokens.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/configs4_80layer_tokens.json not generated yet")
    gold = json.load(open(path))
    cfg = dict(orc.LLAMA_8B, **{k: gold["model"][k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")})
    assert cfg["dim"] == 8192 and cfg["n_layers"] == 80
    P, n = gold["prompt_len"], len(gold["tokens"])
    try:
        gm = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    except lnb.LnbError as e:                                # (a GPU with less than 141 GB free)
        pytest.skip("the 80-layer model does not fit: %s" % str(e)[:120])
    assert gm.weight_bytes() > 140e9
    prompt = lnb.synth_tokens(gold["prompt_seed"], P, cfg["vocab_size"])
    for label, setup in (("latency", lambda c: c), ("throughput", lambda c: c.set_schedule("throughput")), ("serial Z", lambda c: c.set_attention(-1, 1))):
        gc = setup(lnb.InferenceContext(gm, P + n + 1))
        _, first = gc.Forward(prompt, 0, want_logits=False)
        got, _ = gc.decode_greedy(first, P, n - 1)
        assert [first] + [int(t) for t in got] == gold["tokens"], label
        gc.close()
    gm.close()


@pytest.mark.parametrize("which", ["configs4", "configs1"])
def test_every_logit_of_every_row_at_full_depth_is_the_oracles(lnb, which):
    """Not only the argmax: SHA-256 of the raw f32 bits of EVERY logits row -- all rows of the prompt's Forward (f32 matrix cores) and the row of each of the next greedy
    one-token steps (the chain kernels) -- against the CPU oracle's (tests/golden/<which>_logits.json, made by tests/golden/make_logits_hashes.py on the GPU box's host:
    configs4 = dim 8192 x 80 layers, 141 GB; configs1 = the 8B shape, 128-token prompt).  llamatransformer.go:145-180 returns these rows; any rounding anywhere in the 80 / 32
    blocks that differed from the reference's would change a hash."""
    import hashlib
    path = os.path.join(ROOT, "tests", "golden", "%s_logits.json" % which)
    if not os.path.exists(path):
        pytest.skip("tests/golden/%s_logits.json not generated yet" % which)
    g = json.load(open(path))
    cfg = dict(orc.LLAMA_8B, **{k: g["model"][k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")})
    P, K = g["prompt_len"], len(g["steps"])
    try:
        gm = lnb.LlamaTransformer(**cfg).fill_synthetic(g["weights_seed"]).finalize()
    except lnb.LnbError as e:
        pytest.skip("the model does not fit: %s" % str(e)[:120])

    def row_hash(row):
        return hashlib.sha256(np.ascontiguousarray(row, dtype=np.float32).view(np.uint32).astype("<u4").tobytes()).hexdigest()

    gc = lnb.InferenceContext(gm, P + K + 2)
    lg, tok = gc.Forward(lnb.synth_tokens(g["prompt_seed"], P, cfg["vocab_size"]), 0)
    assert [row_hash(lg[i]) for i in range(P)] == g["prompt_rows_logits_sha256"] and int(tok) == g["first_token"]
    for k, st in enumerate(g["steps"]):
        assert int(tok) == st["input_token"]
        lg, tok = gc.Forward(np.array([tok], dtype=np.int32), P + k)
        assert row_hash(lg[0]) == st["logits_sha256"] and int(tok) == st["argmax"], k
    gc.close(); gm.close()


def test_configs3_literal_eight_stages_of_four_blocks_on_one_gpu(lnb):
    """llamatransformer.go:156-164 cut as BASELINE.json configs[3] says: 8 stages x 4 whole blocks of the Llama-3.1-8B shape, 512-token prompts
    ([512, 4096] bf16 = 4 MiB per hop), 2N = 16 sequences in flight on the N = 8 schedule (pipeline.run_ticks_native: rank r runs item t - 2r at
    tick t), every rank stepped tick by tick as 8 processes would, hand-offs through the in-process transport.  Every token of every sequence must
    equal the single-process device run of the same prompt on a whole-model handle (tokens of THAT path are the oracle's: tests/test_gpu_full_8b.py)."""
    import pipeline
    cfg = dict(lnb.LLAMA_8B)
    world, P, n_decode = 8, 512, 8
    n_seq = 2 * world
    cuts = [3 * 4 * r for r in range(world + 1)]              # thirds of a block: 4 whole blocks per stage
    stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **cfg).fill_synthetic(1234).finalize() for a, b in zip(cuts[:-1], cuts[1:])]
    assert sum(st.weight_bytes() for st in stages) > 15e9
    ctxs = [[lnb.InferenceContext(st, P + n_decode + 2).set_schedule("throughput") for _ in range(n_seq)] for st in stages]
    pipes = [lnb.Pipeline(stages[r], r, world, loopback_group="cfg3") for r in range(world)]
    prompts = [lnb.synth_tokens(99 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
    n_ticks = (1 + n_decode) * n_seq + 2 * (world - 1)
    state = [None] * world
    for t in range(n_ticks):
        for r in range(world):
            state[r] = pipeline.run_ticks_native(r, world, pipes[r], ctxs[r], prompts, n_decode, t, t + 1, state[r])
    for p_ in pipes:
        p_.sync()
    got = [[int(pipes[-1].read_tokens(q, 1)[0]) for q in state[-1]["slots"][s]] for s in range(n_seq)]
    for p_ in pipes:
        p_.close()
    for cs in ctxs:
        for c in cs:
            c.close()
    for st in stages:
        st.close()
    whole = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    for s in range(n_seq):
        wc = lnb.InferenceContext(whole, P + n_decode + 2)
        _, first = wc.Forward(prompts[s], 0, want_logits=False)
        more, _ = wc.decode_greedy(first, P, n_decode)
        assert got[s] == [first] + [int(t) for t in more], s
        wc.close()
    whole.close()
    # ... and every sequence against the CPU ORACLE's continuation of its 512-token prompt on the full 32-layer model (tests/golden/configs1_multi_P512_tokens.json:
    # made on the GPU box's host cores by tests/golden/make_multi_prompt_tokens.py 512 16 10; tests/test_golden_files.py checks that the file is there)
    gpath = os.path.join(ROOT, "tests", "golden", "configs1_multi_P512_tokens.json")
    if os.path.exists(gpath):
        g = json.load(open(gpath))
        assert g["prompt_len"] == P and [k for k in g["sequences"] if k < n_seq]
        for s in (k for k in g["sequences"] if k < n_seq):       # (the file is sparse: sequences 0, 4, 8, 12)
            m_ = min(len(got[s]), len(g["tokens"][str(s)]))
            assert m_ >= 1 + n_decode and got[s][:m_] == g["tokens"][str(s)][:m_], s


def test_configs4_literal_eight_stages_of_ten_blocks_on_one_gpu(lnb):
    """BASELINE.json configs[4] cut as it says -- the 70B-like shape (dim 8192 x 80 layers) as an 8-stage layer pipeline, 10 whole blocks per stage (17.1 GB each, + the
    2.1 GB head on the last), llamatransformer.go:156-164 sharded -- with all eight stages resident on ONE GPU (the same 141 GB), 2N = 16 sequences in flight on the N = 8
    schedule (pipeline.run_ticks_native), hand-offs of [rows, 8192] bf16 through the in-process transport, every rank stepped tick by tick as 8 processes would.  Sequence 0
    has the prompt of tests/golden/configs4_80layer_tokens.json: its tokens must be the CPU ORACLE's; every sequence must equal the single-process whole-model run."""
    import pipeline
    gpath = os.path.join(ROOT, "tests", "golden", "configs4_80layer_tokens.json")
    if not os.path.exists(gpath):
        pytest.skip("tests/golden/configs4_80layer_tokens.json not generated yet")
    gold = json.load(open(gpath))
    cfg = dict(orc.LLAMA_8B, **{k: gold["model"][k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")})
    world, P, n_decode = 8, gold["prompt_len"], 8
    n_seq = 2 * world
    cuts = [3 * 10 * r for r in range(world + 1)]             # thirds of a block: 10 whole blocks per stage
    try:
        stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **cfg).fill_synthetic(gold["weights_seed"]).finalize() for a, b in zip(cuts[:-1], cuts[1:])]
    except lnb.LnbError as e:
        pytest.skip("the eight stages do not fit: %s" % str(e)[:120])
    assert sum(st.weight_bytes() for st in stages) > 140e9
    ctxs = [[lnb.InferenceContext(st, P + n_decode + 2).set_schedule("throughput") for _ in range(n_seq)] for st in stages]
    pipes = [lnb.Pipeline(stages[r], r, world, loopback_group="cfg4") for r in range(world)]
    prompts = [lnb.synth_tokens(gold["prompt_seed"] + s, P, cfg["vocab_size"]) for s in range(n_seq)]
    n_ticks = (1 + n_decode) * n_seq + 2 * (world - 1)
    state = [None] * world
    for t in range(n_ticks):
        for r in range(world):
            state[r] = pipeline.run_ticks_native(r, world, pipes[r], ctxs[r], prompts, n_decode, t, t + 1, state[r])
    for p_ in pipes:
        p_.sync()
    got = [[int(pipes[-1].read_tokens(q, 1)[0]) for q in state[-1]["slots"][s]] for s in range(n_seq)]
    for p_ in pipes:
        p_.close()
    for cs in ctxs:
        for c in cs:
            c.close()
    for st in stages:
        st.close()
    assert got[0] == gold["tokens"][:1 + n_decode]           # sequence 0 = the oracle's continuation of the full 80-layer model
    whole = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    for s in range(n_seq):
        wc = lnb.InferenceContext(whole, P + n_decode + 2)
        _, first = wc.Forward(prompts[s], 0, want_logits=False)
        more, _ = wc.decode_greedy(first, P, n_decode)
        assert got[s] == [first] + [int(t) for t in more], s
        wc.close()
    whole.close()


def test_configs4_literal_batched_ticks_eight_stages_without_the_second_copy(lnb):
    """The same literal cut of configs[4] (8 stages x 10 blocks of the dim-8192 shape on one GPU) with BATCHES as the unit that moves through the stages
    (lnb_pipeline_tick_batch; 2N = 16 groups of 16 sequences in flight = 256 generations): no stage has room for a second weight copy next to 141 GB, so every group step is
    one pass over the stage's RESIDENT layouts with the group's sequences as rows of gemm_stream_kernel (what `bench.py --gpus 8` runs on a rank whose copy does not fit).
    Sequence 0 of group 0 = the 80-layer ORACLE golden; a spread of the others = the single-process whole-model run."""
    import pipeline
    gpath = os.path.join(ROOT, "tests", "golden", "configs4_80layer_tokens.json")
    if not os.path.exists(gpath):
        pytest.skip("tests/golden/configs4_80layer_tokens.json not generated yet")
    gold = json.load(open(gpath))
    cfg = dict(orc.LLAMA_8B, **{k: gold["model"][k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")})
    world, P, n_decode, n = 8, gold["prompt_len"], 6, 16
    G = 2 * world
    cuts = [3 * 10 * r for r in range(world + 1)]
    try:
        stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **cfg).fill_synthetic(gold["weights_seed"]).finalize() for a, b in zip(cuts[:-1], cuts[1:])]
    except lnb.LnbError as e:
        pytest.skip("the eight stages do not fit: %s" % str(e)[:120])
    assert all(st.batch_bytes() == 0 for st in stages)
    ctxs = [[[lnb.InferenceContext(st, P + n_decode + 2) for _ in range(n)] for _ in range(G)] for st in stages]            # [rank][group][seq]
    pipes = [lnb.Pipeline(stages[r], r, world, loopback_group="cfg4b") for r in range(world)]
    prompts = [[lnb.synth_tokens(gold["prompt_seed"] + n * g + s, P, cfg["vocab_size"]) for s in range(n)] for g in range(G)]
    first_slots = {}
    for g in range(G):
        for s in range(n):
            for r in range(world):
                slot = pipeline.prefill_through_pipeline(r, world, pipes[r], ctxs[r][g][s], prompts[g][s])
            first_slots[(g, s)] = slot
    batches = [[lnb.Batch(ctxs[r][g]).set_state(None, [P] * n) for g in range(G)] for r in range(world)]
    n_ticks = n_decode * G + 2 * (world - 1)
    state = [None] * world
    for t in range(n_ticks):
        for r in range(world):
            state[r] = pipeline.run_ticks_native_batched(r, world, pipes[r], batches[r], n_decode, t, t + 1, state[r])
    for p_ in pipes:
        p_.sync()
    got = {}
    for g in range(G):
        steps = [pipes[-1].read_tokens(q, n) for q in state[-1]["slots"][g]]          # [step][seq]
        for s in range(n):
            got[(g, s)] = [int(pipes[-1].read_tokens(first_slots[(g, s)], 1)[0])] + [int(st_[s]) for st_ in steps]
    for r in range(world):
        for b in batches[r]:
            b.check_error(); b.close()
        pipes[r].close()
        for grp_ in ctxs[r]:
            for c in grp_:
                c.close()
        stages[r].close()
    assert got[(0, 0)] == gold["tokens"][:1 + n_decode]
    whole = lnb.LlamaTransformer(**cfg).fill_synthetic(gold["weights_seed"]).finalize()
    for g, s in [(0, s_) for s_ in range(n)] + [(g_, (5 * g_) % n) for g_ in range(1, G)]:
        wc = lnb.InferenceContext(whole, P + n_decode + 2)
        _, first = wc.Forward(prompts[g][s], 0, want_logits=False)
        more, _ = wc.decode_greedy(first, P, n_decode)
        assert got[(g, s)] == [first] + [int(t) for t in more], (g, s)
        wc.close()
    whole.close()


@pytest.mark.parametrize("copy", [False, True])
def test_sequences_of_a_128_batch_on_the_full_model_are_their_oracle_continuations(lnb, copy):
    """Batched exact decode at FULL depth (llamatransformer.go:215-254 for 128 generations at once): 128 prompts of 128 tokens on the 32-layer 8B shape, one pass over the
    weights per step for all of them (rows of gemm_stream_kernel; without and with the matrix-core copy), 1 + 11 tokens each -- sequences 0, 32, 64 and 96 against the CPU
    ORACLE's continuation of THEIR OWN prompts (tests/golden/configs1_multi_P128_tokens.json, tests/golden/make_multi_prompt_tokens.py on the GPU box's host cores; the
    other 124 are tied to these by tests/test_gpu_batch.py: every sequence of a batch = its own single run)."""
    gpath = os.path.join(ROOT, "tests", "golden", "configs1_multi_P128_tokens.json")
    if not os.path.exists(gpath):
        pytest.skip("tests/golden/configs1_multi_P128_tokens.json not generated yet")
    g = json.load(open(gpath))
    cfg, P, n, K = dict(lnb.LLAMA_8B), g["prompt_len"], 128, 11
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(g["weights_seed"]).finalize()
    if copy:
        gm.enable_batch()
    ctxs = [lnb.InferenceContext(gm, P + K + 8) for _ in range(n)]
    firsts = [c.Forward(lnb.synth_tokens(g["prompt_seed_base"] + s, P, cfg["vocab_size"]), 0, want_logits=False)[1] for s, c in enumerate(ctxs)]
    b = lnb.Batch(ctxs)
    got, _ = b.decode(firsts, [P] * n, K)
    assert len(g["sequences"]) >= 4 and max(g["sequences"]) < n
    for s in g["sequences"]:                                     # (sparse: 0, 32, 64, 96 -- one sequence in every quarter of the batch, i.e. in four different row groups of the products)
        assert [firsts[s]] + [int(t) for t in got[s]] == g["tokens"][str(s)][:1 + K], s
    b.close()
    for c in ctxs:
        c.close()
    gm.close()


def test_runtime_info_reports_the_queues_the_streams_really_get(lnb):
    info = lnb.runtime_info(0, probe_queues=True)
    assert info["abi_version"] == lnb.ABI_VERSION and info["n_cus"] >= 1 and info["arch"].startswith("gfx")
    assert info["shader_clock_khz"] > 0 and info["wall_clock_khz"] > 0
    want = int(os.environ.get("GPU_MAX_HW_QUEUES", "16"))
    assert info["hw_queues_env"] == want
    # torch (conftest / other tests) may have initialised HIP before the library was loaded in THIS process: then the library's default came too late
    # and the record says so; otherwise the measured number is what the environment asked for (within the probe's resolution: 32 / rounds)
    if not (info["hip_initialised_before_load"] and info["hw_queues_set_by_library"]):
        assert info["hw_queues_measured"] >= min(want, 8), info
    assert lnb.queue_warning(2, dict(info, hw_queues_measured=0, hw_queues_expected=4)) is None
    assert "4 hardware queues" in lnb.queue_warning(8, dict(info, hw_queues_measured=0, hw_queues_expected=4))


def _bits32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _xcfg(dim, n_heads, n_kv_heads, **kw):
    return dict(orc.TINY, dim=dim, n_heads=n_heads, n_kv_heads=n_kv_heads, max_seq_len=2304, **kw)


ONE_CFGS = {"h8kv2_hd64": _xcfg(512, 8, 2), "h16kv4_hd128": _xcfg(2048, 16, 4), "h4kv2_hd64_plain_grid": _xcfg(256, 4, 2), "h8kv8_hd32": _xcfg(256, 8, 8)}
ONE_PROMPTS = {"h8kv2_hd64": (40, 300, 511, 512, 700, 1100, 4100), "h16kv4_hd128": (300, 1100), "h4kv2_hd64_plain_grid": (200, 1030), "h8kv8_hd32": (520, 1500)}


@pytest.mark.parametrize("name", sorted(ONE_CFGS))
def test_one_launch_long_context_attention_every_form_against_the_oracle(lnb, name):
    """llamatransformer.go:409-514 for one-token calls, operations_impl.go:478-511: attn_one_kernel (round 6: scores + softmax + PV in ONE launch, the
    (head, slice) workgroups exchanging their share of the scores inside the launch) against the oracle -- logits bits, tokens, KV bits -- at contexts on
    both sides of 512 (below: every workgroup scores the whole row itself; above: the exchange), through the forced serial denominator, through the
    poll's time-out path (every workgroup scores every block itself) and against the two-launch form it replaces; then the captured greedy loop."""
    cfg = ONE_CFGS[name]
    om = orc.Model(**cfg).fill_synthetic(77).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(77).finalize()
    for P in ONE_PROMPTS[name]:
        toks = orc.synth_tokens(41000 + P, P, cfg["vocab_size"])
        oc = orc.Context(om, P + 8)
        _, tok0 = oc.forward(toks, 0, want_logits=False)
        ref, tok = [], tok0
        for i in range(4):
            lo, tok_n = oc.forward([tok], P + i)
            ref.append((lo, tok_n)); tok = tok_n
        # (threshold 0 = the long-context forms at every context; flags: 8 one launch, 9 + serial Z, 2 two launches, 4 time-out path, 5 both)
        for flags in (8, 9, 2, 3, 4, 5, 102, 103, 202):      # (1xx: the two launches with the round-2 PV kernel, LNB_ATTN_LAZY=0, instead of the lazily certified one;
            os.environ["LNB_ATTN_LAZY"] = "0" if flags // 100 == 1 else "1"      #  2xx: the scores launch NOT touching the V rows for the PV launch, LNB_ATTN_TOUCH=0 -- the default does)
            os.environ["LNB_ATTN_TOUCH"] = "0" if flags // 100 == 2 else "1"
            flags %= 100
            gc = lnb.InferenceContext(gm, P + 8).set_attention(0, flags)
            _, t0 = gc.Forward(toks, 0, want_logits=False)
            assert t0 == tok0
            tok = t0
            z0 = gc.zseq_count()                             # (head_dim 32: the prompt's rows go through the row-per-workgroup kernel, which counts its serial walks too)
            for i in range(4):
                lg, tg = gc.Forward(np.array([tok], dtype=np.int32), P + i)
                assert (_bits32(ref[i][0]) == _bits32(lg)).all() and tg == ref[i][1], (name, P, flags, i)
                tok = tg
            assert gc.zseq_count() - z0 == (4 * cfg["n_layers"] * cfg["n_heads"] if flags & 1 else 0), (name, P, flags)
            for layer in range(cfg["n_layers"]):
                assert (oc.cache(layer, 0)[:P + 4] == gc.CacheK(layer)[:P + 4]).all() and (oc.cache(layer, 1)[:P + 4] == gc.CacheV(layer)[:P + 4]).all()
            gc.close()
        os.environ.pop("LNB_ATTN_LAZY", None); os.environ.pop("LNB_ATTN_TOUCH", None)
        for flags in (8, 4, 2):                              # graph replays: the arrival counters carry over from launch to launch without a reset
            gc = lnb.InferenceContext(gm, P + 8).set_attention(0, flags)
            _, t0 = gc.Forward(toks, 0, want_logits=False)
            got, _ = gc.decode_greedy(t0, P, 4)
            assert [int(t) for t in got] == [r[1] for r in ref], (name, P, flags)
            gc.close()
        oc.close()
    gm.close(); om.close()


def test_two_contexts_in_one_launch_attention_never_wait_for_each_other(lnb):
    """two contexts on the LATENCY schedule (the default), both with the one-launch attention switched on, decoding at long context from two host threads: their one-launch attention kernels may
    each hold CUs the other's workgroups want -- the bounded poll + local recomputation must end both runs with the single-context tokens"""
    import threading
    cfg = dict(orc.LLAMA_8B, n_layers=2, max_seq_len=2304)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    P, N = 1500, 24
    prompts = [lnb.synth_tokens(7 + s, P, cfg["vocab_size"]) for s in range(2)]
    ref = []
    for s in range(2):
        c = lnb.InferenceContext(gm, P + N + 2)
        _, f = c.Forward(prompts[s], 0, want_logits=False)
        t, _ = c.decode_greedy(f, P, N)
        ref.append([f] + [int(x) for x in t]); c.close()
    ctxs = [lnb.InferenceContext(gm, P + N + 2).set_attention(-1, 8) for _ in range(2)]      # (the one-launch form is opt-in)
    firsts = [c.Forward(p_, 0, want_logits=False)[1] for c, p_ in zip(ctxs, prompts)]
    out = [None, None]

    def run(s):
        t, _ = ctxs[s].decode_greedy(firsts[s], P, N)
        out[s] = [firsts[s]] + [int(x) for x in t]
    th = [threading.Thread(target=run, args=(s,)) for s in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a one-launch attention kernel is still waiting"
    assert out == ref
    for c in ctxs:
        c.close()
    gm.close()


@pytest.mark.parametrize("heads,kv_heads,dim,rows", [(4, 2, 512, 37), (4, 4, 256, 83), (8, 2, 1024, 16), (4, 2, 512, 160), (8, 2, 512, 530)])
def test_two_query_tiles_per_wave_prefill_attention_ragged_rows(lnb, monkeypatch, heads, kv_heads, dim, rows):
    """attn_mfma2_kernel (round 6: 32 query rows per wave, two accumulator chains fed by one K operand; llamatransformer.go:409-514 for multi-row calls)
    forced on at every row count (LNB_ATTN_MFMA2=16; its default is 512 rows and up): row counts that leave tile B empty (16), partial (37, 83) or
    cross several workgroups (160, 530), head_dim 128 and 64, then a ragged chunk at start_pos > 0 (the reference's modulo-broadcast mask over
    T > S) and decode steps that read the cache it wrote -- logits bits against the oracle, and against the 16-row kernel it replaces."""
    cfg = dict(orc.TINY, n_heads=heads, n_kv_heads=kv_heads, dim=dim, n_layers=2, vocab_size=512, max_seq_len=1024)
    om = orc.Model(**cfg).fill_synthetic(31).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(31).finalize()
    toks = orc.synth_tokens(8, 2 * rows, cfg["vocab_size"])
    oc = orc.Context(om, 2 * rows + 8)
    ref = [oc.forward(toks[lo_:hi_], lo_) for lo_, hi_ in ((0, rows), (rows, 2 * rows))]
    for form in ("16", "0"):
        monkeypatch.setenv("LNB_ATTN_MFMA2", form)
        gc = lnb.InferenceContext(gm, 2 * rows + 8)
        for (lo_, hi_), (lo, ao) in zip(((0, rows), (rows, 2 * rows)), ref):
            lg, ag = gc.Forward(toks[lo_:hi_], lo_)
            assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag, (form, lo_)
        if form == "16":
            tok = ag
            for i in range(3):
                lo, to = oc.forward([tok], 2 * rows + i)
                lg, tg = gc.Forward(np.array([tok], dtype=np.int32), 2 * rows + i)
                assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and to == tg
                tok = to
        gc.close()
    oc.close(); gm.close(); om.close()


@pytest.mark.parametrize("heads,kv_heads,dim,rows", [(4, 2, 512, 37), (4, 4, 256, 83), (8, 2, 1024, 16), (4, 2, 512, 160), (8, 2, 512, 530)])
def test_prefill_attention_with_the_scores_kept_between_its_passes(lnb, monkeypatch, heads, kv_heads, dim, rows):
    """attn_mfma3_kernel (round 6, the default of an exact prefill: pass 1 keeps the sixteen-bit exp-table indices of its scores in the context's score_idx
    scratch, pass 2 reads them back instead of running the q.k chain again, look-ups issued a tile ahead; llamatransformer.go:409-514 for multi-row calls,
    operations_impl.go:478-511) against the oracle and against attn_mfma_kernel (LNB_ATTN_SIDX_MB=0), which computes the scores twice: odd and even tile
    counts, partial last tiles, head_dim 128 and 64, a ragged chunk at start_pos > 0 (the modulo-broadcast mask over T > S; the scratch grows between the two
    calls), a scratch too small for the call (1 MB cap: the old kernel runs, same bits), then decode steps that read the cache the prefill wrote."""
    cfg = dict(orc.TINY, n_heads=heads, n_kv_heads=kv_heads, dim=dim, n_layers=2, vocab_size=512, max_seq_len=1100)
    om = orc.Model(**cfg).fill_synthetic(33).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(33).finalize()
    toks = orc.synth_tokens(9, 2 * rows, cfg["vocab_size"])
    oc = orc.Context(om, 2 * rows + 8)
    ref = [oc.forward(toks[lo_:hi_], lo_) for lo_, hi_ in ((0, rows), (rows, 2 * rows))]
    monkeypatch.setenv("LNB_ATTN_SIDX_KEEP_MB", "0")        # the first one-token call gives the scratch back (default: only above 256 MB), the steps below run behind that release
    for cap in ("4096", "0", "1"):
        monkeypatch.setenv("LNB_ATTN_SIDX_MB", cap)
        gc = lnb.InferenceContext(gm, 2 * rows + 8)
        for (lo_, hi_), (lo, ao) in zip(((0, rows), (rows, 2 * rows)), ref):
            lg, ag = gc.Forward(toks[lo_:hi_], lo_)
            assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag, (cap, lo_)
            need = heads * ((rows + 15) // 16) * ((hi_ + 15) // 16) * 512                     # bytes of score indices this call keeps (lnb_api.cpp: check_call)
            assert gc.prefill_attention_form() == (3 if need <= (int(cap) << 20) else 1), (cap, lo_, need)      # lnb_ctx_prefill_attention_form: which kernel really ran
        for layer in range(cfg["n_layers"]):
            assert (oc.cache(layer, 0)[:2 * rows] == gc.CacheK(layer)[:2 * rows]).all() and (oc.cache(layer, 1)[:2 * rows] == gc.CacheV(layer)[:2 * rows]).all()
        if cap == "4096":
            tok = ag
            for i in range(3):
                lo, to = oc.forward([tok], 2 * rows + i)
                lg, tg = gc.Forward(np.array([tok], dtype=np.int32), 2 * rows + i)
                assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and to == tg
                tok = to
        gc.close()
    oc.close(); gm.close(); om.close()


@pytest.mark.parametrize("rows", [16, 40, 130])
def test_blgp_matrix_core_feed_from_the_chain_layouts_is_bit_exact(lnb, monkeypatch, rows):
    """gemm_blgp_kernel (round 6, opt-in LNB_GEMM_BLGP=2): v_mfma_f32_16x16x1_f32 with four WEIGHT blocks per wave and the B operand broadcast by BLGP -- every
    chain-layout product (wq|wk|wv + RoPE + KV append, gate|up + SiLU, output) of a multi-row Forward, ragged row and tile counts, against the oracle's logits bits
    (operations_lineartransform.go:46-65: the k-ordered chain), then decode steps on the cache it wrote."""
    cfg = dict(orc.TINY, n_layers=2, vocab_size=1000, max_seq_len=512)       # vocab 1000: a partial last tile group of the output product
    om = orc.Model(**cfg).fill_synthetic(11).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(11).finalize()
    toks = orc.synth_tokens(3, rows, cfg["vocab_size"])
    oc = orc.Context(om, rows + 8)
    lo, ao = oc.forward(toks, 0)
    monkeypatch.setenv("LNB_GEMM_BLGP", "2")
    gc = lnb.InferenceContext(gm, rows + 8)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    tok = ag
    for i in range(3):
        lo1, to = oc.forward([tok], rows + i)
        lg1, tg = gc.Forward(np.array([tok], dtype=np.int32), rows + i)
        assert (lo1.view(np.uint32) == lg1.view(np.uint32)).all() and to == tg
        tok = to
    gc.close(); oc.close(); gm.close(); om.close()


@pytest.mark.parametrize("rows,copy", [(64, False), (100, False), (200, True), (130, True)])
def test_two_weight_tiles_per_wave_in_the_prefill_products_is_bit_exact(lnb, monkeypatch, rows, copy):
    """gemm_stream_kernel's TT form (round 6: a wave of wo / w2 carries two neighbouring weight tiles x four batch tiles, the gate|up product's shape; on by itself
    for long prompts, forced here with LNB_GS_TT=1 + LNB_GS_NTW=4) from the row-broadcast layout and from the M16 copy, ragged row counts, against the oracle's logits
    bits (operations_lineartransform.go:46-65) and against the one-tile form (LNB_GS_TT=0)."""
    cfg = dict(orc.TINY, dim=384, n_heads=6, n_kv_heads=2, n_layers=2, vocab_size=1000, max_seq_len=512, multiple_of=128)       # (ffn hidden 1408 = 11 chunks of 128)
    om = orc.Model(**cfg).fill_synthetic(12).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(12).finalize()
    if copy:
        gm.enable_batch()
    toks = orc.synth_tokens(4, rows, cfg["vocab_size"])
    oc = orc.Context(om, rows + 8)
    lo, ao = oc.forward(toks, 0)
    monkeypatch.setenv("LNB_GS_NTW", "4")
    for tt in ("1", "0"):
        monkeypatch.setenv("LNB_GS_TT", tt)
        gc = lnb.InferenceContext(gm, rows + 8)
        lg, ag = gc.Forward(toks, 0)
        assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag, tt
        gc.close()
    oc.close(); gm.close(); om.close()


@pytest.mark.parametrize("quad", ["1", "2"])
def test_down_projection_on_one_quad_chain_wave_is_bit_exact(lnb, monkeypatch, quad):
    """rung (a') of the FFN ladder (profiles/r06_ffn_stream.md): w2 re-tiled [N/16][K/8][16][8] and run by gemv_quad_kernel<16, KS, 4, R> -- its 16 rows per
    workgroup on ONE quad_perm chain wave (LNB_RW_W2=16 LNB_W2_QUAD=1|2: 128- / 256-step stages).  A measurement form, but a reachable one: logits bits, tokens
    and the captured greedy loop against the oracle (operations_lineartransform.go:46-65 for llamatransformer.go:619)."""
    cfg = dict(orc.TINY, dim=512, n_heads=4, n_kv_heads=2, n_layers=2)        # ffn_hidden 1536 = 12 x 128 = 6 x 256 steps
    om = orc.Model(**cfg).fill_synthetic(5).finalize()
    monkeypatch.setenv("LNB_RW_W2", "16"); monkeypatch.setenv("LNB_W2_QUAD", quad)
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(5).finalize()
    assert gm.ffn_hidden % 256 == 0
    toks = orc.synth_tokens(9, 9, cfg["vocab_size"])
    oc, gc = orc.Context(om, 40), lnb.InferenceContext(gm, 40)
    lo, ao = oc.forward(toks, 0)
    lg, ag = gc.Forward(toks, 0)
    assert (lo.view(np.uint32) == lg.view(np.uint32)).all() and ao == ag
    tok = ag
    for i in range(4):
        lo1, to = oc.forward([tok], 9 + i)
        lg1, tg = gc.Forward(np.array([tok], dtype=np.int32), 9 + i)
        assert (lo1.view(np.uint32) == lg1.view(np.uint32)).all() and to == tg
        tok = to
    got, _ = gc.decode_greedy(tok, 13, 8)
    ref, _ = orc.Context(om, 40).generate(toks, 13)
    assert [int(t) for t in got] == [int(t) for t in ref[5:13]]
    ms = gc.profile_ffn_pair(20, 4, 5, 0)                    # (the pair-timing entry point itself: both kernels on two streams, w2 5 us behind)
    assert ms > 0
    gc.close(); oc.close(); gm.close(); om.close()
