"""Full-size parity at BASELINE.json's configs[1], literally: Llama-3.1-8B shape, synthetic weights (seed 1234), the 128-token
synthetic prompt bench.py uses, then N greedy tokens (default 128, LNB_TEST_8B_NEW_TOKENS up to 287).

The device continuation must be token-id identical to the CPU oracle's (north_star).  The oracle's continuation of exactly this
run is committed as tests/golden/configs1_tokens.json (written by tests/golden/make_configs1_tokens.py: ~6 minutes of host time
for 288 tokens); on the GPU box the oracle is run again for the prefill (all 128 logits rows, expected bit-identical; within 1e-2
required) and the first LNB_TEST_8B_ORACLE_STEPS decode steps (default 24), so the golden file itself is re-derived where it is used.
The oracle needs ~16 GB of host RAM."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

N_PROMPT = 128
N_NEW = int(os.environ.get("LNB_TEST_8B_NEW_TOKENS", "128"))
N_ORACLE = int(os.environ.get("LNB_TEST_8B_ORACLE_STEPS", "24"))
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "configs1_tokens.json")))


def test_llama8b_configs1_128_token_prompt_continuation_is_token_identical():
    import lnb
    lnb.build()
    assert GOLD["prompt_len"] == N_PROMPT and N_NEW <= len(GOLD["tokens"])
    seq_len = N_PROMPT + N_NEW
    gm = lnb.LlamaTransformer(**lnb.LLAMA_8B).fill_synthetic(GOLD["weights_seed"]).finalize()
    assert gm.weight_bytes() > 16e9                                                # 15.0 GB streamed + 1.05 GB embedding table
    prompt = orc.synth_tokens(GOLD["prompt_seed"], N_PROMPT, 128256)
    assert (prompt == lnb.synth_tokens(GOLD["prompt_seed"], N_PROMPT, 128256)).all()
    gc = lnb.InferenceContext(gm, seq_len)
    lg_gpu, first = gc.Forward(prompt, 0, want_logits=True)
    rest, ms = gc.decode_greedy(first, N_PROMPT, N_NEW - 1)
    got = [first] + [int(t) for t in rest]
    assert got == GOLD["tokens"][:N_NEW], "first mismatch with the oracle's golden continuation at %d" % next(
        i for i, (a, b) in enumerate(zip(got, GOLD["tokens"])) if a != b)

    # re-deriving the golden file costs ~1 minute of a 64+-thread host (the oracle's 128-row prefill of the 8B shape is 1e12 scalar MACs);
    # on a small host it would take the better part of an hour, so there the committed oracle continuation above is the whole check
    if (os.cpu_count() or 1) < 24 and os.environ.get("LNB_TEST_8B_FORCE_ORACLE") != "1":
        gc.close(); gm.close()
        return
    om = orc.Model(**orc.LLAMA_8B).fill_synthetic(GOLD["weights_seed"]).finalize()
    # spot-check the device copy of two big tensors against the oracle's generator (layout round trip at scale)
    for name in ("layers.31.feed_forward.w2.weight", "layers.0.attention.wk.weight"):
        ref = om.get_tensor(name)
        assert (gm.get_tensor(name, ref.size) == ref).all(), name
    oc = orc.Context(om, seq_len)
    lg_cpu, first_cpu = oc.forward(prompt, 0, want_logits=True)
    assert np.abs(lg_cpu - lg_gpu).max() <= 1e-2                                  # north_star tolerance
    exact = float((lg_cpu.view(np.uint32) == lg_gpu.view(np.uint32)).mean())
    print("prefill logits bit-identical fraction: %.6f" % exact)
    assert exact == 1.0
    ref = [first_cpu]
    tok, pos = first_cpu, N_PROMPT
    for _ in range(min(N_ORACLE, N_NEW - 1)):
        _, tok = oc.forward([tok], pos, want_logits=False)
        ref.append(tok); pos += 1
    assert ref == GOLD["tokens"][:len(ref)] == got[:len(ref)]                      # the golden file, re-derived on this box
    for layer in (0, 31):
        assert (oc.cache(layer, 0)[:pos] == gc.CacheK(layer)[:pos]).all()
        assert (oc.cache(layer, 1)[:pos] == gc.CacheV(layer)[:pos]).all()
    print("decode %d steps: %.3f ms/token on device" % (N_NEW - 1, ms / (N_NEW - 1)))
    gc.close(); gm.close(); oc.close(); om.close()


def test_llama8b_long_prefill_on_the_matrix_cores_equals_the_row_by_row_path():
    """configs[2]-style size-independent property at the full 8B shape: a 512-token prefill through the f32-MFMA GEMMs must be
    bit-identical (last-row logits, next token, K/V of the first and last layer) to the same prefill done one row launch per
    token row by the S = 1 kernels (which the test above pins to the oracle).  The oracle itself would need ~5 minutes per
    512 tokens, so the two device paths are compared through their digests in separate processes (the switch is an env)."""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r'''
import hashlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import lnb
S = 512
m = lnb.LlamaTransformer(**lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, S + 8)
toks = lnb.synth_tokens(7, S, 128256)
lg, tok = c.Forward(toks, 0, want_logits=True)
h = hashlib.sha256()
h.update(np.ascontiguousarray(lg[-1]).tobytes())
for layer in (0, 31):
    h.update(c.CacheK(layer)[:S].tobytes()); h.update(c.CacheV(layer)[:S].tobytes())
nxt, _ = c.decode_greedy(tok, S, 4)
print("DIGEST", h.hexdigest(), tok, [int(t) for t in nxt])
''' % (root, os.path.join(root, "llama-nuts-and-bolts_amd"))
    outs = []
    for mfma in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, LNB_PREFILL_MFMA=mfma), timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1], outs
