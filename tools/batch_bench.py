#!/usr/bin/env python3
"""Batched exact decode of the 8B shape in isolation (run under rocprofv3 by tools/gpu_profile_r03.sh): n prompts of --prompt-len tokens,
then --steps batched greedy steps; prints one JSON line (aggregate tokens/s, per-kernel-class times of the batched step)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llama-nuts-and-bolts_amd")]
import lnb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--prompt-len", type=int, default=128)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--profile-iters", type=int, default=0)
ap.add_argument("--no-copy", action="store_true", help="batches from the resident chain layouts (no lnb_model_enable_batch: weights_second_copy_bytes = 0)")
a = ap.parse_args()
lnb.build()
cfg = dict(lnb.LLAMA_8B, n_layers=a.layers)
m = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
if not a.no_copy:
    m.enable_batch()
ctxs = [lnb.InferenceContext(m, a.prompt_len + a.steps + 12) for _ in range(a.n)]
firsts = [c.Forward(lnb.synth_tokens(99 + s, a.prompt_len, cfg["vocab_size"]), 0, want_logits=False)[1] for s, c in enumerate(ctxs)]
b = lnb.Batch(ctxs)
warm, _ = b.decode(firsts, [a.prompt_len] * a.n, 4)
t0 = time.perf_counter()
got, ms = b.decode([int(w[-1]) for w in warm], [a.prompt_len + 4] * a.n, a.steps)
wall = time.perf_counter() - t0
res = {"n": a.n, "second_copy": not a.no_copy, "LNB_GS_NTW": os.environ.get("LNB_GS_NTW", ""), "steps": a.steps, "layers": a.layers, "tokens_per_s": round(a.n * a.steps / wall, 1), "ms_per_step": round(1e3 * wall / a.steps, 4), "hip_event_ms_per_step": round(ms / a.steps, 4)}
if a.profile_iters:
    names = ["norm+wqkv+rope", "attention", "wo+residual", "norm+w1|w3+silu", "w2+residual", "norm+output", "whole block"]
    res["kernels_us"] = {names[w]: round(1e3 * b.profile_kernel(w, a.prompt_len + 4, a.profile_iters), 2) for w in range(7)}
print(json.dumps(res))
