#!/usr/bin/env python3
"""Generates tests/golden/configs4_<L>layer_tokens.json: the CPU ORACLE's greedy continuation on a cut of the configs[4] shape (BASELINE.json:
random-init Llama shape dim 8192, 64 / 8 heads, head_dim 128, FFN 28672, vocab 128256 -- "70B-like"), first L layers + norm + output.
L = 10 is ONE stage of the 8-GPU layer pipeline (80 blocks / 8 GPUs): 17 GB of bf16 matrices + 4.2 GB of embedding and head.
16-token prompt (one Forward on the f32 matrix cores) + 16 greedy tokens (the one-token kernels at dim 8192: wq|wk|wv as 40-row quad blocks,
w1|w3 as 112-row blocks, the fused RMSNorm over 8192 terms).  VERDICT r5 task 4(a): configs[4] evidence used to stop at 2 layers.

    python tests/golden/make_configs4_cut_tokens.py [n_layers=10] [n_tokens=17] [out_dir]
(minutes on the 8 vCPUs of the build container -- where the committed file was made -- or on the GPU box's host cores.)
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 17
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))
P, SEED_W, SEED_P = 16, 1234, 99
cfg = dict(orc.LLAMA_8B, dim=8192, n_layers=L, n_heads=64, n_kv_heads=8, multiple_of=4096)
t0 = time.time()
om = orc.Model(**cfg).fill_synthetic(SEED_W).finalize()
t_fill = time.time() - t0
prompt = orc.synth_tokens(SEED_P, P, cfg["vocab_size"])
oc = orc.Context(om, P + N + 1)
toks, secs = oc.generate(prompt, N)
out = {"what": "oracle greedy continuation on the %d-layer cut of the configs[4] shape (dim 8192, 64/8 heads, FFN 28672, vocab 128256): synthetic weights "
               "seed %d, prompt synth_tokens(%d, %d, vocab)" % (L, SEED_W, SEED_P, P),
       "generator": "tests/golden/make_configs4_cut_tokens.py %d %d" % (L, N), "prompt_len": P, "n_layers": L, "weights_seed": SEED_W, "prompt_seed": SEED_P,
       "model": {k: cfg[k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "multiple_of", "ffn_dim_multiplier", "max_seq_len")},
       "prompt_sha256": hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest(),
       "tokens": [int(t) for t in toks],
       "tokens_sha256": hashlib.sha256(toks.astype("<i4").tobytes()).hexdigest(),
       "oracle_seconds": round(time.time() - t0, 1), "weights_fill_seconds": round(t_fill, 1), "oracle_threads": oc.nthreads}
os.makedirs(OUT, exist_ok=True)
json.dump(out, open(os.path.join(OUT, "configs4_%dlayer_tokens.json" % L), "w"), indent=1)
print("wrote %d tokens of the %d-layer cut in %.0f s (weights %.0f s)" % (len(toks), L, time.time() - t0, t_fill))
