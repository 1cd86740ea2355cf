#!/usr/bin/env python3
"""Generates tests/golden/<name>_logits.json: SHA-256 of the raw f32 bits of every logits row the CPU ORACLE produces -- all rows of the prompt's Forward and the
row of each of the following greedy steps -- so that a device run is pinned to the oracle on EVERY logit of EVERY row, not only on the argmax
(llamatransformer.go:145-180: Forward returns [S, vocab] logits; inference.go:207-211 takes the argmax of the last row).

    python tests/golden/make_logits_hashes.py configs4 [n_steps=8] [out_dir]     # dim 8192 x 80 layers (141 GB in host memory), 16-token prompt
    python tests/golden/make_logits_hashes.py configs1 [n_steps=8] [out_dir]     # the 8B shape, 128-token prompt
(GPU box host, 64 threads: configs4 ~3 minutes, configs1 ~2.)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

which = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))
SEED_W, SEED_P = 1234, 99
if which == "configs4":
    cfg, P = dict(orc.LLAMA_8B, dim=8192, n_layers=80, n_heads=64, n_kv_heads=8, multiple_of=4096), 16
elif which == "configs1":
    cfg, P = dict(orc.LLAMA_8B), 128
elif which == "tiny":                                      # (the script's own smoke run)
    cfg, P = dict(orc.TINY), 16
else:
    sys.exit("configs4 | configs1")


def row_hash(row):
    return hashlib.sha256(np.ascontiguousarray(row, dtype=np.float32).view(np.uint32).astype("<u4").tobytes()).hexdigest()


t0 = time.time()
om = orc.Model(**cfg).fill_synthetic(SEED_W).finalize()
prompt = orc.synth_tokens(SEED_P, P, cfg["vocab_size"])
oc = orc.Context(om, P + K + 2)
lg, tok = oc.forward(prompt, 0)
prompt_rows = [row_hash(lg[i]) for i in range(P)]
steps, toks = [], [int(tok)]
for k in range(K):
    lg, nxt = oc.forward(np.array([toks[-1]], dtype=np.int32), P + k)
    steps.append({"input_token": toks[-1], "position": P + k, "logits_sha256": row_hash(lg[0]), "argmax": int(nxt)})
    toks.append(int(nxt))
out = {"what": "SHA-256 of the raw f32 bits of the oracle's logits rows: %s shape %s, synthetic weights seed %d, prompt synth_tokens(%d, %d, vocab); all %d prompt rows of ONE Forward, then %d greedy one-token steps"
               % (which, {k: cfg[k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "multiple_of")}, SEED_W, SEED_P, P, P, K),
       "generator": "tests/golden/make_logits_hashes.py %s %d" % (which, K), "model": {k: cfg[k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "multiple_of")},
       "prompt_len": P, "weights_seed": SEED_W, "prompt_seed": SEED_P, "prompt_rows_logits_sha256": prompt_rows, "first_token": toks[0], "steps": steps,
       "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": oc.nthreads}
os.makedirs(OUT, exist_ok=True)
json.dump(out, open(os.path.join(OUT, "%s_logits.json" % which), "w"), indent=1)
print("wrote %d + %d row hashes (%s) in %.0f s; tokens %s" % (P, K, which, time.time() - t0, toks))
