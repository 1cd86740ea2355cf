#!/usr/bin/env python3
"""Generates tests/golden/configs2_2layer_tokens.json: the greedy continuation of the configs[2] workload (4096-token prompt, long-context
decode attention at T > 4096) as the CPU ORACLE computes it, on the Llama-3.1-8B shape CUT TO TWO LAYERS (same dim 4096, 32/8 heads,
head_dim 128, FFN 14336, vocab 128256, synthetic weights seed 1234, prompt synth_tokens(99, 4096, vocab)).

The full 32-layer model at this prompt length is out of the oracle's reach on a test box (about 30 T MAC for the prefill), the
two-layer cut runs every kernel of the path at the full head geometry -- the (H & 7) == 0 XCD-remap branch of the attention grids, the
long-context scores / PV kernels at T = 4097.., the 4096-row matrix-core prefill.  `bench.py --model llama8b-2l --prompt-len 4096`
compares its tokens with this file and refuses to print a number on a mismatch; tests/test_gpu_round3.py re-derives them.

    python tests/golden/make_configs2_2layer_tokens.py [n_tokens=100]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
P, SEED_W, SEED_P = 4096, 1234, 99
cfg = dict(orc.LLAMA_8B, n_layers=2, max_seq_len=2304)          # 4608 RoPE rows: positions up to 4096 + N lie beyond the reference's table
t0 = time.time()
om = orc.Model(**cfg).fill_synthetic(SEED_W).finalize()
prompt = orc.synth_tokens(SEED_P, P, cfg["vocab_size"])
oc = orc.Context(om, P + N + 1)
toks, secs = oc.generate(prompt, N)
out = {"what": "oracle greedy continuation of the configs[2] workload on the two-layer cut of the Llama-3.1-8B shape: synthetic weights seed %d, "
               "prompt synth_tokens(%d, %d, vocab), max_seq_len 2304 (4608 RoPE rows)" % (SEED_W, SEED_P, P),
       "generator": "tests/golden/make_configs2_2layer_tokens.py", "prompt_len": P, "n_layers": 2, "weights_seed": SEED_W, "prompt_seed": SEED_P,
       "prompt_sha256": hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest(),
       "tokens": [int(t) for t in toks],
       "tokens_sha256": hashlib.sha256(toks.astype("<i4").tobytes()).hexdigest(),
       "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": oc.nthreads}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs2_2layer_tokens.json"), "w"), indent=1)
print("wrote %d tokens in %.0f s" % (len(toks), time.time() - t0))
