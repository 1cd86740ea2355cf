#!/usr/bin/env python3
"""Generates tests/golden/configs2_<L>layer_tokens.json: the greedy continuation of the configs[2] workload (4096-token prompt, long-context
decode attention at T > 4096) as the CPU ORACLE computes it, on the Llama-3.1-8B shape cut to its first L layers (same dim 4096, 32/8
heads, head_dim 128, FFN 14336, vocab 128256, synthetic weights seed 1234, prompt synth_tokens(99, 4096, vocab)).

make_configs2_2layer_tokens.py made the two-layer file in this container (359 s on 8 vCPUs); this is the same script with the depth as an
argument, meant for a host with many cores -- the GPU box: `python tests/golden/make_configs2_cut_tokens.py 8` takes a few minutes on its 64
oracle threads; the full 32 layers would take about four times that again.  Depth matters because the two-layer cut exercises every
kernel but not the accumulation of a deep stack (activations after 8 or 32 residual blocks have a different dynamic range).
`bench.py --model llama8b-8l --prompt-len 4096` compares its tokens with the 8-layer file and refuses to print on a mismatch.

    python tests/golden/make_configs2_cut_tokens.py <n_layers> [n_tokens=100] [out_dir]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

L = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))
P, SEED_W, SEED_P = 4096, 1234, 99
cfg = dict(orc.LLAMA_8B, n_layers=L, max_seq_len=2304)          # 4608 RoPE rows: positions up to 4096 + N lie beyond the reference's table
t0 = time.time()
om = orc.Model(**cfg).fill_synthetic(SEED_W).finalize()
prompt = orc.synth_tokens(SEED_P, P, cfg["vocab_size"])
oc = orc.Context(om, P + N + 1)
toks, secs = oc.generate(prompt, N)
out = {"what": "oracle greedy continuation of the configs[2] workload on the %d-layer cut of the Llama-3.1-8B shape: synthetic weights seed %d, "
               "prompt synth_tokens(%d, %d, vocab), max_seq_len 2304 (4608 RoPE rows)" % (L, SEED_W, SEED_P, P),
       "generator": "tests/golden/make_configs2_cut_tokens.py %d" % L, "prompt_len": P, "n_layers": L, "weights_seed": SEED_W, "prompt_seed": SEED_P,
       "prompt_sha256": hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest(),
       "tokens": [int(t) for t in toks],
       "tokens_sha256": hashlib.sha256(toks.astype("<i4").tobytes()).hexdigest(),
       "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": oc.nthreads}
os.makedirs(OUT, exist_ok=True)
json.dump(out, open(os.path.join(OUT, "configs2_%dlayer_tokens.json" % L), "w"), indent=1)
print("wrote %d tokens of the %d-layer cut in %.0f s" % (len(toks), L, time.time() - t0))
