#!/usr/bin/env python3
"""Generates tests/golden/configs1_tokens.json: the greedy continuation of BASELINE.json configs[1] as the CPU ORACLE computes it
(Llama-3.1-8B shape, synthetic weights seed 1234, the 128-token synthetic prompt of bench.py = synth_tokens(99, 128, vocab)).

The oracle (oracle/lnb_oracle.c) is the C restatement of the Go reference path; it needs ~16 GB of host RAM and, on 8 cores,
about 1 s per token.  bench.py asserts the tokens its timed run produced against this file and tests/test_gpu_full_8b.py
compares the device's whole 128 + N run with it (and re-derives a prefix with the oracle on the GPU box).

    python tests/golden/make_configs1_tokens.py [n_tokens=288]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 288
P, SEED_W, SEED_P = 128, 1234, 99
t0 = time.time()
om = orc.Model(**orc.LLAMA_8B).fill_synthetic(SEED_W).finalize()
prompt = orc.synth_tokens(SEED_P, P, orc.LLAMA_8B["vocab_size"])
oc = orc.Context(om, P + N + 1)
toks, secs = oc.generate(prompt, N)
out = {"what": "oracle greedy continuation of configs[1]: Llama-3.1-8B shape, synthetic weights seed %d, prompt synth_tokens(%d, %d, vocab)" % (SEED_W, SEED_P, P),
       "generator": "tests/golden/make_configs1_tokens.py", "prompt_len": P, "weights_seed": SEED_W, "prompt_seed": SEED_P,
       "prompt_sha256": hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest(),
       "tokens": [int(t) for t in toks],
       "tokens_sha256": hashlib.sha256(toks.astype("<i4").tobytes()).hexdigest(),
       "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": oc.nthreads}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs1_tokens.json"), "w"), indent=1)
print("wrote %d tokens in %.0f s" % (len(toks), time.time() - t0))
