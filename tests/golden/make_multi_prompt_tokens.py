#!/usr/bin/env python3
"""Generates tests/golden/configs1_multi_P<P>_tokens.json: the CPU ORACLE's greedy continuations of MANY prompts on the full Llama-3.1-8B shape
(synthetic weights seed 1234) -- sequence s has the prompt bench.py, pipeline.py and the batch tests give it: synth_tokens(99 + s, P, vocab).
Sequence 0 at P = 128 is configs[1]'s prompt (tests/golden/configs1_tokens.json holds its long continuation).

What they pin: every sequence of a BATCH (bench.py sequences_in_flight_batched: 128 prompts per pass over the weights), of the sequences in flight of the
layer pipeline (2N prompts), and of configs[3]'s literal shape (8 stages x 4 blocks, 512-token prompts) -- against the oracle itself instead of against the
device's own single-sequence run.

    python tests/golden/make_multi_prompt_tokens.py <P> <n_seq> <n_tokens> [out_dir] [shards=4]
One oracle process per shard (64 threads each: the OpenMP team collapses beyond one socket's cores), the 16 GB model filled once per process.  P = 128,
128 sequences, 53 tokens: ~11 s per sequence; P = 512, 16 sequences, 9 tokens: ~35 s per sequence (GPU box host: 256 hardware threads).
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SEED_W, SEED_P0 = 1234, 99


def shape():
    from oracle import oracle as orc
    return orc.TINY if os.environ.get("LNB_GOLDEN_TINY") else orc.LLAMA_8B      # (LNB_GOLDEN_TINY=1: the script's own smoke run on a test shape)


def shard_main(P, first, count, N, path):
    from oracle import oracle as orc
    om = orc.Model(**shape()).fill_synthetic(SEED_W).finalize()
    out = {}
    for s in range(first, first + count):
        prompt = orc.synth_tokens(SEED_P0 + s, P, shape()["vocab_size"])
        oc = orc.Context(om, P + N + 1)
        toks, _ = oc.generate(prompt, N)
        oc.close()
        out[str(s)] = [int(t) for t in toks]
        json.dump(out, open(path, "w"))                      # (after every sequence: a cut-off run keeps what it has)
    om.close()


if __name__ == "__main__":
    if sys.argv[1] == "--shard":
        shard_main(*[int(v) for v in sys.argv[2:6]], sys.argv[6])
        sys.exit(0)
    P, n_seq, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.dirname(os.path.abspath(__file__))
    shards = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    per = (n_seq + shards - 1) // shards
    procs, parts = [], []
    for k in range(shards):
        first, count = k * per, max(0, min(per, n_seq - k * per))
        if count == 0:
            continue
        part = os.path.join(out_dir, "multi_P%d_part%d.json" % (P, k))
        parts.append(part)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--shard", str(P), str(first), str(count), str(N), part]))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.exit("a shard failed: %s" % rcs)
    toks = {}
    for part in parts:
        toks.update(json.load(open(part)))
        os.remove(part)
    from oracle import oracle as orc
    import numpy as np
    seqs = [toks[str(s)] for s in range(n_seq)]
    flat = np.array(seqs, dtype="<i4")
    prompts = np.stack([orc.synth_tokens(SEED_P0 + s, P, shape()["vocab_size"]) for s in range(n_seq)]).astype("<i4")
    out = {"what": "oracle greedy continuations of %d prompts on the full Llama-3.1-8B shape (32 layers): synthetic weights seed %d, prompt of sequence s = synth_tokens(%d + s, %d, vocab); "
                   "%d tokens each (the first one is the prefill's)" % (n_seq, SEED_W, SEED_P0, P, N),
           "generator": "tests/golden/make_multi_prompt_tokens.py %d %d %d" % (P, n_seq, N), "prompt_len": P, "n_seq": n_seq, "n_tokens": N,
           "weights_seed": SEED_W, "prompt_seed_base": SEED_P0, "prompts_sha256": hashlib.sha256(prompts.tobytes()).hexdigest(),
           "tokens": seqs, "tokens_sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
           "oracle_seconds": round(time.time() - t0, 1), "oracle_processes": len(parts), "oracle_threads_each": orc.default_threads()}
    json.dump(out, open(os.path.join(out_dir, "configs1_multi_P%d_tokens.json" % P), "w"))
    print("wrote %d x %d tokens (P = %d) in %.0f s" % (n_seq, N, P, time.time() - t0))
