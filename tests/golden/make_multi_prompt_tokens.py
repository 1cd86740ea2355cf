#!/usr/bin/env python3
"""Generates tests/golden/configs1_multi_P<P>_tokens.json: the CPU ORACLE's greedy continuations of MANY prompts on the full Llama-3.1-8B shape
(synthetic weights seed 1234) -- sequence s has the prompt bench.py, pipeline.py and the batch tests give it: synth_tokens(99 + s, P, vocab).
Sequence 0 at P = 128 is configs[1]'s prompt (tests/golden/configs1_tokens.json holds its long continuation).

What they pin: every sequence of a BATCH (bench.py sequences_in_flight_batched: 128 prompts per pass over the weights), of the sequences in flight of the
layer pipeline (2N prompts), and of configs[3]'s literal shape (8 stages x 4 blocks, 512-token prompts) -- against the oracle itself instead of against the
device's own single-sequence run.

    python tests/golden/make_multi_prompt_tokens.py <P> <n_seq> <n_tokens> [out_dir] [shards=2] [first_seq=0]      (merges into the file in out_dir)
One oracle process per shard, each pinned to its own 64 logical CPUs (64 threads each: the OpenMP team collapses beyond one socket's cores), the 16 GB model filled once per process.  The oracle
sustains 15-18 GMAC/s with 64 threads on the GPU box's host (7 GMAC per token row): ~80 s for a 128-token prompt + 55 tokens, ~4 min for a 512-token prompt + 9; the
committed files hold four sequences each (a first run of four unpinned processes collapsed to one sequence per 25 minutes and was assembled with --assemble).
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


def shard_main(P, first, count, N, path, k=0):
    # shard k keeps to its own 64 logical CPUs (0-63, 64-127, ...: on the 2 x 64-core hosts the first 128 are the physical cores of the two sockets): four unpinned
    # 64-thread teams on 128 cores trip over each other's spinning barriers (the first attempt of this script: > 25 min for what one team does in 8)
    try:
        ncpu = os.cpu_count() or 1
        lo = (64 * k) % max(64, ncpu - ncpu % 64)
        os.sched_setaffinity(0, range(lo, min(lo + 64, ncpu)))
    except (AttributeError, OSError, ValueError):
        pass
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


def write_golden(P, toks, out_dir, secs, nproc, note=None):
    """toks: {sequence index (str): tokens}.  The file is SPARSE and RAGGED: it lists the sequences it holds, each with as many tokens as its run made, and a later run
    MERGES into the file that is there (same P, same seeds)."""
    from oracle import oracle as orc
    import numpy as np
    path = os.path.join(out_dir, "configs1_multi_P%d_tokens.json" % P)
    old = json.load(open(path)) if os.path.exists(path) else None
    if old and old["prompt_len"] == P and old["weights_seed"] == SEED_W and old["prompt_seed_base"] == SEED_P0:
        for k, v in old["tokens"].items():
            if len(v) > len(toks.get(k, [])):
                toks[k] = v
    ids = sorted(int(k) for k in toks)
    flat = np.concatenate([np.array(toks[str(k)], dtype="<i4") for k in ids])
    prompts = np.stack([orc.synth_tokens(SEED_P0 + k, P, shape()["vocab_size"]) for k in ids]).astype("<i4")
    runs = (old.get("runs", [{"oracle_seconds": old.get("oracle_seconds"), "oracle_processes": old.get("oracle_processes"), "note": old.get("note")}]) if old else []) + \
           [{"oracle_seconds": round(secs, 1), "oracle_processes": nproc, "note": note}]
    out = {"what": "oracle greedy continuations of %d prompts on the full Llama-3.1-8B shape (32 layers): synthetic weights seed %d, prompt of sequence s = synth_tokens(%d + s, %d, vocab); "
                   "the first token of a sequence is the prefill's; sequences %s" % (len(ids), SEED_W, SEED_P0, P, ids),
           "generator": "tests/golden/make_multi_prompt_tokens.py %d ..." % P, "prompt_len": P, "sequences": ids, "n_tokens": {str(k): len(toks[str(k)]) for k in ids},
           "weights_seed": SEED_W, "prompt_seed_base": SEED_P0, "prompts_sha256": hashlib.sha256(prompts.tobytes()).hexdigest(),
           "tokens": {str(k): toks[str(k)] for k in ids}, "tokens_sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
           "runs": runs, "oracle_threads_each": orc.default_threads()}
    json.dump(out, open(path, "w"))
    print("wrote %d sequences (P = %d, %s) in %.0f s" % (len(ids), P, ids, secs))


if __name__ == "__main__":
    if sys.argv[1] == "--assemble":                          # --assemble <P> <out_dir> <part files...>: what a cut-off run left behind
        P = int(sys.argv[2]); toks = {}
        for part in sys.argv[4:]:
            toks.update(json.load(open(part)))
        write_golden(P, toks, sys.argv[3], 0.0, len(sys.argv[4:]), "assembled from part files")
        sys.exit(0)
    if sys.argv[1] == "--shard":
        shard_main(*[int(v) for v in sys.argv[2:6]], sys.argv[6], int(sys.argv[7]) if len(sys.argv) > 7 else 0)
        sys.exit(0)
    P, n_seq, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.dirname(os.path.abspath(__file__))
    shards = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    first0 = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # (sequences first0 .. first0 + n_seq - 1)
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    per = (n_seq + shards - 1) // shards
    procs, parts = [], []
    for k in range(shards):
        first, count = first0 + k * per, max(0, min(per, n_seq - k * per))
        if count == 0:
            continue
        part = os.path.join(out_dir, "multi_P%d_part%d.json" % (P, k))
        parts.append(part)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--shard", str(P), str(first), str(count), str(N), part, str(k)]))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.exit("a shard failed: %s" % rcs)
    toks = {}
    for part in parts:
        toks.update(json.load(open(part)))
        os.remove(part)
    write_golden(P, toks, out_dir, time.time() - t0, len(parts), "sequences %d..%d, %d tokens each" % (first0, first0 + n_seq - 1, N))
