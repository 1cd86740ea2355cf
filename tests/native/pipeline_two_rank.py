"""Two pipeline ranks (gloo) sharing ONE GPU: each owns half of the tiny model's blocks behind the C ABI (pipeline.LnbStage), the
hidden state and the token ring go through pipeline.run_ticks exactly as in bench.py --gpus 2 (RCCL there, gloo + host staging
here).  The last rank checks every generated token against the CPU oracle's greedy loop.  LNB_TEST_BATCH=nb: then the same with BATCHES of nb
sequences as the unit (pipeline.run_ticks_batched_torch).  Launched by tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llama-nuts-and-bolts_amd")]
import lnb  # noqa: E402
import pipeline  # noqa: E402
from oracle import oracle as orc  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dict(orc.TINY)
P, n_decode = 24, 9                                           # 24-row prefill: the matrix-core path on every stage
n_seq = world * int(os.environ.get("LNB_TEST_MULT", "2"))     # 2: the overlapped schedule of bench.py --gpus N; 1: lock step
cut = int(os.environ.get("LNB_TEST_CUT", "0"))               # > 0: rank 0 holds block parts [0, cut), rank 1 the rest (3 parts per block)
parts = None if cut <= 0 else ((0, cut) if rank == 0 else (cut, 3 * cfg["n_layers"]))
stage = pipeline.LnbStage(lnb, torch, cfg, rank, world, n_seq, P + n_decode + 8, 0, parts)
prompts = [lnb.synth_tokens(99 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
st = pipeline.run_ticks(rank, world, stage, dist, torch, prompts, n_decode, "cuda:0")
if rank == world - 1:
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    for s in range(n_seq):
        ref, _ = orc.Context(om, P + n_decode + 8).generate(prompts[s], 1 + n_decode)
        assert list(ref) == st["produced"][s], (s, list(ref), st["produced"][s])
    print("PIPELINE_TWO_RANK_OK", st["produced"][0][:4])
if rank == 0:
    assert all(len(r) == n_decode for r in st["received"])
dist.barrier()
stage.close()
nb = int(os.environ.get("LNB_TEST_BATCH", "0"))
if nb > 0:
    # the BATCHED tick of the same fallback: groups of nb sequences move through the two stages, one pass over a stage's weights per group and
    # step (a pipe without a transport + pipeline.run_ticks_batched_torch); every sequence of every group against the oracle's greedy loop
    G = 2 * world
    copy = os.environ.get("LNB_TEST_BATCH_COPY", "1") == "1"
    stage = pipeline.LnbStage(lnb, torch, cfg, rank, world, G * nb, P + n_decode + 8, 0, (0, 3) if rank == 0 else (3, 6))
    if copy:
        stage.model.enable_batch()
    prompts = [lnb.synth_tokens(199 + q, P, cfg["vocab_size"]) for q in range(G * nb)]
    firsts = pipeline.prefill_torch(rank, world, stage, dist, torch, prompts, "cuda:0")
    pipe = lnb.Pipeline(stage.model, rank, world, host_transport=True)
    assert pipe.comm_count() == 0
    bats = [lnb.Batch(stage.ctx[g * nb:(g + 1) * nb]).set_state(firsts[g * nb:(g + 1) * nb] if rank == 0 else None, [P] * nb) for g in range(G)]
    try:
        pipe.tick_batch(run=None, send=bats[0])
        raise AssertionError("a pipe without a transport accepted a send")
    except lnb.LnbError as e:
        assert "no transport" in str(e)
    sb = pipeline.run_ticks_batched_torch(rank, world, pipe, bats, dist, torch, n_decode, "cuda:0", cfg["dim"], 0, G * 3)      # in two windows, as bench.py runs it
    sb = pipeline.run_ticks_batched_torch(rank, world, pipe, bats, dist, torch, n_decode, "cuda:0", cfg["dim"], G * 3, None, sb)
    pipe.sync()
    for b in bats:
        b.check_error()
    if rank == world - 1:
        om = orc.Model(**cfg).fill_synthetic(1234).finalize()
        for q in range(G * nb):
            g, j = divmod(q, nb)
            got = [firsts[q]] + [int(pipe.read_tokens(sl + j, 1)[0]) for sl in sb["slots"][g]]
            ref, _ = orc.Context(om, P + n_decode + 8).generate(prompts[q], 1 + n_decode)
            assert list(ref) == got, (q, list(ref), got)
        print("PIPELINE_TWO_RANK_BATCHED_OK", G * nb)
    dist.barrier()
    for b in bats:
        b.close()
    pipe.close()
    stage.close()
dist.destroy_process_group()
