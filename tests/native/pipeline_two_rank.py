"""Two pipeline ranks (gloo) sharing ONE GPU: each owns half of the tiny model's blocks behind the C ABI (pipeline.LnbStage), the
hidden state and the token ring go through pipeline.run_ticks exactly as in bench.py --gpus 2 (RCCL there, gloo + host staging
here).  The last rank checks every generated token against the CPU oracle's greedy loop.  Launched by tests/test_gpu_parity.py."""
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
dist.destroy_process_group()
