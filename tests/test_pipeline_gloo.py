"""world_size-2 (and 3) CPU test of the layer-pipeline schedule (llama-nuts-and-bolts_amd/pipeline.py) over gloo with a
pure-python stage: checks the tick schedule, the grouped isend/irecv pairing (no deadlock), the token ring back to
rank 0, KV-like per-sequence state and the split untimed/timed windows bench.py uses -- against a single-process
evaluation of the same staged function."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import pipeline  # noqa: E402

DIM, VOCAB, P = 8, 97, 5


class FakeStage(pipeline.Stage):
    """stage r of `world`: a deterministic integer 'layer stack' with per-sequence state (a stand-in for the KV cache)"""

    def __init__(self, rank, world, n_seq):
        self.rank, self.world = rank, world
        self.buf = {s: torch.zeros(P, DIM, dtype=torch.int16) for s in range(n_seq)}
        self.kv = {s: 0 for s in range(n_seq)}
        self.log = []

    def hidden_buffer(self, seq, rows):
        return self.buf[seq][:rows]

    def run(self, seq, rows, start_pos, tokens):
        h = self.buf[seq][:rows]
        if tokens is not None:
            assert self.rank == 0
            for i, t in enumerate(tokens):
                h[i] = torch.arange(DIM, dtype=torch.int16) * 3 + int(t) % 50 + start_pos + i
        self.kv[seq] = (self.kv[seq] * 31 + int(h.to(torch.int64).sum()) + start_pos) % 1009
        h.copy_(((h.to(torch.int32) * (self.rank + 2) + self.kv[seq]) % 251).to(torch.int16))
        self.log.append((seq, rows, start_pos))
        if self.rank == self.world - 1:
            return (int(h[rows - 1].to(torch.int64).sum()) * 7 + seq) % VOCAB
        return None


def reference(world, prompts, n_decode):
    stages = [FakeStage(r, world, len(prompts)) for r in range(world)]
    out = []
    for s, pr in enumerate(prompts):
        toks, cur, pos, rows = [], np.asarray(pr, dtype=np.int32), 0, len(pr)
        for step in range(n_decode + 1):
            h = None
            for r, st in enumerate(stages):
                if r > 0:
                    st.buf[s][:rows] = h
                t = st.run(s, rows, pos, cur if r == 0 else None)
                h = st.buf[s][:rows].clone()
            toks.append(t)
            pos, rows, cur = (len(pr) if step == 0 else pos + 1), 1, np.array([t], dtype=np.int32)
        out.append(toks)
    return out


def _worker(rank, world, port, n_decode, split, q, mult=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prompts = [np.arange(P, dtype=np.int32) * (s + 2) % VOCAB for s in range(world * mult)]
    stage = FakeStage(rank, world, world * mult)
    if split:
        st = pipeline.run_ticks(rank, world, stage, dist, torch, prompts, n_decode, "cpu", 0, split)
        dist.barrier()
        st = pipeline.run_ticks(rank, world, stage, dist, torch, prompts, n_decode, "cpu", split, None, st)
    else:
        st = pipeline.run_ticks(rank, world, stage, dist, torch, prompts, n_decode, "cpu")
    q.put((rank, st["produced"], st["received"], stage.log))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_decode,split,mult", [(2, 6, 0, 1), (2, 6, 7, 1), (3, 4, 5, 1),
                                                        (2, 6, 0, 2), (2, 5, 9, 2), (3, 4, 11, 2), (3, 3, 1, 2), (4, 3, 17, 2)])
def test_pipeline_schedule_matches_single_process(world, n_decode, split, mult):
    """mult 1: lock-step schedule, `world` sequences; mult 2: 2*world sequences, stages two ticks apart, the exchange posted before
    and completed after each tick's compute (what bench.py --gpus N runs)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_decode, split, q, mult)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, produced, received, log = q.get(timeout=120)
        res[rank] = (produced, received, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_seq = world * mult
    prompts = [np.arange(P, dtype=np.int32) * (s + 2) % VOCAB for s in range(n_seq)]
    ref = reference(world, prompts, n_decode)
    assert res[world - 1][0] == ref                                   # tokens produced by the last stage
    assert res[0][1] == [r[:-1] for r in ref]                         # tokens rank 0 received back (all but the final one)
    for r in range(world):                                            # every rank ran every item exactly once, in item order
        log = res[r][2]
        assert len(log) == n_seq * (n_decode + 1)
        assert log[:n_seq] == [(s, P, 0) for s in range(n_seq)]
        assert log[n_seq:2 * n_seq] == [(s, 1, P) for s in range(n_seq)]


class FakeBatch:
    """a group of sequences of one FakeStage: the boundary buffers a batched tick exchanges (hidden rows [n, DIM], n token words)"""

    def __init__(self, stage, seqs):
        self.stage, self.seqs = stage, list(seqs)
        self.x = torch.zeros(len(self.seqs), DIM, dtype=torch.int16)
        self.ring = torch.zeros(len(self.seqs), dtype=torch.int32)
        self.pos = {s: P for s in self.seqs}

    def boundary_tensor(self, which):
        return self.x if which == 0 else self.ring


class FakePipe:
    """a pipe without a transport (lnb_pipeline_init_host): tick_batch runs the group's one-token stage step and logs the last stage's tokens"""

    def __init__(self, rank, world):
        self.rank, self.world, self.log, self.ran = rank, world, [], []

    def tick_batch(self, run=None, send=None, recv=None):
        assert send is None and recv is None
        st, slot = run.stage, len(self.log)
        for j, s in enumerate(run.seqs):
            if self.rank > 0:
                st.buf[s][:1] = run.x[j]
            tok = st.run(s, 1, run.pos[s], np.array([int(run.ring[j])], dtype=np.int32) if self.rank == 0 else None)
            run.x[j] = st.buf[s][0]
            run.pos[s] += 1
            if self.rank == self.world - 1:
                run.ring[j] = tok
                self.log.append(tok)
        self.ran.append(tuple(run.seqs))
        return slot if self.rank == self.world - 1 else -1

    def sync(self):
        pass


def _batched_worker(rank, world, port, n_decode, nb, split, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 2 * world
    prompts = [np.arange(P, dtype=np.int32) * (s + 2) % VOCAB for s in range(G * nb)]
    stage = FakeStage(rank, world, G * nb)
    firsts = pipeline.prefill_torch(rank, world, stage, dist, torch, prompts, "cpu")
    bats = [FakeBatch(stage, range(g * nb, (g + 1) * nb)) for g in range(G)]
    if rank == 0:
        for g, b in enumerate(bats):
            b.ring[:] = torch.tensor(firsts[g * nb:(g + 1) * nb], dtype=torch.int32)     # Batch.set_state(tokens, ...)
    pipe = FakePipe(rank, world)
    st = pipeline.run_ticks_batched_torch(rank, world, pipe, bats, dist, torch, n_decode, "cpu", DIM, 0, split or None)
    if split:
        dist.barrier()
        st = pipeline.run_ticks_batched_torch(rank, world, pipe, bats, dist, torch, n_decode, "cpu", DIM, split, None, st)
    toks = None
    if rank == world - 1:
        toks = [[firsts[g * nb + j]] + [pipe.log[sl + j] for sl in st["slots"][g]] for g in range(G) for j in range(nb)]
    q.put((rank, toks, firsts if rank == 0 else None, pipe.ran))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_decode,nb,split", [(2, 5, 3, 0), (2, 4, 2, 5), (3, 3, 2, 7)])
def test_batched_ticks_through_torch_distributed_match_single_process(world, n_decode, nb, split):
    """the torch.distributed fallback's BATCHED tick (pipeline.prefill_torch + run_ticks_batched_torch: groups of nb sequences as the unit,
    2*world groups in flight, hidden rows downstream and the groups' token words back to rank 0), with stand-ins for the library's pipe and
    batches: every sequence's tokens equal the single-process evaluation, every rank ran every group's every step once, in item order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batched_worker, args=(r, world, port, n_decode, nb, split, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, toks, firsts, ran = q.get(timeout=120)
        res[rank] = (toks, firsts, ran)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G = 2 * world
    prompts = [np.arange(P, dtype=np.int32) * (s + 2) % VOCAB for s in range(G * nb)]
    ref = reference(world, prompts, n_decode)
    assert res[world - 1][0] == ref
    assert res[0][1] == [r[0] for r in ref]                           # the prefill's token ring reached rank 0
    for r in range(world):
        assert res[r][2] == [tuple(range(g * nb, (g + 1) * nb)) for _ in range(n_decode) for g in range(G)]


def test_single_rank_pipeline_is_the_plain_greedy_loop():
    prompts = [np.arange(P, dtype=np.int32) * 2 % VOCAB]
    st = pipeline.run_ticks(0, 1, FakeStage(0, 1, 1), dist, torch, prompts, 5, "cpu")
    assert st["produced"] == reference(1, prompts, 5)


def test_stage_layers_cover_every_block_once_and_lighten_the_last_stage():
    import pipeline
    for world, L in [(1, 32), (2, 32), (4, 32), (8, 32), (8, 80), (3, 2), (5, 7), (8, 8), (2, 1)]:
        cuts = [pipeline.stage_layers(r, world, L) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == L
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(lo <= hi for lo, hi in cuts)
    assert [hi - lo for lo, hi in (pipeline.stage_layers(r, 8, 32) for r in range(8))] == [4, 4, 4, 5, 4, 4, 4, 3]
    assert [hi - lo for lo, hi in (pipeline.stage_layers(r, 4, 32) for r in range(4))] == [8, 9, 8, 7]
    assert pipeline.stage_layers(0, 1, 32) == (0, 32)


def test_stage_parts_partition_is_contiguous_and_no_worse_than_whole_blocks():
    import pipeline
    costs = pipeline.part_costs(dict(dim=4096, n_heads=32, n_kv_heads=8, vocab_size=128256), 14336)
    assert abs(sum(costs[:3]) - 140) < 3 and abs(costs[3] - 170) < 5               # the measured 8B block / head times (us)
    for world, L, c in [(1, 32, costs), (2, 32, costs), (4, 32, costs), (8, 32, costs), (8, 80, costs), (3, 2, (0.35, 0.33, 0.32, 1.2)),
                        (5, 7, (0.35, 0.33, 0.32, 1.2)), (8, 8, (1, 1, 1, 1)), (2, 1, (1, 1, 1, 1)), (4, 2, (1, 2, 3, 4)), (6, 2, (1, 1, 1, 9))]:
        cuts = [pipeline.stage_parts(r, world, L, *c) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 3 * L
        for (a, b), (a2, b2) in zip(cuts, cuts[1:]):
            assert b == a2
        assert all(b > a for a, b in cuts)

        def tick(parts):
            return max(sum(c[u % 3] for u in range(a, b)) + (c[3] if b == 3 * L else 0) for a, b in parts)
        if L >= world:
            blocks = [pipeline.stage_layers(r, world, L, c[3] / sum(c[:3])) for r in range(world)]
            assert tick(cuts) <= tick([(3 * a, 3 * b) for a, b in blocks]) + 1e-9
    eight = [pipeline.stage_parts(r, 8, 32, *costs) for r in range(8)]
    assert max(b - a for a, b in eight) <= 13                                       # no stage above 4 1/3 blocks (5 with whole blocks)


def test_stage_parts_reaches_the_min_max_optimum():
    """binary search + greedy fill against an exhaustive dynamic programme on small random instances"""
    import functools
    import random
    import pipeline
    rnd = random.Random(5)
    for _ in range(200):
        L, world = rnd.randint(1, 6), rnd.randint(1, 7)
        c = (rnd.uniform(0.1, 3), rnd.uniform(0.1, 3), rnd.uniform(0.1, 3), rnd.uniform(0, 6))
        n = 3 * L
        if world > n:
            with pytest.raises(ValueError):
                pipeline.stage_parts(0, world, L, *c)
            continue
        cost = [c[u % 3] for u in range(n)]
        cost[-1] += c[3]
        pre = [0.0]
        for x in cost:
            pre.append(pre[-1] + x)

        @functools.lru_cache(None)
        def best(i, k):
            if k == 1:
                return pre[n] - pre[i]
            return min(max(pre[j] - pre[i], best(j, k - 1)) for j in range(i + 1, n - k + 2))
        cuts = [pipeline.stage_parts(r, world, L, *c) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(b > a for a, b in cuts) and all(x[1] == y[0] for x, y in zip(cuts, cuts[1:]))
        got = max(pre[b] - pre[a] for a, b in cuts)
        assert got <= best(0, world) * (1 + 1e-9) + 1e-12, (L, world, c, cuts)


def test_bench_gpus_n_is_one_plain_command():
    """`python bench.py --gpus N` spawns its own ranks (no torchrun): rank 0's stdout is the one JSON line, the other ranks' output goes to
    stderr, every rank sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and a failing rank fails the command.  --dry-run: no GPU is touched."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["env"]["WORLD_SIZE"] == "4" and d["env"]["RANK"] == "0" and d["env"]["MASTER_ADDR"] == "127.0.0.1"
    for k in (1, 2, 3):
        assert "rank %d of 4 up" % k in r.stderr
    # under a launcher (WORLD_SIZE already set) nobody spawns: the process IS a rank
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120,
                        env=dict(env, RANK="2", LOCAL_RANK="2", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r2.returncode == 0 and r2.stdout.strip() == "rank 2 of 4 up"
    # a rank that fails takes the command down with its exit code
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True, text=True, timeout=120,
                        env=dict(env, LNB_DRY_RUN_FAIL_RANK="1"))
    assert r3.returncode == 7 and "rank 1 exited with code 7" in r3.stderr
