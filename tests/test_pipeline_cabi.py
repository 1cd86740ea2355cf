"""The pipeline's exchange behind the C ABI (lnb_pipeline_*, include/lnb.h): what can be checked without a second GPU.

CPU: the library resolves RCCL on first use (dlopen of librccl.so.1 + every entry point the pipeline calls) and reports its errors
through lnb_last_error; argument validation; the native tick schedule (pipeline.run_ticks_native) posts MATCHING sends and receives
on every pair of neighbouring ranks in every tick, never runs an item before its input was received, and closes the token ring.
GPU (-m gpu): a one-stage pipe on one GPU -- stage steps as captured graphs, device-side token ring, pinned token log -- generates
the oracle's tokens for several sequences in flight.  (Ranks > 1 need one GPU each: the driver's multi-GPU run.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lnb():
    import lnb as _lnb
    _lnb.build()
    return _lnb


def _gpu_here():
    import glob
    return os.path.exists("/dev/kfd") and bool(glob.glob("/dev/dri/renderD*"))


def test_rccl_is_resolved_on_first_use_and_errors_surface(lnb):
    """In a child process: on a box without a GPU librccl's own teardown can abort at exit ("double free or corruption") after it was
    asked for an id -- that must not be this test process's exit.  The child reports what it saw before it exits."""
    import json
    import subprocess
    code = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
import lnb
L = lnb.lib()
res = {"null_rc": L.lnb_pipeline_unique_id(None), "null_msg": L.lnb_last_error().decode()}
buf = C.create_string_buffer(128)
res["rc"] = L.lnb_pipeline_unique_id(buf)
res["msg"] = L.lnb_last_error().decode()
res["any"] = any(buf.raw)
res["mapped"] = "librccl" in open("/proc/self/maps").read()
print("RESULT " + json.dumps(res), flush=True)
""" % os.path.join(ROOT, "llama-nuts-and-bolts_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1][7:])
    assert res["null_rc"] != 0 and "null argument" in res["null_msg"]
    if res["rc"] == 0:                          # a GPU box -- or an RCCL build that hands out ids without touching a device
        assert res["any"]
    else:                                       # librccl was loaded and called: it is RCCL that reports the missing device, not the loader
        assert not _gpu_here()
        assert "ncclGetUniqueId failed" in res["msg"], res["msg"]
    assert res["mapped"]                        # the loader mapped the library with every symbol the pipeline needs


def test_pipeline_init_validates_its_arguments(lnb):
    L = lnb.lib()
    out = C.c_void_p()
    assert L.lnb_pipeline_init(None, 0, 1, None, C.byref(out)) != 0 and b"null argument" in L.lnb_last_error()
    assert L.lnb_pipeline_tick(None, None, 0, 0, None, None, 0, None, 0, None) != 0 and b"null argument" in L.lnb_last_error()
    assert L.lnb_pipeline_sync(None) != 0
    assert L.lnb_pipeline_destroy(None) == 0


def _tcp_worker(rank, world, port, q):
    import pipeline
    g = pipeline.TcpGroup(rank, world, "127.0.0.1", port, timeout=30.0)
    got = g.broadcast(b"\x07" * 128 if rank == 0 else None)
    mx = g.all_reduce(float(rank) + 0.5, max)
    mn = g.all_reduce(1 if rank != 1 else 0, min)
    g.barrier()
    costs = g.broadcast([1.0, 2.0, 3.0, 4.0] if rank == 0 else None)
    g.close()
    q.put((rank, got, mx, mn, costs))


def test_tcp_control_plane_broadcast_reduce_barrier():
    """the torch-free control plane of the native bench path: three processes on 127.0.0.1"""
    import multiprocessing as mp
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=60) for _ in range(3))
    for p_ in procs:
        p_.join(30)
    for r, got, mx, mn, costs in res:
        assert got == b"\x07" * 128 and mx == 2.5 and mn == 0 and costs == [1.0, 2.0, 3.0, 4.0]


class _FakePipe:
    def __init__(self, rank):
        self.rank, self.ticks = rank, []

    def tick(self, **kw):
        self.ticks.append(kw)
        return len(self.ticks) - 1


@pytest.mark.parametrize("world,n_decode", [(2, 3), (3, 2), (4, 4), (8, 2)])
def test_native_schedule_posts_matching_sends_and_receives(world, n_decode):
    import pipeline
    P, n_seq = 5, 2 * world
    prompts = [np.arange(P, dtype=np.int32) + s for s in range(n_seq)]
    pipes = [_FakePipe(r) for r in range(world)]
    ctxs = [["ctx%d_%d" % (r, s) for s in range(n_seq)] for r in range(world)]
    for r in range(world):
        pipeline.run_ticks_native(r, world, pipes[r], ctxs[r], prompts, n_decode)
    n_ticks = len(pipes[0].ticks)
    assert all(len(p.ticks) == n_ticks for p in pipes)
    seq_of = lambda name: int(name.split("_")[1])
    got_input = [set() for _ in range(world)]           # (rank) -> {(seq, phase)} inputs received so far
    phase = [[0] * n_seq for _ in range(world)]         # next phase each rank will run per sequence
    ran = [[] for _ in range(world)]
    for t in range(n_ticks):
        for r in range(world):
            kw = pipes[r].ticks[t]
            if kw.get("run") is not None:
                s = seq_of(kw["run"]); k = phase[r][s]
                assert kw["run_rows"] == (P if k == 0 else 1) and kw["run_pos"] == (0 if k == 0 else P + k - 1)
                if r == 0:
                    assert (kw["run_tokens"] is not None) == (k == 0)
                    if k > 0:
                        assert (s, k) in got_input[0], "rank 0 ran a decode step before its token arrived"
                else:
                    assert kw["run_tokens"] is None and (s, k) in got_input[r], "rank %d ran (%d,%d) before its input arrived" % (r, s, k)
                ran[r].append((s, k)); phase[r][s] += 1
        for r in range(world):                           # every send of rank r in tick t is received by its peer in the SAME tick
            kw, peer = pipes[r].ticks[t], pipes[(r + 1) % world].ticks[t]
            if kw.get("send") is not None:
                assert peer.get("recv") is not None and seq_of(peer["recv"]) == seq_of(kw["send"])
                assert r == world - 1 or peer["recv_rows"] == kw["send_rows"]      # (the last rank sends the 4-byte token whatever the row count)
                s = seq_of(kw["send"])
                assert (s, phase[r][s] - 1) in ran[r]                      # it sends something it has computed
                got_input[(r + 1) % world].add((s, phase[r][s] - 1 + (1 if r == world - 1 else 0)))
            else:
                assert peer.get("recv") is None
    for r in range(world):
        assert sorted(ran[r]) == sorted((s, k) for s in range(n_seq) for k in range(1 + n_decode))


@pytest.mark.gpu
def test_rccl_exchange_runs_on_hardware_in_a_one_rank_communicator():
    """The exchange of a tick -- ncclGroupStart; ncclSend; ncclRecv; ncclGroupEnd on a non-blocking stream, device buffer to device
    buffer -- against the real librccl on a one-GPU box: a one-rank communicator, both calls addressed to rank 0.  Run in a fresh
    process that never imports torch (torch brings its own HIP runtime and RCCL), like a pipeline rank."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import lnb; lnb.rccl_selftest(0, 4); lnb.rccl_selftest(0, 8192); "
            "lnb.rccl_selftest(0, (1 << 22) + 24); print('rccl selftest ok')" % os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("long_threshold,mode", [(None, "exact"), (10, "exact"), (0, "exact"), (10, "fast")])
def test_one_stage_pipe_on_one_gpu_generates_the_oracle_tokens(lnb, long_threshold, mode):
    """world == 1: no communicator, but everything else of lnb_pipeline_tick -- prefill with host tokens, one-token steps as replays of
    the captured stage graph with the position on the device, the device-side token ring, the pinned token log.
    long_threshold 10: the sequences cross the attention crossover in mid-run (the stage step has one captured graph per attention
    form); 0: the long-context kernels from the first decode step.  mode fast: the tick path runs the tolerance kernels -- compared
    with the same context driven through lnb_forward (the tolerance mode is deterministic, not oracle-identical)."""
    import pipeline
    from oracle import oracle as orc
    cfg = dict(orc.TINY)
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    gm = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    P, n_seq, n_decode = 6, 3, 9
    prompts = [orc.synth_tokens(50 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
    ctxs = [lnb.InferenceContext(gm, P + n_decode + 2).set_mode(mode) for _ in range(n_seq)]
    if long_threshold is not None:
        for c in ctxs:
            c.set_attention(long_threshold, 0)
    pipe = lnb.Pipeline(gm, 0, 1)
    st = pipeline.run_ticks_native(0, 1, pipe, ctxs, prompts, n_decode, 0, n_seq * 4)      # two windows, like bench.py
    pipe.sync()
    pipeline.run_ticks_native(0, 1, pipe, ctxs, prompts, n_decode, n_seq * 4, None, st)
    pipe.sync()
    for s in range(n_seq):
        got = [int(pipe.read_tokens(q, 1)[0]) for q in st["slots"][s]]
        if mode == "exact":
            ref, _ = orc.Context(om, P + n_decode + 2).generate(prompts[s], n_decode + 1)
            ref = [int(t) for t in ref]
        else:
            fc = lnb.InferenceContext(gm, P + n_decode + 2).set_mode(mode)
            _, tok = fc.Forward(prompts[s], 0, want_logits=False)
            ref = [int(tok)]
            for i in range(n_decode):
                _, tok = fc.Forward([tok], P + i, want_logits=False)
                ref.append(int(tok))
            fc.close()
        assert got == ref, s
    with pytest.raises(lnb.LnbError, match="out of range"):
        pipe.read_tokens(0, 10 ** 6)
    if long_threshold is None:
        # the token log is a ring (here shrunk to 8 slots): slots keep counting, old ones are refused, the newest ones read across the wrap
        os.environ["LNB_PIPELINE_LOG_CAP"] = "8"
        try:
            small = lnb.Pipeline(gm, 0, 1)
        finally:
            del os.environ["LNB_PIPELINE_LOG_CAP"]
        rc = lnb.InferenceContext(gm, P + 24)
        slots = [small.tick(run=rc, run_rows=P, run_pos=0, run_tokens=np.ascontiguousarray(prompts[0], dtype=np.int32))]
        slots += [small.tick(run=rc, run_rows=1, run_pos=P + i) for i in range(19)]
        small.sync()
        assert slots == list(range(20))
        ref20, _ = orc.Context(om, P + 24).generate(prompts[0], 20)
        assert [int(t) for t in small.read_tokens(12, 8)] == [int(t) for t in ref20[12:20]]
        assert int(small.read_tokens(15, 1)[0]) == int(ref20[15])
        with pytest.raises(lnb.LnbError, match="overwritten"):
            small.read_tokens(11, 2)
        small.close(); rc.close()
    s1 = lnb.LlamaTransformer(layer_begin=0, layer_end=1, **cfg).fill_synthetic(1234).finalize()
    with pytest.raises(lnb.LnbError, match="only the last stage owns"):
        lnb.Pipeline(s1, 0, 1)
    with pytest.raises(lnb.LnbError, match="out of range"):
        lnb.Pipeline(gm, 2, 2)
    s1.close(); pipe.close()
    for c in ctxs:
        c.close()
    gm.close(); om.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cuts", [(0, 3, 6), (0, 2, 6), (0, 1, 4, 6), (0, 1, 2, 3, 4, 5, 6)])
def test_multi_stage_pipeline_ticks_through_the_in_process_transport(lnb, cuts):
    """Several pipeline STAGES on one GPU, each with its own lnb_pipe, exchanging through the in-process transport (what RCCL send / recv
    are replaced by is a mailbox + a device copy; the tick code -- events between the compute and exchange streams, captured stage graphs,
    device-side token ring, pinned token log -- is the same).  Stages cut between and inside blocks (after an attention part: the hidden
    state; after a gate/up part: hidden state + activations).  Every rank's schedule is stepped tick by tick, as N processes would, and
    every token of the 2N sequences in flight must be the oracle's."""
    import pipeline
    from oracle import oracle as orc
    cfg = dict(orc.TINY)
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    world = len(cuts) - 1
    P, n_seq, n_decode = 5, 2 * world, 7
    stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **cfg).fill_synthetic(1234).finalize() for a, b in zip(cuts[:-1], cuts[1:])]
    ctxs = [[lnb.InferenceContext(st, P + n_decode + 2) for _ in range(n_seq)] for st in stages]
    pipes = [lnb.Pipeline(stages[r], r, world, loopback_group="t%s" % (cuts,)) for r in range(world)]
    prompts = [orc.synth_tokens(70 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
    n_ticks = (1 + n_decode) * n_seq + 2 * (world - 1)
    state = [None] * world
    for t in range(n_ticks):
        for r in range(world):                                # the host order of one tick across the "ranks" is arbitrary: here 0..N-1
            state[r] = pipeline.run_ticks_native(r, world, pipes[r], ctxs[r], prompts, n_decode, t, t + 1, state[r])
    for p_ in pipes:
        p_.sync()
    for s in range(n_seq):
        got = [int(pipes[-1].read_tokens(q, 1)[0]) for q in state[-1]["slots"][s]]
        ref, _ = orc.Context(om, P + n_decode + 2).generate(prompts[s], n_decode + 1)
        assert got == [int(t) for t in ref], (cuts, s)
    for p_ in pipes:
        p_.close()
    for cs in ctxs:
        for c in cs:
            c.close()
    for st in stages:
        st.close()
    om.close()


@pytest.mark.gpu
def test_bench_force_pipeline_on_one_gpu_runs_the_native_tick_path(lnb):
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--model", "tiny", "--prompt-len", "20"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, LNB_FORCE_PIPELINE="1", MASTER_PORT="29577"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "lnb_pipeline_tick" in d["config"]["exchange"] and d["config"]["host_enqueue_us_per_tick"] > 0


class _FakeBatchPipe:
    def __init__(self, rank):
        self.rank, self.ticks = rank, []

    def tick_batch(self, **kw):
        self.ticks.append(kw)
        return len(self.ticks) - 1


@pytest.mark.parametrize("world,n_decode", [(2, 3), (3, 2), (4, 5), (8, 2)])
def test_batched_schedule_posts_matching_sends_and_receives(world, n_decode):
    """run_ticks_native_batched (groups of sequences as items): every send meets its receive in the same tick, no group runs a step before
    its input arrived (hidden states on ranks > 0; the previous step's tokens on rank 0), every rank runs every (group, step) once"""
    import pipeline
    G = 2 * world
    pipes = [_FakeBatchPipe(r) for r in range(world)]
    batches = [["b%d_%d" % (r, g) for g in range(G)] for r in range(world)]
    for r in range(world):
        pipeline.run_ticks_native_batched(r, world, pipes[r], batches[r], n_decode)
    n_ticks = len(pipes[0].ticks)
    assert all(len(p.ticks) == n_ticks for p in pipes)
    grp = lambda name: int(name.split("_")[1])
    got_input = [set() for _ in range(world)]
    step = [[0] * G for _ in range(world)]
    ran = [[] for _ in range(world)]
    for t in range(n_ticks):
        for r in range(world):
            kw = pipes[r].ticks[t]
            if kw.get("run") is not None:
                g = grp(kw["run"]); k = step[r][g]
                if r > 0 or k > 0:
                    assert (g, k) in got_input[r], "rank %d ran step %d of group %d before its input arrived" % (r, k, g)
                ran[r].append((g, k)); step[r][g] += 1
        for r in range(world):
            kw, peer = pipes[r].ticks[t], pipes[(r + 1) % world].ticks[t]
            if kw.get("send") is not None:
                assert peer.get("recv") is not None and grp(peer["recv"]) == grp(kw["send"])
                g = grp(kw["send"])
                assert (g, step[r][g] - 1) in ran[r]
                got_input[(r + 1) % world].add((g, step[r][g] - 1 + (1 if r == world - 1 else 0)))
            else:
                assert peer.get("recv") is None
    for r in range(world):
        assert sorted(ran[r]) == sorted((g, k) for g in range(G) for k in range(n_decode))


@pytest.mark.gpu
@pytest.mark.parametrize("cuts,n", [((0, 6), 3), ((0, 3, 6), 3), ((0, 3, 6), 16), ((0, 3, 6, 9), 2), ((0, 3, 6), 20)])     # 20: the batches are rows of the streaming product
def test_batched_pipeline_ticks_through_the_in_process_transport(lnb, cuts, n):
    """Pipeline stages of whole blocks on one GPU, BATCHES of n sequences as the unit that moves through them (lnb_pipeline_tick_batch):
    prompts prefilled with single-sequence ticks, then every decode step of a group is one pass over each stage's weights for all its
    sequences.  2 x stages groups in flight (one stage: 2 groups), every rank's schedule stepped tick by tick; every token of every
    sequence must be the oracle's."""
    import pipeline
    from oracle import oracle as orc
    cfg = dict(orc.TINY, n_layers=cuts[-1] // 3)
    om = orc.Model(**cfg).fill_synthetic(1234).finalize()
    world = len(cuts) - 1
    P, n_decode = 6, 7
    G = 2 * world if world > 1 else 2
    stages = [lnb.LlamaTransformer(part_begin=a, part_end=b, **cfg).fill_synthetic(1234).finalize().enable_batch() for a, b in zip(cuts[:-1], cuts[1:])]
    ctxs = [[[lnb.InferenceContext(st, P + n_decode + 2) for _ in range(n)] for _ in range(G)] for st in stages]            # [rank][group][seq]
    pipes = [lnb.Pipeline(stages[r], r, world, loopback_group="b%s%d" % (cuts, n)) if world > 1 else lnb.Pipeline(stages[r], 0, 1) for r in range(world)]
    prompts = [[orc.synth_tokens(900 + 16 * g + s, P, cfg["vocab_size"]) for s in range(n)] for g in range(G)]
    first_slots = {}
    for g in range(G):
        for s in range(n):
            for r in range(world):                            # (lock step: rank order inside each sequence's prefill)
                slot = pipeline.prefill_through_pipeline(r, world, pipes[r], ctxs[r][g][s], prompts[g][s])
            first_slots[(g, s)] = slot
    batches = [[lnb.Batch(ctxs[r][g]).set_state(None, [P] * n) for g in range(G)] for r in range(world)]
    n_ticks = n_decode * G + 2 * (world - 1)
    state = [None] * world
    for t in range(n_ticks):
        for r in range(world):
            state[r] = pipeline.run_ticks_native_batched(r, world, pipes[r], batches[r], n_decode, t, t + 1, state[r])
    for p_ in pipes:
        p_.sync()
    for g in range(G):
        steps = [pipes[-1].read_tokens(q, n) for q in state[-1]["slots"][g]]          # [step][seq]
        for s in range(n):
            ref, _ = orc.Context(om, P + n_decode + 2).generate(prompts[g][s], n_decode + 1)
            got = [int(pipes[-1].read_tokens(first_slots[(g, s)], 1)[0])] + [int(st_[s]) for st_ in steps]
            assert got == [int(t) for t in ref], (cuts, n, g, s)
    for r in range(world):
        for b in batches[r]:
            b.close()
        pipes[r].close()
        for grp_ in ctxs[r]:
            for c in grp_:
                c.close()
        stages[r].close()
    om.close()
