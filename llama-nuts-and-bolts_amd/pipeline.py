"""Layer-sharded pipeline over the GPUs of one node (BASELINE.json configs[3], [4]).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).  Rank r owns a contiguous run of
transformer blocks, cut in thirds of a block and balanced by cost (`stage_parts`: 4 1/3 blocks on seven of 8 GPUs, 1 2/3 + the head
on the last) and their KV caches (`lnb_model_create_parts(..., part_begin, part_end)`); rank 0 also owns
tok_embeddings, rank N-1 also norm + output.  The only exchange of the path is point to point: the bf16 hidden
state [S, dim] from rank r to r+1 (8 KiB per decoded token for dim 4096) and the 4-byte next-token id from rank N-1
back to rank 0.  There is no all-reduce / all-gather anywhere (that would be tensor parallelism).

A single greedy sequence is serial across stages, so independent sequences (one InferenceContext each, exactly
what the reference creates per GenerateString call, src/inference/inference.go:174) are kept in flight:
work item i = (phase i // M, sequence i % M), phase 0 = prefill of the prompt, phase k >= 1 = decode step k-1.
Lock step (M = N sequences): at tick t rank r runs item t - r; item (k, s) reaches rank 0 one tick after rank N-1 finished
(k-1, s), so the token ring closes without bubbles.  Overlapped (M = 2N, what bench.py runs): rank r runs item t - 2r, and the
exchange of a tick is in flight while the stage computes another sequence's item (run_ticks).  Either way every tick is ONE
grouped isend/irecv per rank (batch_isend_irecv: ncclGroupStart/End, so the send to r+1 and the receive from r-1 progress
together and cannot deadlock).

The compute of a tick is delegated to a `stage` object so the schedule is testable on CPU (gloo, world_size 2)
with a pure-python stage: tests/test_pipeline_gloo.py.
"""
import json
import os
import time

import numpy as np


class Stage:
    """Interface of one pipeline stage.  Buffers are torch tensors that live where the backend can send them."""

    def hidden_buffer(self, seq, rows):      # int16 [rows, dim] view of the stage's hidden state for sequence `seq`
        raise NotImplementedError

    def hidden_buffers(self, seq, rows, side):   # everything that crosses the stage boundary: side "in" (from upstream) / "out" (downstream)
        return [self.hidden_buffer(seq, rows)]

    def run(self, seq, rows, start_pos, tokens):   # tokens: np.int32[rows] on the first stage else None
        """consume hidden_buffer(seq) (or tokens), leave the output in hidden_buffer(seq); last stage returns the argmax token"""
        raise NotImplementedError

    def synchronize(self):
        pass

    # non-blocking form used by the overlapped schedule: launch() starts the item, collect() waits for it and returns what run() would
    def launch(self, seq, rows, start_pos, tokens):
        self._result = self.run(seq, rows, start_pos, tokens)

    def collect(self):
        self.synchronize()
        r, self._result = self._result, None
        return r


def schedule(rank, world, n_phases, n_seq=None, gap=1):
    """yield (tick, item_index or None) for this rank; item i = (phase i // n_seq, seq i % n_seq); rank r runs item t - gap*r at tick t"""
    n_seq = n_seq or world
    n_items = n_phases * n_seq
    for t in range(n_items + gap * (world - 1)):
        i = t - gap * rank
        yield t, (i if 0 <= i < n_items else None)


def run_ticks(rank, world, stage, dist, torch, prompts, n_decode, device, lo=0, hi=None, state=None):
    """Run ticks [lo, hi) of the schedule; `state` carries the in-flight items and the token tensors between calls
    (bench.py runs an untimed window and then a timed one).  prompts: n_seq np.int32 arrays of equal length P.
    state["produced"][s] = tokens generated for sequence s (last rank); state["received"][s] on rank 0.

    n_seq == world: the lock-step schedule of the module docstring (exchange, then compute, every tick).
    n_seq == 2*world (world > 1): adjacent stages run TWO ticks apart, so the exchange of tick t -- the result of tick t-1 going
    downstream, the input of tick t+1 arriving -- is posted first and runs on the backend's stream WHILE the stage computes tick t's
    item of another sequence; the tick costs max(compute, exchange) instead of their sum.  Item (k, s) leaves the last rank after
    tick i + 2(N-1) and its token reaches rank 0 during the next one: 2N sequences close the ring without a bubble."""
    P = len(prompts[0])
    n_seq = len(prompts)
    n_phases = 1 + n_decode
    n_items = n_phases * n_seq
    overlap = world > 1 and n_seq == 2 * world
    assert overlap or n_seq == world, "run_ticks wants `world` sequences (lock step) or 2*world (overlapped exchange)"
    gap = 2 if overlap else 1
    first, last = rank == 0, rank == world - 1
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    if state is None:
        state = {}
    # what the backend can move: device tensors for RCCL; host tensors for gloo (the CPU tests, and two ranks sharing one GPU)
    comm = "cpu" if (world > 1 and dist.get_backend() == "gloo") else device
    if "tok_out" not in state:
        state.update(prev=None, tok_out=torch.zeros(1, dtype=torch.int32, device=comm),
                     tok_in=torch.zeros(1, dtype=torch.int32, device=comm), tok_next=None, stage_out={}, stage_in={},
                     produced=[[] for _ in range(n_seq)], received=[[] for _ in range(n_seq)])
    tok_out, tok_in = state["tok_out"], state["tok_in"]
    if hi is None:
        hi = n_items + gap * (world - 1)

    def rows_of(item):
        return P if item // n_seq == 0 else 1

    def staging(kind, q, like):                               # one torch-allocated tensor per (direction, buffer index, row count)
        key = (q, like.shape[0])
        t_ = state[kind].get(key)
        if t_ is None:
            t_ = state[kind][key] = torch.empty_like(like, device=comm)
        return t_

    def post(send_item, recv_item):
        """the exchange of one tick: result of `send_item` downstream (hidden state, or the token back to rank 0), input of `recv_item`
        from upstream.  Only tensors from torch's own allocator are handed to the backend: the stage's hidden state lives in memory
        the HIP library allocated, so it is copied (8 KiB per decoded token) into / out of a torch staging tensor."""
        ops, landed, got_tok = [], [], False
        if send_item is not None and world > 1:
            k, s = divmod(send_item, n_seq)
            if not last:
                for q, src in enumerate(stage.hidden_buffers(s, rows_of(send_item), "out")):
                    out = staging("stage_out", q, src)
                    out.copy_(src)
                    ops.append(dist.P2POp(dist.isend, out, nxt))
            elif k + 1 < n_phases:        # the token of the final phase is not needed by rank 0
                ops.append(dist.P2POp(dist.isend, tok_out, nxt))
        if recv_item is not None and world > 1:
            k, s = divmod(recv_item, n_seq)
            if not first:
                for q, dst in enumerate(stage.hidden_buffers(s, rows_of(recv_item), "in")):
                    inn = staging("stage_in", q, dst)
                    ops.append(dist.P2POp(dist.irecv, inn, prv))
                    landed.append((dst, inn))
            elif k > 0:
                ops.append(dist.P2POp(dist.irecv, tok_in, prv))
                got_tok = True
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, landed, got_tok

    def finish(works, landed, got_tok):
        for req in works:
            req.wait()
        for dst, inn in landed:
            dst.copy_(inn)
        if got_tok:
            state["tok_next"] = int(tok_in.item())
        if works and device != "cpu":
            torch.cuda.synchronize()       # RCCL and the copies ran on torch's streams; the HIP library has its own

    def launch(item):                                         # starts the item on the stage's own stream and returns
        k, s = divmod(item, n_seq)
        if k == 0:
            rows, pos, toks = P, 0, (np.ascontiguousarray(prompts[s], dtype=np.int32) if first else None)
        else:
            rows, pos, toks = 1, P + k - 1, None
            if first:
                tok = state["tok_next"] if world > 1 else state["produced"][s][-1]
                state["received"][s].append(tok)
                toks = np.array([tok], dtype=np.int32)
        stage.launch(s, rows, pos, toks)

    def collect(item):
        out = stage.collect()
        if last:
            state["produced"][item % n_seq].append(int(out))
        return int(out) if last else None

    for t, item in schedule(rank, world, n_phases, n_seq, gap):
        if t < lo:
            continue
        if t >= hi:
            break
        if overlap:
            nxt_item = t + 1 - gap * rank
            nxt_item = nxt_item if 0 <= nxt_item < n_items else None
            if item is not None:
                launch(item)                                  # the stage computes on its own stream ...
            pending = post(state["prev"], nxt_item)           # ... while the exchange runs on the backend's
            tok = collect(item) if item is not None else None
            finish(*pending)
            if tok is not None:
                tok_out.fill_(tok)                            # (after the send of the previous token has completed)
        else:
            finish(*post(state["prev"], item))
            if item is not None:
                launch(item)
                tok = collect(item)
                if tok is not None:
                    tok_out.fill_(tok)
        state["prev"] = item
    return state


class TcpGroup:
    """Minimal control plane for one node: rank 0 listens on (addr, port), the others connect; broadcast / all-reduce / barrier of small
    python values over that star.  It exists so that the NATIVE exchange path never imports torch: PyTorch bundles its own HIP runtime
    and RCCL, and an RCCL bound to one runtime must not be handed streams created by another (liblnb_hip.so uses /opt/rocm's)."""

    def __init__(self, rank, world, addr, port, timeout=300.0):
        import pickle
        import socket
        import struct
        self.rank, self.world, self._pickle, self._struct = rank, world, pickle, struct
        self.peers = []
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port)); srv.listen(world); srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                conns[r] = c
            srv.close()
            self.peers = [conns[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.2)
            c.settimeout(timeout)
            c.sendall(struct.pack("<i", rank))
            self.peers = [c]

    @staticmethod
    def _recvn(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("peer closed the control connection")
            buf += part
        return buf

    def _send(self, c, obj):
        raw = self._pickle.dumps(obj)
        c.sendall(self._struct.pack("<q", len(raw)) + raw)

    def _recv(self, c):
        n = self._struct.unpack("<q", self._recvn(c, 8))[0]
        return self._pickle.loads(self._recvn(c, n))

    def broadcast(self, obj):
        """value of rank 0 on every rank"""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for c in self.peers:
                self._send(c, obj)
            return obj
        return self._recv(self.peers[0])

    def all_reduce(self, value, op):
        """op(values of all ranks) on every rank (op: min / max / sum)"""
        if self.world == 1:
            return value
        if self.rank == 0:
            vals = [value] + [self._recv(c) for c in self.peers]
            return self.broadcast(op(vals))
        self._send(self.peers[0], value)
        return self._recv(self.peers[0])

    def barrier(self):
        self.all_reduce(0, max)

    def set_timeout(self, seconds):
        """patience of every later collective (the rendezvous itself is bounded by the constructor's timeout)"""
        for c in self.peers:
            c.settimeout(seconds)

    def close(self):
        for c in self.peers:
            c.close()
        self.peers = []


def run_ticks_native(rank, world, pipe, ctxs, prompts, n_decode, lo=0, hi=None, state=None):
    """The overlapped schedule of run_ticks driven through the C ABI (lnb_pipeline_tick): rank r runs item t - 2r at tick t, sends the
    result of the item it ran in the previous tick and receives the input of the item of the next tick -- all ENQUEUED; nothing is
    synchronised here (the caller calls pipe.sync() where it needs the device to have caught up).  2*world sequences in flight
    (world == 1: the token ring is a device copy inside the library, any number of sequences).
    state["slots"][s] on the last rank = token-log slots of sequence s's tokens, in order (pipe.read_tokens)."""
    P, n_seq = len(prompts[0]), len(prompts)
    n_phases = 1 + n_decode
    n_items = n_phases * n_seq
    gap = 2 if world > 1 else 1
    assert world == 1 or n_seq == 2 * world, "the overlapped schedule keeps 2*world sequences in flight"
    first, last = rank == 0, rank == world - 1
    if state is None:
        state = {"prev": None, "slots": [[] for _ in range(n_seq)]}
    if hi is None:
        hi = n_items + gap * (world - 1)

    def rows_of(item):
        return P if item // n_seq == 0 else 1

    for t, item in schedule(rank, world, n_phases, n_seq, gap):
        if t < lo:
            continue
        if t >= hi:
            break
        kw = {}
        if item is not None:
            k, s = divmod(item, n_seq)
            kw.update(run=ctxs[s], run_rows=rows_of(item), run_pos=0 if k == 0 else P + k - 1,
                      run_tokens=(np.ascontiguousarray(prompts[s], dtype=np.int32) if (first and k == 0) else None))
        prev = state["prev"]
        if prev is not None and world > 1:
            k, s = divmod(prev, n_seq)
            if not last or k + 1 < n_phases:                  # the token of the final phase is not needed by rank 0
                kw.update(send=ctxs[s], send_rows=rows_of(prev))
        nxt_item = t + 1 - gap * rank
        if 0 <= nxt_item < n_items and world > 1:
            k, s = divmod(nxt_item, n_seq)
            if not first or k > 0:
                kw.update(recv=ctxs[s], recv_rows=rows_of(nxt_item))
        slot = pipe.tick(**kw)
        if item is not None and last:
            state["slots"][item % n_seq].append(slot)
        state["prev"] = item
    return state


# ---------------------------------------------------------------------------------------------------------------------
def stage_layers(rank, world, n_layers, head_cost=1.2):
    """[layer_begin, layer_end) of pipeline stage `rank`.  The last stage also runs the final norm + LM head, which costs about
    1.2 transformer blocks of HBM time for the 8B shape (1.05 GB against 0.44 GB per block, but at the fat kernels' bandwidth), so
    the blocks are split by cost rather than evenly: 8 stages of 32 blocks -> 4,4,4,5,4,4,4,3 instead of 8 x 4 (the tick time is
    the slowest stage: 5 blocks instead of 4 blocks + head)."""
    total = n_layers + head_cost
    cuts = [0] * (world + 1)
    cuts[world] = n_layers
    for r in range(1, world):
        c = int(round(r * total / world))
        if n_layers >= world:                             # every stage keeps at least one block
            c = max(cuts[r - 1] + 1, min(c, n_layers - (world - r)))
        cuts[r] = max(cuts[r - 1], min(c, n_layers))
    return cuts[rank], cuts[rank + 1]


def part_costs(cfg, ffn_hidden):
    """(attention part, gate/up part, down part, head) in microseconds of one decode step: weight megabytes x the per-MB time of the
    kernel class that streams them, measured on the 8B shape (thin fused-norm and row-broadcast GEMVs ~0.48, w2 0.38,
    the fat w1|w3 0.20, the output projection 0.16 us/MB; ~8 us for the attention kernel itself)."""
    dim, hd = cfg["dim"], cfg["dim"] // cfg["n_heads"]
    kv = (cfg["n_kv_heads"] if cfg.get("n_kv_heads", -1) > 0 else cfg["n_heads"]) * hd
    mb = 2.0 / 1e6
    attn = 0.48 * (dim * (dim + 2 * kv) + dim * dim) * mb + 8.0
    w13 = 0.20 * (2 * ffn_hidden * dim) * mb
    w2 = 0.38 * (ffn_hidden * dim) * mb
    head = 0.16 * (cfg["vocab_size"] * dim) * mb
    return attn, w13, w2, head


def probe_costs(lnb, cfg, device_index, pos, iters=24):
    """(attention part, gate/up part, down part, head) measured on THIS GPU for THIS shape: a four-block model with the head, HIP-event
    time of each decode kernel class (lnb_profile_kernel).  The table of part_costs is calibrated on the 8B shape; for other shapes
    the thin kernels run at different fractions of the bandwidth (dim 8192: the attention part takes 93 us where the table says
    153), and the partition should follow the machine, not the table."""
    one = dict(cfg, n_layers=int(os.environ.get("LNB_PROBE_LAYERS", "4")))   # (the timing loop cycles through the blocks)
    m = lnb.LlamaTransformer(device=device_index, **one).fill_synthetic(7).finalize(rope_rows=max(pos + 8, 2 * cfg.get("max_seq_len", 2048)))
    c = lnb.InferenceContext(m, pos + 8)
    _, tok = c.Forward(lnb.synth_tokens(3, 4, cfg["vocab_size"]), 0, want_logits=False)   # real activations in the buffers (an all-zero
    c.decode_greedy(tok, 4, 2)                                                            # row sends the exact norm sum down its replay path)
    for which in range(6):
        c.profile_kernel(which, pos, 2)                       # first launches load code objects: not part of the measurement
    t = [c.profile_kernel(which, pos, iters) * 1e3 for which in range(6)]          # qkv, attention, wo, w1|w3, w2, head (us)
    c.close(); m.close()
    return t[0] + t[1] + t[2], t[3], t[4], t[5]


def stage_parts(rank, world, n_layers, attn_cost=0.35, w13_cost=0.33, w2_cost=0.32, head_cost=1.2):
    """[part_begin, part_end) of pipeline stage `rank` in thirds of a block (3l = attention part of block l, 3l+1 = gate/up part,
    3l+2 = down part; lnb_model_create_parts).  Contiguous partition of the 3*n_layers parts into `world` non-empty stages that
    minimises the slowest stage (= the tick of the pipeline), the last stage also carrying the head: binary search on the bound +
    greedy fill (optimal for min-max).  8B shape on 8 GPUs: the slowest stage is 608 us (whole blocks: 699, ideal 580)."""
    n = 3 * n_layers
    if world > n:
        raise ValueError("more pipeline stages (%d) than block parts (%d)" % (world, n))
    cost = [(attn_cost, w13_cost, w2_cost)[u % 3] for u in range(n)]
    cost[-1] += head_cost

    def cuts_for(bound):
        cuts, acc = [0], 0.0
        for u in range(n):
            if cost[u] > bound:
                return None
            if acc > 0 and acc + cost[u] > bound:
                cuts.append(u); acc = 0.0
            acc += cost[u]
        cuts.append(n)
        return cuts if len(cuts) - 1 <= world else None

    lo, hi = max(cost), sum(cost)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if cuts_for(mid) is None:
            lo = mid
        else:
            hi = mid
    cuts = cuts_for(hi)
    while len(cuts) - 1 < world:                              # fewer stages than ranks: split the longest splittable stage
        j = max(range(len(cuts) - 1), key=lambda q: cuts[q + 1] - cuts[q])
        cuts.insert(j + 1, (cuts[j] + cuts[j + 1] + 1) // 2)
    return cuts[rank], cuts[rank + 1]


class LnbStage(Stage):
    """One GPU's share of the model behind the C ABI (lnb_forward_stage)."""

    def __init__(self, lnb, torch, cfg, rank, world, n_seq, seq_len, device_index, parts=None, costs=None):
        import ctypes as C
        self.lnb, self.torch, self.C = lnb, torch, C      # torch may be None (native exchange path): only the zero-copy views need it
        L = cfg["n_layers"]
        self.first, self.last = rank == 0, rank == world - 1
        probe = lnb.ModelArgs(**dict(lnb.LLAMA_8B, **cfg))
        self.ffn_hidden = lnb.lib().lnb_model_ffn_hidden_dim(C.byref(probe))
        self.costs = tuple(costs) if costs is not None else part_costs(cfg, self.ffn_hidden)      # (costs: measured by probe_costs)
        pb, pe = parts if parts is not None else stage_parts(rank, world, L, *self.costs)     # (parts: a test's own cut)
        self.parts = (pb, pe)
        self.model = lnb.LlamaTransformer(device=device_index, part_begin=pb, part_end=pe, **cfg).fill_synthetic(1234)
        self.model.finalize(rope_rows=max(seq_len, 2 * cfg["max_seq_len"]))
        self.ctx = [lnb.InferenceContext(self.model, seq_len) for _ in range(n_seq)]
        self.dim, self.device_index = cfg["dim"], device_index
        self._views = {}

    def _view(self, seq, rows, which):
        key = (seq, rows, which)
        if key not in self._views:
            ptr = self.lnb.lib().lnb_ctx_hidden_ptr(self.ctx[seq].h, which)
            width = self.ffn_hidden if which == 2 else self.dim

            class _Wrap:   # zero-copy view of the library's device buffer (CUDA array interface v2)
                __cuda_array_interface__ = {"shape": (rows, width), "typestr": "<i2", "data": (int(ptr), False), "version": 2}
            self._views[key] = self.torch.as_tensor(_Wrap(), device="cuda:%d" % self.device_index)
        return self._views[key]

    def hidden_buffer(self, seq, rows):
        return self._view(seq, rows, 0)

    def hidden_buffers(self, seq, rows, side):
        """the [rows, dim] hidden state, plus the [rows, ffn_hidden] gate*up activations when that boundary of the stage lies between a
        block's gate/up part and its down part (part index 3l+2)"""
        edge = self.parts[0] if side == "in" else self.parts[1]
        return [self._view(seq, rows, 0)] + ([self._view(seq, rows, 2)] if edge % 3 == 2 else [])

    def run(self, seq, rows, start_pos, tokens):
        C, L = self.C, self.lnb.lib()
        am = C.c_int32(-2)
        tok_p = tokens.ctypes.data_as(C.c_void_p) if tokens is not None else None
        self.lnb._chk(L.lnb_forward_stage(self.ctx[seq].h, tok_p, rows, start_pos, None, C.byref(am) if self.last else None))
        return am.value if self.last else None

    def launch(self, seq, rows, start_pos, tokens):           # lnb_forward_stage_begin: enqueue and return
        tok_p = tokens.ctypes.data_as(self.C.c_void_p) if tokens is not None else None
        self.lnb._chk(self.lnb.lib().lnb_forward_stage_begin(self.ctx[seq].h, tok_p, rows, start_pos, 1 if self.last else 0))
        self._inflight = seq

    def collect(self):                                        # lnb_forward_stage_end: wait, return the last stage's token
        am = self.C.c_int32(-2)
        self.lnb._chk(self.lnb.lib().lnb_forward_stage_end(self.ctx[self._inflight].h, self.C.byref(am) if self.last else None))
        return am.value if self.last else None

    def close(self):
        for c in self.ctx:
            c.close()
        self.model.close()


def run_single_stream_native(rank, world, pipe, ctx, prompt, n_decode, lo=0, hi=None, state=None):
    """ONE greedy sequence through the pipeline (configs[3]'s "single-stream" figure): inherently serial across the stages -- token t + 1
    needs token t from the last rank -- so every rank simply enqueues, per step, the receive of its input, its stage step and the send
    of its result; the events inside lnb_pipeline_tick order them on the device and nothing is synchronised here.  Steps [lo, hi) of
    1 + n_decode (step 0 = the prompt).  state["slots"] on the last rank = token-log slots in order."""
    P = len(prompt)
    first, last = rank == 0, rank == world - 1
    if state is None:
        state = {"slots": []}
    if hi is None:
        hi = 1 + n_decode
    for k in range(lo, hi):
        rows, pos = (P, 0) if k == 0 else (1, P + k - 1)
        if world > 1 and (not first or k > 0):
            pipe.tick(recv=ctx, recv_rows=rows)               # hidden state from rank - 1; on rank 0 the token of step k - 1 from the last rank
        slot = pipe.tick(run=ctx, run_rows=rows, run_pos=pos, run_tokens=(np.ascontiguousarray(prompt, dtype=np.int32) if (first and k == 0) else None))
        if last:
            state["slots"].append(slot)
        if world > 1 and (not last or k + 1 < 1 + n_decode):  # (the token of the final step is not needed by rank 0)
            pipe.tick(send=ctx, send_rows=rows)
    return state


def prefill_through_pipeline(rank, world, pipe, ctx, prompt):
    """one sequence's prompt through all stages with the single-sequence ticks; afterwards rank 0's context holds the first generated token
    in its device token word (received from the last rank) -- the state lnb_batch_set_state(tokens = NULL) picks up"""
    P = len(prompt)
    first, last = rank == 0, rank == world - 1
    if world > 1 and not first:
        pipe.tick(recv=ctx, recv_rows=P)
    slot = pipe.tick(run=ctx, run_rows=P, run_pos=0, run_tokens=(np.ascontiguousarray(prompt, dtype=np.int32) if first else None))
    if world > 1:
        pipe.tick(send=ctx, send_rows=P)                      # (the last rank: its token word to rank 0)
        if first:
            pipe.tick(recv=ctx, recv_rows=1)
    return slot


def run_ticks_native_batched(rank, world, pipe, batches, n_decode, lo=0, hi=None, state=None):
    """The overlapped schedule with BATCHES as items (lnb_pipeline_tick_batch): group g = batches[g] (every rank's batch over its stage's
    contexts of the same sequences); item i = (decode step i // G, group i % G); rank r runs item t - gap*r at tick t, sends the result of
    the item it ran in the previous tick and receives the input of the item of the next tick.  2*world groups in flight (world == 1: any
    number; the token ring stays inside the batch).  state["slots"][g] on the last rank = first token-log slot of each of the group's steps."""
    G = len(batches)
    n_items = n_decode * G
    gap = 2 if world > 1 else 1
    assert world == 1 or G == 2 * world, "the overlapped schedule keeps 2*world groups in flight"
    first, last = rank == 0, rank == world - 1
    if state is None:
        state = {"prev": None, "slots": [[] for _ in range(G)]}
    if hi is None:
        hi = n_items + gap * (world - 1)
    for t, item in schedule(rank, world, n_decode, G, gap):
        if t < lo:
            continue
        if t >= hi:
            break
        kw = {}
        if item is not None:
            kw["run"] = batches[item % G]
        prev = state["prev"]
        if prev is not None and world > 1:
            k, g = divmod(prev, G)
            if not last or k + 1 < n_decode:                  # the tokens of the final step are not needed by rank 0
                kw["send"] = batches[g]
        nxt = t + 1 - gap * rank
        if 0 <= nxt < n_items and world > 1:
            k, g = divmod(nxt, G)
            if not first or k > 0:                            # (step 0's tokens are already in the contexts: the prefill's ring)
                kw["recv"] = batches[g]
        slot = pipe.tick_batch(**kw)
        if item is not None and last:
            state["slots"][item % G].append(slot)
        state["prev"] = item
    return state


def device_view(torch, ptr, shape, typestr, device_index):
    """zero-copy torch view of a device buffer the library owns (CUDA array interface v2)"""
    class _Wrap:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Wrap(), device="cuda:%d" % device_index)


def prefill_torch(rank, world, stage, dist, torch, prompts, device):
    """every sequence's prompt through all stages, one after the other, with torch.distributed moving the hidden state (the fallback's
    counterpart of prefill_through_pipeline).  -> on rank 0 (and the last rank): the first generated token of every sequence."""
    P = len(prompts[0])
    first, last = rank == 0, rank == world - 1
    comm = "cpu" if (world > 1 and dist.get_backend() == "gloo") else device
    firsts = []
    for q, prompt in enumerate(prompts):
        if not first:
            for dst in stage.hidden_buffers(q, P, "in"):
                inn = torch.empty_like(dst, device=comm)
                dist.recv(inn, rank - 1)
                dst.copy_(inn)
            if device != "cpu":
                torch.cuda.synchronize()
        stage.launch(q, P, 0, np.ascontiguousarray(prompt, dtype=np.int32) if first else None)
        tok = stage.collect()
        if not last:
            for src in stage.hidden_buffers(q, P, "out"):
                out = torch.empty_like(src, device=comm)
                out.copy_(src)
                dist.send(out, rank + 1)
        else:
            firsts.append(int(tok))
    if world > 1 and (first or last):                         # the token ring of the prefill, all sequences at once
        t_ = torch.tensor(firsts if last else [0] * len(prompts), dtype=torch.int32, device=comm)
        if last:
            dist.send(t_, 0)
        else:
            dist.recv(t_, world - 1)
            firsts = [int(v) for v in t_.tolist()]
    return firsts


def run_ticks_batched_torch(rank, world, pipe, batches, dist, torch, n_decode, device, dim, lo=0, hi=None, state=None):
    """run_ticks_native_batched with the exchange done HERE (the fallback when RCCL cannot form its communicator, and what two ranks sharing
    one GPU over gloo run): `pipe` has no transport (lnb.Pipeline(..., host_transport=True)), a tick enqueues the group's stage step
    (tick_batch(run=...): captured graph, positions advance on the device), and while it runs the result of the previous tick's group goes
    downstream and the input of the next tick's group arrives through torch staging tensors (batch_isend_irecv: one grouped exchange per
    tick); then the stage step is waited for (pipe.sync) and the received rows are copied into the group's boundary buffers
    (lnb_batch_boundary_ptr: hidden states [n, dim], or the n token words on the ring's edge).  Same schedule, same item order, same slots."""
    G = len(batches)
    n_items = n_decode * G
    gap = 2 if world > 1 else 1
    assert world == 1 or G == 2 * world, "the overlapped schedule keeps 2*world groups in flight"
    first, last = rank == 0, rank == world - 1
    nxt_rank, prv_rank = (rank + 1) % world, (rank - 1) % world
    comm = "cpu" if (world > 1 and dist.get_backend() == "gloo") else device
    dev_index = int(str(device).split(":")[1]) if ":" in str(device) else 0
    if state is None:
        state = {"prev": None, "slots": [[] for _ in range(G)], "views": {}, "staging": {}}
    if hi is None:
        hi = n_items + gap * (world - 1)

    def view(g, which):                                       # 0: hidden states [n, dim] (bf16 bits), 1: the n token words
        key = (g, which)
        if key not in state["views"] and hasattr(batches[g], "boundary_tensor"):      # (a test's stand-in batch on the CPU)
            state["views"][key] = batches[g].boundary_tensor(which)
        if key not in state["views"]:
            n = len(batches[g].ctxs)
            state["views"][key] = device_view(torch, batches[g].boundary_ptr(which), (n, dim) if which == 0 else (n,), "<i2" if which == 0 else "<i4", dev_index)
        return state["views"][key]

    def staging(kind, g, which, like):
        key = (kind, g, which)
        if key not in state["staging"]:
            state["staging"][key] = torch.empty_like(like, device=comm)
        return state["staging"][key]

    for t, item in schedule(rank, world, n_decode, G, gap):
        if t < lo:
            continue
        if t >= hi:
            break
        slot = pipe.tick_batch(run=batches[item % G]) if item is not None else None     # enqueued on the batch's own stream ...
        ops, landed = [], []
        prev = state["prev"]
        if prev is not None and world > 1:                    # ... while the exchange runs on the backend's (the previous tick ended with pipe.sync: prev's result is complete)
            k, g = divmod(prev, G)
            if not last or k + 1 < n_decode:                  # the tokens of the final step are not needed by rank 0
                src = view(g, 1 if last else 0)
                out = staging("out", g, 1 if last else 0, src)
                out.copy_(src)
                ops.append(dist.P2POp(dist.isend, out, nxt_rank))
        nx = t + 1 - gap * rank
        if 0 <= nx < n_items and world > 1:
            k, g = divmod(nx, G)
            if not first or k > 0:                            # (step 0's tokens: Batch.set_state put them into the ring's words)
                dst = view(g, 1 if first else 0)
                inn = staging("in", g, 1 if first else 0, dst)
                ops.append(dist.P2POp(dist.irecv, inn, prv_rank))
                landed.append((dst, inn))
        works = dist.batch_isend_irecv(ops) if ops else []
        pipe.sync()
        for req in works:
            req.wait()
        for dst, inn in landed:
            dst.copy_(inn)
        if works and device != "cpu":
            torch.cuda.synchronize()                          # the backend and the copies ran on torch's streams; the library has its own
        if item is not None and last:
            state["slots"][item % G].append(slot)
        state["prev"] = item
    return state


def blocks_split(rank, world, n_layers):
    """configs[3] literally: n_layers / world whole blocks per GPU (the head on top of the last stage's share)"""
    return 3 * (rank * n_layers // world), 3 * ((rank + 1) * n_layers // world)


def _golden_check(tokens, prompt_len, model_name):
    """sequence 0 of the pipeline run has the headline's prompt (synth_tokens(99, P)): its tokens against the CPU oracle's golden
    continuation of configs[1] (tests/golden/configs1_tokens.json), as bench.py's one-GPU line does"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "configs1_tokens.json")
    if model_name != "Llama-3.1-8B" or prompt_len != 128 or not os.path.exists(path):
        return None
    gold = json.load(open(path))["tokens"]
    n = min(len(tokens), len(gold))
    agree = next((i for i in range(n) if int(tokens[i]) != gold[i]), n)
    return {"compared": n, "identical_prefix": agree, "golden": "tests/golden/configs1_tokens.json (CPU oracle)"}


def _multi_golden_check(seqs, prompt_len, model_name):
    """the sequences in flight that the (sparse) multi-prompt golden holds (sequence s has the prompt synth_tokens(99 + s, P)) against the CPU oracle's continuation of their own prompts on the full model:
    tests/golden/configs1_multi_P<P>_tokens.json (tests/golden/make_multi_prompt_tokens.py), as bench.py's one-GPU sections do"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "configs1_multi_P%d_tokens.json" % prompt_len)
    if model_name != "Llama-3.1-8B" or not seqs or not os.path.exists(path):
        return None
    g = json.load(open(path))
    ids = [k for k in g["sequences"] if k < len(seqs)]
    same, per, first_bad = 0, 0, None
    for q in ids:
        got, gold = [int(t) for t in seqs[q]], g["tokens"][str(q)]
        per = min(len(got), len(gold))
        agree = next((i for i in range(per) if got[i] != gold[i]), per)
        same += int(agree == per)
        if agree < per and first_bad is None:
            first_bad = {"sequence": q, "token": agree, "got": got[agree], "oracle": gold[agree]}
    return {"sequences_compared": ids, "sequences_identical": same, "tokens_each": per, "first_mismatch": first_bad,
            "golden": "tests/golden/configs1_multi_P%d_tokens.json (CPU oracle: each of these sequences against the continuation of ITS prompt)" % prompt_len}


def one_gpu_anchor(lnb, grp, rank, world, cfg, local, mode, P, W, K, n_seq, seq_len, batched, sched="throughput"):
    """The same workloads on ONE GPU holding the whole model, measured in this run on rank 0's GPU while the other ranks wait: the anchor
    of the line's efficiency figures (value_N / (N x anchor)).  Unbatched: n_seq sequences in flight through the one-GPU form of the tick
    path.  Batched (if the pipeline ran it): G groups of nb sequences.  Short timed windows (the anchor needs no more than a few percent)."""
    out = {}
    if rank == 0:
        try:
            Ka, Wa = min(K, 24), min(W, 2)
            m = lnb.LlamaTransformer(device=local, **cfg).fill_synthetic(1234).finalize(rope_rows=max(seq_len, 2 * cfg["max_seq_len"]))
            ctxs = [lnb.InferenceContext(m, seq_len).set_mode(mode).set_schedule(sched) for _ in range(n_seq)]
            pp = lnb.Pipeline(m, 0, 1, None)
            prm = [lnb.synth_tokens(99 + q, P, cfg["vocab_size"]) for q in range(n_seq)]
            n_dec = Wa + Ka
            st = run_ticks_native(0, 1, pp, ctxs, prm, n_dec, 0, n_seq * (1 + Wa))
            pp.sync()
            t0 = time.perf_counter()
            run_ticks_native(0, 1, pp, ctxs, prm, n_dec, n_seq * (1 + Wa), n_seq * (1 + Wa + Ka), st)
            pp.sync()
            out["unbatched_tokens_per_s"] = round(Ka * n_seq / (time.perf_counter() - t0), 2)
            out["unbatched_sequences_in_flight"] = n_seq
            pp.close()
            for c in ctxs:
                c.close()
            if batched and mode == "exact":
                G, nb = batched
                try:
                    m.enable_batch()
                except lnb.LnbError:                         # no room for the matrix-core copy: the batches run as rows on the resident layouts
                    out["batched_without_the_second_copy"] = True
                cb = [lnb.InferenceContext(m, seq_len) for _ in range(G * nb)]
                firsts = [c.Forward(lnb.synth_tokens(99 + q, P, cfg["vocab_size"]), 0, want_logits=False)[1] for q, c in enumerate(cb)]
                bats = [lnb.Batch(cb[g * nb:(g + 1) * nb]) for g in range(G)]
                toks = []
                for g, b in enumerate(bats):
                    w_, _ = b.decode(firsts[g * nb:(g + 1) * nb], [P] * nb, Wa) if Wa > 0 else (None, 0.0)
                    toks.append([int(w_[q][-1]) for q in range(nb)] if Wa > 0 else firsts[g * nb:(g + 1) * nb])
                lnb._chk(lnb.lib().lnb_ctx_synchronize(cb[0].h))
                t0 = time.perf_counter()
                for g, b in enumerate(bats):                 # (one GPU: the groups one after another -- a batch step already fills the chip)
                    b.decode(toks[g], [P + Wa] * nb, Ka)
                out["batched_tokens_per_s"] = round(Ka * G * nb / (time.perf_counter() - t0), 2)
                out["batched_sequences_in_flight"] = G * nb
                for b in bats:
                    b.close()
                for c in cb:
                    c.close()
            m.close()
            out["note"] = "whole model on rank 0's GPU alone, %d timed steps; the other ranks wait at a barrier" % Ka
        except lnb.LnbError as e:
            out = {"skipped": str(e)}
    grp.barrier()
    return grp.broadcast(out if rank == 0 else None)


def preflight(lnb, grp, rank, world, local):
    """First contact of a multi-GPU run, BEFORE anything is built or timed: every rank reports the device it sits on, what that device can
    reach peer-to-peer, and whether RCCL works on it at all (lnb_pipeline_selftest: a one-rank communicator running exactly a tick's grouped
    ncclSend + ncclRecv).  The records of all ranks come back to every rank; an error on ANY rank ends ALL of them with one message."""
    rec = {"rank": rank, "device": local, "pid": os.getpid(), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    try:
        n_dev = lnb.device_count()
        rec["visible_devices"] = n_dev
        rec.update(lnb.device_info(local))
        rec["peer_access"] = [lnb.can_access_peer(local, j) for j in range(n_dev)]
        rec["pci"] = lnb.pci_bus_id(local)
        t0 = time.perf_counter()
        lnb.rccl_selftest(local, 1 << 16)
        rec["rccl_selftest"] = "ok (%.1f s incl. loading librccl)" % (time.perf_counter() - t0)
    except Exception as e:
        rec["error"] = "%s: %s" % (type(e).__name__, e)
    recs = grp.all_reduce([rec], lambda vs: sorted(sum(vs, []), key=lambda r: r["rank"]))
    # the GPU a rank sits on is its PCI bus id, not its index: a launcher that hands every rank ONE visible device makes them all "device 0"
    devs = [r.get("pci") or r["device"] for r in recs]
    bad = ["rank %d: %s" % (r["rank"], r["error"]) for r in recs if "error" in r]
    if world > 1 and len(set(devs)) != len(devs) and os.environ.get("LNB_PIPELINE_BACKEND", "nccl") == "nccl":
        bad.append("ranks share a GPU (devices %s): RCCL wants one GPU per rank" % devs)
    return recs, bad


def abort_all(rank, what, details):
    """one clear message (rank 0 prints it), every rank exits non-zero: bench.py's launcher then stops whatever is left"""
    import sys
    if rank == 0:
        sys.stderr.write("bench.py --gpus N ABORTED %s:\n  %s\n" % (what, "\n  ".join(details)))
        sys.stderr.flush()
    os._exit(4)


def bench_main(args, cfg, name):
    """bench.py --gpus N under torchrun: weak scaling, 2N sequences in flight, one rank per GPU.

    Data plane: RCCL point-to-point INSIDE the library (lnb_pipeline_tick: ncclSend / ncclRecv straight from / into the stage's device
    buffers, stage steps as captured graphs, no per-tick synchronisation).  Control plane (the 128-byte RCCL id, barriers, max over
    ranks, the measured part costs): a small TCP star (TcpGroup) -- the native path never imports torch, so the process holds exactly one
    HIP runtime and one RCCL.  LNB_PIPELINE_EXCHANGE=torch (or a failed native init on any rank) runs the exchange through
    torch.distributed instead (run_ticks: staging tensors + batch_isend_irecv), which is also what the gloo tests use."""
    import sys
    import lnb
    # librccl prints a version banner to the C stdout (flushed at exit, i.e. AFTER anything python printed): keep the process's
    # stdout for the one JSON line and send everything else written to fd 1 to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:                                   # (bench.py spawns the ranks itself when WORLD_SIZE is not set: this is a mismatched manual launch)
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch `python bench.py --gpus N` plainly, or under torch.distributed.run with --nproc-per-node N" % (world, args.gpus))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")           # (only used by the single-process LNB_FORCE_PIPELINE=1 run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("LNB_PIPELINE_BACKEND", "nccl")      # "gloo": ranks may share a GPU (tests); the exchange is staged on the host
    exchange = os.environ.get("LNB_PIPELINE_EXCHANGE", "native" if backend == "nccl" else "torch")
    P, W, K = args.prompt_len, args.warmup, args.steps
    seq_len = P + W + K + 8
    mode = getattr(args, "mode", "exact")
    # sequences in flight on every rank: the throughput forms of the one-token kernels (lnb_ctx_set_schedule); LNB_PIPELINE_SCHED=latency for the A/B
    sched = os.environ.get("LNB_PIPELINE_SCHED", "throughput")
    preflight_rec = None
    stage, dist, torch = None, None, None
    if exchange == "native":
        n_dev = lnb.device_count()
        local %= max(1, n_dev)
        # the torchrun agent's own store listens on MASTER_PORT: the control star takes a port next to it
        # first contact is bounded: a rank that never shows up, a device without peer access or an RCCL that cannot start ends ALL ranks with one
        # message inside LNB_PREFLIGHT_TIMEOUT seconds (default 120: eight ranks initialising HIP on a cold node at once have never been timed here) instead of a hang somewhere inside the first exchange
        t_first = float(os.environ.get("LNB_PREFLIGHT_TIMEOUT", "120"))
        try:
            grp = TcpGroup(rank, world, os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + int(os.environ.get("LNB_CONTROL_PORT_OFFSET", "23")), timeout=t_first)
        except Exception as e:
            sys.stderr.write("bench.py --gpus %d ABORTED at the rendezvous: rank %d saw %s: %s after %.0f s (are all %d ranks running?)\n" % (world, rank, type(e).__name__, e, t_first, world))
            os._exit(4)
        pre, bad = preflight(lnb, grp, rank, world, local) if world > 1 or os.environ.get("LNB_FORCE_PREFLIGHT") == "1" else ([], [])
        if rank == 0 and pre:
            sys.stderr.write("[preflight] %s\n" % json.dumps(pre))
        if bad:
            abort_all(rank, "in the preflight (before anything was built or timed)", bad)
        # every rank is there and healthy: from here the control star only has to notice a rank that DIES (its socket closes at once); the
        # patience covers rank 0's probe and the slowest rank's stage build on a cold box.  Forming the communicator stays bounded by the watchdog below.
        grp.set_timeout(max(t_first, 180.0, float(os.environ.get("LNB_COMM_INIT_TIMEOUT", "180")) + 60.0))
        n_seq = 2 * world if world > 1 else int(os.environ.get("LNB_PIPELINE_SEQS", "2"))
        costs = None
        if world > 1 and os.environ.get("LNB_PIPELINE_PROBE", "1") != "0":
            # rank 0 times the three block parts and the head on its GPU; every rank cuts the model with the same numbers
            costs = grp.broadcast([float(v) for v in probe_costs(lnb, cfg, local, P + W + K // 2)] if rank == 0 else None)
        stage = LnbStage(lnb, None, cfg, rank, world, n_seq, seq_len, local, costs=costs)
        for c in stage.ctx:
            c.set_mode(mode).set_schedule(sched)             # several sequences in flight per rank: the co-residency-friendly kernel forms
        ok, pipe, why = 1, None, None
        import threading
        # ncclCommInitRank blocks until all N ranks have joined: a watchdog ends this rank (and, through the launcher, the others) if it does not return
        # (its own, longer patience -- LNB_COMM_INIT_TIMEOUT, default 180 s: the first communicator of a cold 8-GPU node detects the topology and opens its xGMI
        # rings, which has never been timed here; a false abort would cost the run, a true hang still ends)
        t_comm = max(t_first, float(os.environ.get("LNB_COMM_INIT_TIMEOUT", "180")))
        dog = threading.Timer(t_comm, lambda: (sys.stderr.write("bench.py --gpus %d ABORTED: rank %d's ncclCommInitRank did not return within %.0f s "
                                                                "(a peer missing, or no peer-to-peer path between the GPUs)\n" % (world, rank, t_comm)), os._exit(4)))
        dog.daemon = True
        try:
            uid = grp.broadcast(lnb.Pipeline.unique_id() if (rank == 0 and world > 1) else None)
            if world > 1:
                dog.start()
            pipe = lnb.Pipeline(stage.model, rank, world, uid)
        except Exception as e:                                # (every rank must take the same path: agree below)
            why = "%s: %s" % (type(e).__name__, e)
            ok = 0
        dog.cancel()
        counts = grp.all_reduce([(rank, pipe.comm_count() if pipe is not None else 0, why)], lambda vs: sorted(sum(vs, [])))
        if world > 1 and any(c != world for _, c, _ in counts):
            ok = 0
        if grp.all_reduce(ok, min) == 0:
            details = ["rank %d: communicator of %d ranks (wanted %d)%s" % (r, c, world, "" if not w else " -- " + w) for r, c, w in counts]
            if os.environ.get("LNB_PIPELINE_FALLBACK") != "torch":
                abort_all(rank, "while forming the %d-rank RCCL communicator (LNB_PIPELINE_FALLBACK=torch retries through torch.distributed)" % world, details)
            sys.stderr.write("[rank %d] native RCCL exchange unavailable (%s): falling back to torch.distributed\n" % (rank, "; ".join(details)))
            if pipe is not None:
                pipe.close()
            stage.close(); stage = None
            grp.close()
            exchange = "torch (native RCCL init failed on some rank)"
        else:
            grp.set_timeout(300.0)
            preflight_rec = {"ranks": pre, "rccl_comm_count_per_rank": [c for _, c, _ in counts]} if pre else None
    if exchange == "native":
        prompts = [lnb.synth_tokens(99 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
        n_decode = W + K
        t_split, t_end = n_seq * (1 + W), n_seq * (1 + W + K)

        def slots_tokens(sl):
            return [int(t) for t in pipe.read_tokens(sl[0], len(sl))] if sl and sl == list(range(sl[0], sl[0] + len(sl))) else [int(pipe.read_tokens(q, 1)[0]) for q in sl]

        def measure():
            """untimed prefill + W warm-up rounds, then EXACTLY K timed decode rounds of the n_seq sequences in flight, bracketed by a device
            sync + barrier on both sides, max over ranks; then the same model with ONE sequence (single-stream) on a fresh context"""
            st = run_ticks_native(rank, world, pipe, stage.ctx, prompts, n_decode, 0, t_split)
            pipe.sync(); grp.barrier()
            t0 = time.perf_counter()
            run_ticks_native(rank, world, pipe, stage.ctx, prompts, n_decode, t_split, t_end, st)
            t_host = time.perf_counter() - t0                # host time of ENQUEUEING the K*n_seq ticks of the timed window
            pipe.sync(); grp.barrier()
            wall = grp.all_reduce(time.perf_counter() - t0, max)
            toks0 = slots_tokens(st["slots"][0]) if rank == world - 1 else None
            toks_all = None
            if rank == world - 1:
                try:
                    toks_all = [slots_tokens(st["slots"][q]) for q in range(n_seq)]
                except lnb.LnbError:                          # (log wrapped on a very long run: sequence 0 above was read first and still stands)
                    toks_all = None
            # single stream: sequence 0 again on its (reset) context, W_s warm-up + K_s timed steps
            Ks, Ws = min(K, int(os.environ.get("LNB_SINGLE_STREAM_STEPS", "32"))), min(W, 4)
            c0 = stage.ctx[0]; c0.reset()
            c0.set_schedule("latency")                       # ONE sequence in the pipe: nothing to share the CUs with, the single stream's forms
            ss = run_single_stream_native(rank, world, pipe, c0, prompts[0], Ws + Ks, 0, 1 + Ws)
            pipe.sync(); grp.barrier()
            t1 = time.perf_counter()
            run_single_stream_native(rank, world, pipe, c0, prompts[0], Ws + Ks, 1 + Ws, 1 + Ws + Ks, ss)
            pipe.sync(); grp.barrier()
            wall_s = grp.all_reduce(time.perf_counter() - t1, max)
            toks_s = slots_tokens(ss["slots"]) if rank == world - 1 else None
            c0.set_schedule(sched)
            info = grp.all_reduce([(rank, pipe.comm_count(), toks0, toks_s, toks_all)], lambda vs: sorted(sum(vs, [])))
            toks0 = info[-1][2]; toks_s = info[-1][3]; toks_all = info[-1][4]
            return {"wall": wall, "host_enqueue_us_per_tick": round(1e6 * t_host / max(1, K * n_seq), 1),
                    "rccl_comm_count_per_rank": [c for _, c, _, _, _ in info], "tokens_seq0": toks0, "tokens_all": toks_all,
                    "single_stream": {"tokens_per_s": round(Ks / wall_s, 2), "steps": Ks, "ms_per_token": round(1e3 * wall_s / Ks, 4),
                                      "tokens_equal_sequence0_of_the_batch": bool(toks_s is not None and toks0 is not None and toks_s[:len(toks0)] == toks0[:len(toks_s)])}}

        def measure_batched(nb):
            """the same pipeline with BATCHES of nb sequences as the unit that moves through the stages (lnb_pipeline_tick_batch): 2*world groups in
            flight, each decode step of a group = ONE pass over a stage's weights for its nb sequences (exact per sequence).  Whole-block stages."""
            lb, le = stage_layers(rank, world, cfg["n_layers"], head_cost=(costs[3] / max(1e-9, sum(costs[:3])) if costs else 1.2))
            G = 2 * world if world > 1 else int(os.environ.get("LNB_PIPELINE_SEQS", "2"))
            # every rank must take the same path through the collectives below: set the stage up, then AGREE that it worked everywhere
            # (an allocation can fail on one rank only) before anything is exchanged
            st, err, copy = None, None, True
            try:
                st = LnbStage(lnb, None, cfg, rank, world, G * nb, seq_len, local, parts=(3 * lb, 3 * le), costs=costs)
                try:
                    st.model.enable_batch()
                except lnb.LnbError:                         # no room for the matrix-core copy on this rank (the head rank holds the extra output.weight
                    copy = False                             # copy): its batches run as rows on the resident layouts, the hand-off is the same [n, dim]
            except lnb.LnbError as e:
                err = str(e)
            if grp.all_reduce(0 if err else 1, min) == 0:
                errs = grp.all_reduce([(rank, err)], lambda vs: sorted(sum(vs, [])))
                if st is not None:
                    st.close()
                return {"skipped": "; ".join("rank %d: %s" % (r, e) for r, e in errs if e)}
            uid2 = grp.broadcast(lnb.Pipeline.unique_id() if (rank == 0 and world > 1) else None)
            pp = lnb.Pipeline(st.model, rank, world, uid2)
            prm = [lnb.synth_tokens(99 + q, P, cfg["vocab_size"]) for q in range(G * nb)]          # sequence 0 = the headline's prompt
            first_slot0, first_slots = None, []
            for q in range(G * nb):
                sl = prefill_through_pipeline(rank, world, pp, st.ctx[q], prm[q])
                first_slots.append(sl)
                if q == 0:
                    first_slot0 = sl
            pp.sync(); grp.barrier()
            bats = [lnb.Batch(st.ctx[g * nb:(g + 1) * nb]).set_state(None, [P] * nb) for g in range(G)]
            n_dec = W + K
            sb = run_ticks_native_batched(rank, world, pp, bats, n_dec, 0, G * W)
            pp.sync(); grp.barrier()
            t0 = time.perf_counter()
            run_ticks_native_batched(rank, world, pp, bats, n_dec, G * W, G * (W + K), sb)
            t_host = time.perf_counter() - t0
            pp.sync(); grp.barrier()
            wall_b = grp.all_reduce(time.perf_counter() - t0, max)
            for b_ in bats:
                b_.check_error()                             # ticks only enqueue: a bad token / all-NaN row is latched on the device, not raised
            toks0 = None
            if rank == world - 1:
                toks0 = [int(pp.read_tokens(first_slot0, 1)[0])] + [int(pp.read_tokens(q, 1)[0]) for q in sb["slots"][0]]
            multi = None
            if rank == world - 1:                            # every sequence of every group: its prefill token + the group's steps (a step's nb tokens are one contiguous run of the log)
                try:
                    steps = [[pp.read_tokens(q, nb) for q in sb["slots"][g_]] for g_ in range(G)]
                    multi = _multi_golden_check([[int(pp.read_tokens(first_slots[g_ * nb + j_], 1)[0])] + [int(stp[j_]) for stp in steps[g_]] for g_ in range(G) for j_ in range(nb)], P, name)
                except lnb.LnbError as e:                    # (a long run: the log keeps the newest 65536 tokens, the early slots are gone -- no rank may leave the collectives for that)
                    multi = {"skipped": str(e)[:200]}
            info = grp.all_reduce([(rank, pp.comm_count(), toks0)], lambda vs: sorted(sum(vs, [])))
            multi = grp.all_reduce([(rank, multi)], lambda vs: sorted(sum(vs, [])))[-1][1]
            cuts = grp.all_reduce([(rank, lb, le, copy)], lambda vs: sorted(sum(vs, [])))
            res_b = {"wall": wall_b, "sequences_in_flight": G * nb, "groups": G, "batch": nb, "blocks_per_gpu": [e_ - b_ for _, b_, e_, _ in cuts],
                     "second_weight_copy_per_rank": [bool(c_) for _, _, _, c_ in cuts],
                     "cut": "whole blocks (lnb_model_enable_batch refuses a stage cut inside a block: the batched hand-off is [n, dim] only)",
                     "tokens_per_s": round(K * G * nb / wall_b, 2), "host_enqueue_us_per_tick": round(1e6 * t_host / max(1, K * G), 1),
                     "rccl_comm_count_per_rank": [c for _, c, _ in info], "tokens_vs_oracle_golden": _golden_check(info[-1][2], P, name) if info[-1][2] else None,
                     "sequences_vs_oracle_golden": multi}
            for b_ in bats:
                b_.close()
            pp.close(); st.close()
            return res_b

        m_bal = measure()
        wall = m_bal["wall"]
        toks0 = m_bal.pop("tokens_seq0")
        toks_all = m_bal.pop("tokens_all")
        extra = {"exchange": "RCCL point-to-point inside the library (lnb_pipeline_tick), stage steps as captured graphs",
                 "sequences_vs_oracle_golden": _multi_golden_check(toks_all, P, name),
                 "host_enqueue_us_per_tick": m_bal["host_enqueue_us_per_tick"], "rccl_comm_count_per_rank": m_bal["rccl_comm_count_per_rank"],
                 "single_stream": m_bal["single_stream"], "tokens_vs_oracle_golden": _golden_check(toks0, P, name) if toks0 else None}
        if rank == world - 1 and os.environ.get("LNB_PIPELINE_DUMP_TOKENS"):
            json.dump({"0": toks0}, open(os.environ["LNB_PIPELINE_DUMP_TOKENS"], "w"))
        pipe.close()
        # configs[3] literally -- n_layers / N whole blocks per GPU -- next to the cost-balanced cut above (the headline): same run again
        if world > 1 and cfg["n_layers"] % world == 0 and os.environ.get("LNB_PIPELINE_LITERAL_SPLIT", "1") != "0":
            stage.close()
            stage = LnbStage(lnb, None, cfg, rank, world, n_seq, seq_len, local, parts=blocks_split(rank, world, cfg["n_layers"]), costs=costs)
            for c in stage.ctx:
                c.set_mode(mode).set_schedule(sched)
            uid = grp.broadcast(lnb.Pipeline.unique_id() if rank == 0 else None)
            pipe = lnb.Pipeline(stage.model, rank, world, uid)
            m_lit = measure()
            m_lit.pop("tokens_seq0")
            extra["literal_blocks_split"] = dict(m_lit, blocks_per_gpu=cfg["n_layers"] // world, tokens_per_s=round(K * n_seq / m_lit.pop("wall"), 2))
            pipe.close()
        nb = int(os.environ.get("LNB_PIPELINE_BATCH", "64" if mode == "exact" else "0"))      # (more than 16: the groups are rows of the streaming product, NOTES 5.11; 64: 4.7 tokens per ms of an 8B-shape step against 3.1 at 32)
        mb = None
        if nb > 0:
            mb = measure_batched(min(nb, 128))
            extra["batched"] = dict(mb)
            extra["batched"].pop("wall", None)
        # The line's `value` is the UNBATCHED figure -- 2N independent sequences in flight, one-token ticks, the kernels of the N = 1 line --
        # so that the 1 -> 8 curve compares like with like; the batched pipeline (a serving workload: one pass over a stage's weights per
        # group of sequences) is reported next to it, each with a same-workload ONE-GPU anchor measured in this run on rank 0's GPU.
        extra["value_definition"] = ("%d independent sequences in flight, one-token ticks through the %d stages (exact single-sequence kernels in their %s forms: "
                                     "lnb_ctx_set_schedule; the N = 1 line's sequences_in_flight section is the same regime on one GPU)" % (n_seq, world, sched))
        extra["schedule"] = sched
        extra["preflight"] = preflight_rec
        extra["value_single_stream"] = m_bal["single_stream"]["tokens_per_s"]
        extra["value_batched"] = mb.get("tokens_per_s") if isinstance(mb, dict) else None
        tick_us = 1e6 * wall / max(1, K * n_seq)
        pred = []
        for r in range(world):
            pb_, pe_ = stage_parts(r, world, cfg["n_layers"], *stage.costs)
            c3 = stage.costs
            pred.append(round(sum(c3[q % 3] for q in range(pb_, pe_)) + (c3[3] if r == world - 1 else 0.0), 1))
        extra["stage_time_us"] = {"predicted_per_rank_from_the_probe": pred, "slowest_predicted": max(pred), "measured_tick": round(tick_us, 1),
                                  "note": "a tick = one sequence's one-token step on one stage; with 2N sequences in flight the pipeline's period is the slowest stage"}
        if os.environ.get("LNB_PIPELINE_ANCHOR", "1") != "0":
            extra["one_gpu_anchor"] = one_gpu_anchor(lnb, grp, rank, world, cfg, local, mode, P, W, K, n_seq, seq_len,
                                                     (mb["groups"], mb["batch"]) if isinstance(mb, dict) and "groups" in mb else None, sched)
            an = extra["one_gpu_anchor"]
            if an.get("unbatched_tokens_per_s"):
                extra["efficiency_vs_one_gpu"] = {"unbatched": round((K * n_seq / wall) / (world * an["unbatched_tokens_per_s"]), 4)}
                if an.get("batched_tokens_per_s") and extra["value_batched"]:
                    extra["efficiency_vs_one_gpu"]["batched"] = round(extra["value_batched"] / (world * an["batched_tokens_per_s"]), 4)
                extra["efficiency_vs_one_gpu"]["definition"] = "value_N / (N x the same number of sequences in flight on ONE GPU holding the whole model)"
        grp.close()
    else:
        import datetime
        import torch
        import torch.distributed as dist
        if backend == "gloo":
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
        device = "cuda:%d" % local
        patience = datetime.timedelta(seconds=300)               # a lost peer fails the run instead of hanging it

        def probe(bcast_device):
            if not (world > 1 and os.environ.get("LNB_PIPELINE_PROBE", "1") != "0"):
                return None
            t = torch.zeros(4, dtype=torch.float64, device=bcast_device)
            if rank == 0:
                t += torch.tensor(probe_costs(lnb, cfg, local, P + W + K // 2), dtype=torch.float64).to(t.device)
            dist.broadcast(t, 0)
            return [float(v) for v in t.tolist()]

        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device), timeout=patience)   # rank -> GPU mapping is explicit
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)
        # 2*world sequences: the exchange of a tick overlaps the compute of another sequence's item (run_ticks); LNB_PIPELINE_OVERLAP=0
        # falls back to the lock-step schedule with `world` sequences
        n_seq = world * (2 if world > 1 and os.environ.get("LNB_PIPELINE_OVERLAP", "1") != "0" else 1)
        costs = probe(device if backend == "nccl" else "cpu")
        stage = LnbStage(lnb, torch, cfg, rank, world, n_seq, seq_len, local, costs=costs)
        for c in stage.ctx:
            c.set_mode(mode).set_schedule(sched)
        prompts = [lnb.synth_tokens(99 + s, P, cfg["vocab_size"]) for s in range(n_seq)]
        n_decode = W + K
        t_split = n_seq * (1 + W)              # prefill phase + W warm-up decode rounds
        t_end = n_seq * (1 + W + K)
        state = run_ticks(rank, world, stage, dist, torch, prompts, n_decode, device, 0, t_split)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        # the timed window: every rank runs exactly K*n_seq items (K decode steps of each sequence in flight)
        run_ticks(rank, world, stage, dist, torch, prompts, n_decode, device, t_split, t_end, state)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        tmax = torch.tensor([wall], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
        extra = {"exchange": "torch.distributed batch_isend_irecv through staging tensors (%s)" % exchange}

        def measure_batched_torch(nb):
            """the batched pipeline of the native path (measure_batched) with the exchange through torch.distributed: groups of nb sequences,
            2*world groups in flight, whole-block stages, a pipe without a transport (run_ticks_batched_torch)"""
            lb, le = stage_layers(rank, world, cfg["n_layers"], head_cost=(costs[3] / max(1e-9, sum(costs[:3])) if costs else 1.2))
            G = 2 * world
            # Every rank agrees that its set-up worked BEFORE any rank enters a collective (a rank whose LnbStage / Batch / Pipeline raised
            # used to leave while the others blocked in prefill_torch's recv until the backend's timeout -- the hang the preflight exists to remove).
            # The unbatched stage is closed first: its weights, contexts and graphs are not needed any more and the second stage wants the memory.
            cpu_dev = device if backend == "nccl" else "cpu"

            def all_ok(ok, what):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cpu_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return None if int(flag.item()) == 1 else {"skipped": "%s failed on at least one rank (this rank: %s)" % (what, "ok" if ok else "failed")}
            st2 = pp = None
            bats = []
            copy = True
            err = None
            try:
                stage.close()
                st2 = LnbStage(lnb, torch, cfg, rank, world, G * nb, seq_len, local, parts=(3 * lb, 3 * le), costs=costs)
                try:
                    st2.model.enable_batch()
                except lnb.LnbError:
                    copy = False                             # (no room for the matrix-core copy on this rank: its batches run as rows on the resident layouts)
                pp = lnb.Pipeline(st2.model, rank, world, host_transport=True)
            except Exception as e:                           # noqa: BLE001 -- whatever it is, the other ranks must hear about it
                err = "%s: %s" % (type(e).__name__, e)
            skipped = all_ok(err is None, "batched stage set-up")
            if skipped:
                if err:
                    skipped["this_rank_error"] = err[:300]
                if pp:
                    pp.close()
                if st2:
                    st2.close()
                return skipped
            prm = [lnb.synth_tokens(99 + q, P, cfg["vocab_size"]) for q in range(G * nb)]
            firsts = prefill_torch(rank, world, st2, dist, torch, prm, device)
            try:
                bats = [lnb.Batch(st2.ctx[g * nb:(g + 1) * nb]).set_state(firsts[g * nb:(g + 1) * nb] if rank == 0 else None, [P] * nb) for g in range(G)]
            except Exception as e:                           # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            skipped = all_ok(err is None, "batch creation")
            if skipped:
                if err:
                    skipped["this_rank_error"] = err[:300]
                for b_ in bats:
                    b_.close()
                pp.close(); st2.close()
                return skipped
            sb = run_ticks_batched_torch(rank, world, pp, bats, dist, torch, W + K, device, cfg["dim"], 0, G * W)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_ticks_batched_torch(rank, world, pp, bats, dist, torch, W + K, device, cfg["dim"], G * W, G * (W + K), sb)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            wb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(wb, op=dist.ReduceOp.MAX)
            for b_ in bats:
                b_.check_error()
            box = [None]
            if rank == world - 1:
                box[0] = [firsts[0]] + [int(pp.read_tokens(q, 1)[0]) for q in sb["slots"][0]]
            dist.broadcast_object_list(box, src=world - 1)
            res_b = {"sequences_in_flight": G * nb, "groups": G, "batch": nb, "tokens_per_s": round(K * G * nb / float(wb.item()), 2),
                     "second_weight_copy_on_this_rank": copy, "cut": "whole blocks (the batched hand-off is [n, dim] only)",
                     "exchange": "pipe without a transport (lnb_pipeline_init_host) + torch.distributed batch_isend_irecv of lnb_batch_boundary_ptr's buffers",
                     "tokens_vs_oracle_golden": _golden_check(box[0], P, name) if box[0] else None, "tokens_seq0": box[0][:8] if box[0] else None}
            for b_ in bats:
                b_.close()
            pp.close(); st2.close()
            return res_b

        nb = int(os.environ.get("LNB_PIPELINE_BATCH", "16" if mode == "exact" else "0"))
        if world > 1 and nb > 0:
            extra["batched"] = measure_batched_torch(min(nb, 128))
            extra["value_batched"] = extra["batched"]["tokens_per_s"]
    if rank == 0:
        import bench as _b
        tokens = K * n_seq
        tps = tokens / wall
        a = {k: cfg[k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size")}
        Tbar = P + W + (K - 1) / 2.0 + 1.0
        B = _b.algorithmic_bytes_per_token(a, stage.model.ffn_hidden, Tbar)
        res = {"metric": "decode tokens/s Llama-3.1-8B bf16 @1/2/4/8 MI355X; % HBM roofline", "value": round(tps, 2), "unit": "tokens/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1000.0 * wall / K, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               # `value` changed meaning between rounds (r03: the batched pipeline; r04: unbatched ticks, latency forms; r05: unbatched ticks, throughput forms):
               # files of different schema versions are not like for like
               "schema": "lnb-bench-multigpu/3 (value = unbatched one-token ticks, throughput kernel forms; batched and single-stream figures under config)",
               "config": dict({"workload": "%s bf16, %dxMI355X layer pipeline (blocks per GPU %s), RCCL p2p hidden-state hand-off, %d sequences in flight, "
                                           "prompt %d -> +%d tokens each" % (name, world, ",".join("%.3g" % ((stage_parts(r, world, cfg["n_layers"], *stage.costs)[1]
                                                                                                                - stage_parts(r, world, cfg["n_layers"], *stage.costs)[0]) / 3.0)
                                                                                                     for r in range(world)), n_seq, P, K),
                                "prompt_len": P, "sequences_in_flight": n_seq, "parallelism": "pp%d" % world,
                                "part_costs_us": [round(v, 1) for v in stage.costs],
                                "mode": "exact-order (token-id identical to the CPU reference path)" if mode == "exact" else "fast (opt-in tolerance mode)"}, **extra),
               "roofline": {"bound": "hbm", "achieved": round(tps * B / 1e9, 1), "peak": _b.PEAK_HBM_GBS * world, "unit": "GB/s",
                            "frac": round(tps * B / 1e9 / (_b.PEAK_HBM_GBS * world), 4), "traffic": None,
                            "note": "whole job: tokens/s x algorithmic bytes per token over N x 8 TB/s"}}
        if getattr(args, "cpu_steps", 0) > 0:                # the same bounded CPU sample as the one-GPU line (after the timed regions; the other ranks are done)
            res["cpu_baseline"] = _b.cpu_baseline(cfg, lnb.synth_tokens(99, P, cfg["vocab_size"])[:8], args.cpu_steps)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    os.close(json_fd)
    stage.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0
