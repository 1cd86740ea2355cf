// lnb_batch_kernels.h -- batched EXACT decode: up to 128 independent sequences per pass over the weights (16 as matrix-instruction columns, more as rows).  Included at the end of
// lnb_kernels.hip (one translation unit: it reuses the RMSNorm prologue, the epilogue arithmetic and the ring-load idiom defined there).
//
// Reference: the reference shares W across the rows of a call (src/ml/operations_lineartransform.go:173-193) and creates one context per
// generation (src/inference/inference.go:174); N generations in flight are N independent one-token Forward calls per step.  Here their
// tokens are the COLUMNS of one matrix product: v_mfma_f32_16x16x4_f32 is bit for bit the reference's k-ordered chain
// acc = fma(x_k, w_k, acc) for each of its 16 x 16 outputs (NOTES.md 5.6, tools/mfma_exact.hip), so column s carries sequence s's chains
// unchanged -- same bits as its single-sequence run -- while the weights are streamed from HBM ONCE for all of them.
//
//   mfma_stream_kernel   weights (M16 layout, lnb_device.h) HBM -> VGPR -> matrix-core A operand, activations of the batch (B-operand
//                        layout "xt") L2 -> VGPR -> B operand; no LDS, no barrier, every wave independent and persistent over its jobs;
//                        epilogues of the decode kernels per column (RoPE + KV append with the column's own position and cache,
//                        SiLU*up, residual) -- the next product's activations are written straight in the B-operand layout;
//   batch_rmsnorm_xt_kernel   one workgroup per sequence: the exact parallel norm sum of the GEMV prologue (lnb_seqsum.h), output in xt;
//   attn_exact_kernel    (lnb_kernels.hip) takes the per-sequence position / caches from the batch tables;
//   batch_embed_kernel, batch_argmax_kernel   per-sequence token feedback: each context keeps its own position, token word and log.
#pragma once

// ---- one-time re-tile of a resident matrix into the M16 layout (lnb_model_enable_batch) -------------------------------------------
__global__ void m16_from_tiled_kernel(const uint16_t* src, uint16_t* dst, int rows, int K, int RW, int NCH) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)rows * NCH * K;
    if (idx >= total) return;
    const int k = (int)(idx % K); const size_t rc = idx / K; const int c = (int)(rc % NCH), n = (int)(rc / NCH);
    dst[m16_index(n, k, c, K, NCH)] = src[tiled_index(n, k, c, K, RW, NCH)];
}

// ---- ring loads (asm: hipcc's waitcnt pass drains vmcnt(0) around loop-carried register prefetch; retired by hand-counted waits) -----
// the immediate offset selects the unit m of the chunk: it must be a literal in the asm text
template <int M> DEVINL void ld_unit_nt(u32x4& d, unsigned voff, const char* sb) {
    static_assert(M >= 0 && M < 4, "unit");
    if constexpr (M == 0) asm volatile("global_load_dwordx4 %0, %1, %2 nt ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 nt ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 3) asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 nt ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
}
template <int M> DEVINL void ld_unit(u32x4& d, unsigned voff, const char* sb) {       // activations: every CU reads them, keep them cached
    static_assert(M >= 0 && M < 4, "unit");
    if constexpr (M == 0) asm volatile("global_load_dwordx4 %0, %1, %2 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 3) asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
}
template <int N, int L> DEVINL void wait_chunk(u32x4 (&b)[L]) {
    static_assert(L == 8 || L == 12, "loads per chunk");
    if constexpr (L == 8) asm volatile("s_waitcnt vmcnt(%8) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "n"(N) : "memory");
    if constexpr (L == 12) asm volatile("s_waitcnt vmcnt(%12) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]) : "n"(N) : "memory");
}
DEVINL float unit_elem(const u32x4& v, int e) { const uint32_t d = v[e >> 1]; return __uint_as_float((e & 1) ? (d & 0xFFFF0000u) : (d << 16)); }

// epilogue of one 16-row tile-chain: this lane holds column s (a sequence), output rows n0 .. n0+3 (n0 % 4 == 0)
template <int EPI> DEVINL void stream_epilogue(const StreamParams& p, const f32x4& g, const f32x4& u, int s, int n0) {
    if (s >= p.nseq) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int n = n0 + r;
        if (n >= p.n_rows) continue;
        const size_t o = (size_t)s * p.n_rows + n;
        if (EPI == EPI_STORE) p.out[o] = bf_trunc(g[r]);
        else if (EPI == EPI_RESID) p.out[o] = bf_trunc(bf_wide(p.res[o]) + bf_wide(bf_trunc(g[r])));                    // ml.Add, impl:320-332
        else if (EPI == EPI_SILU_MUL) {                                                                                   // activations.go:36-39, :614
            const uint16_t gs = bf_trunc(p.silu[bf_trunc(g[r])]);
            p.out_xt[xt_index(s, n)] = bf_trunc(bf_wide(gs) * bf_wide(bf_trunc(u[r])));
        } else if (EPI == EPI_QKV_ROPE) {                                                                                 // llamatransformer.go:297-403
            const int pos = p.tab->st[s]->pos;                                                                            // this sequence's own position
            const uint16_t mine = bf_trunc(g[r]), other = bf_trunc(g[r ^ 1]);                                             // RoPE partner 2i <-> 2i+1: same lane
            if (n < p.q_dim + p.kv_dim) {
                const int d = n % p.head_dim, i = d >> 1;
                const float2 cs = *(const float2*)(p.cis + ((size_t)pos * (p.head_dim >> 1) + i) * 2);
                const double cr = (double)cs.x, ci = (double)cs.y;
                uint16_t r16;
                if ((n & 1) == 0) { const double a = (double)bf_wide(mine), bb = (double)bf_wide(other); r16 = bf_trunc((float)(a * cr - bb * ci)); }
                else              { const double a = (double)bf_wide(other), bb = (double)bf_wide(mine); r16 = bf_trunc((float)(a * ci + bb * cr)); }
                if (n < p.q_dim) p.q_out[(size_t)s * p.q_dim + n] = r16;
                else {                                                                                                    // :402, K cache [kv head][d/8][position][8]
                    const int kc = n - p.q_dim, kh = kc / p.head_dim;
                    p.kv->ck[s][(((size_t)kh * (p.head_dim >> 3) + (d >> 3)) * p.tab->seq_len[s] + pos) * 8 + (d & 7)] = r16;
                }
            } else p.kv->cv[s][(size_t)pos * p.kv_dim + (n - p.q_dim - p.kv_dim)] = mine;                                 // :403
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mfma_stream_kernel<ACC, EPI>: a wave owns ACC tile-chains at a time (job j = tile-chains j*ACC .. j*ACC + ACC-1: the gate and up chains of
// one tile of w1|w3, two tiles of a fat matrix, or one tile of a thin one) and walks K in 128-step chunks: 4 x 16 B weight loads per chain
// and 4 x 16 B activation loads per lane and chunk, R chunks in flight (hand-counted ring), then per chunk 32 k-groups in order g = 4e + m:
// unpack (one VALU op per operand: bf16 -> f32 is a shift or a mask) in batches of 4*EB groups AHEAD of runs of back-to-back MFMAs.
// Measured (tools/mfma_stream_bench.hip, MI355X): a dependent v_mfma_f32_16x16x4_f32 chain issues every 32.4 cycles back to back, 52
// with ONE VALU op between two of them (the accumulator forwarding is lost), 36.6-39 with the VALU in batches; so the unpacks never sit
// between two MFMAs of a run.  One wave per SIMD; persistent over its jobs (the load ring runs across job boundaries).
// grid.x = min(ceil(n_jobs / 4), CUs), block 256, no LDS.
// ------------------------------------------------------------------------------------------------
template <int ACC, int EPI>
__global__ __launch_bounds__(256) void mfma_stream_kernel(StreamParams p) {
    constexpr int R = ACC == 1 ? 4 : 3, L = ACC * 4 + 4, EB = ACC == 1 ? 8 : 4;     // (two chains with R = 4 need 260+ registers: values start living in AGPRs)
    static_assert(R * L <= 60, "vmcnt is a 6-bit counter");
    static_assert(EPI != EPI_SILU_MUL || ACC == 2, "gate and up chains of a tile travel together");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = blockIdx.x * 4 + wave, TW = gridDim.x * 4;
    const int nchunks = p.K >> 7;
    if (gw >= p.n_jobs) return;
    const long long t_begin = p.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int njobs_mine = (p.n_jobs - gw + TW - 1) / TW;
    const int T = njobs_mine * nchunks;
    const size_t chain_bytes = (size_t)nchunks * 4096;
    const unsigned aoff = (unsigned)(((lane & 15) * 4 + (lane >> 4)) * 16), boff = (unsigned)lane * 16u;
    const int last_chain = p.n_chains - 1;
    u32x4 buf[R][L];
    // issue cursor: next chunk to load = (job ij, chunk ic); past the end it stays put and re-reads
    int ij = gw, ic = 0, issued = 0;
    auto issue_next = [&](u32x4 (&dst)[L]) {
        const char* xb = (const char*)p.xt + (size_t)ic * 4096;
#pragma unroll
        for (int a = 0; a < ACC; a++) {
            int tc = ij * ACC + a; tc = tc < last_chain ? tc : last_chain;          // (a ragged last job loads a valid chain and drops the result)
            const char* wb = (const char*)p.w + (size_t)tc * chain_bytes + (size_t)ic * 4096;
            ld_unit_nt<0>(dst[a * 4 + 0], aoff, wb); ld_unit_nt<1>(dst[a * 4 + 1], aoff, wb);
            ld_unit_nt<2>(dst[a * 4 + 2], aoff, wb); ld_unit_nt<3>(dst[a * 4 + 3], aoff, wb);
        }
        ld_unit<0>(dst[ACC * 4 + 0], boff, xb); ld_unit<1>(dst[ACC * 4 + 1], boff, xb);
        ld_unit<2>(dst[ACC * 4 + 2], boff, xb); ld_unit<3>(dst[ACC * 4 + 3], boff, xb);
        if (issued + 1 < T) { issued++; if (++ic == nchunks) { ic = 0; ij += TW; } }
    };
#pragma unroll
    for (int j = 0; j < R; j++) issue_next(buf[j]);
    f32x4 acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c = 0, job = gw;
    for (int t0 = 0; t0 < T; t0 += R) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (t0 + j < T) {
                wait_chunk<(R - 1) * L, L>(buf[j]);          // chunk t0+j landed; R-1 younger ones stay in flight
#pragma unroll
                for (int e0 = 0; e0 < 8; e0 += EB) {
                    float av[ACC][4 * EB], bv[4 * EB];
#pragma unroll
                    for (int ee = 0; ee < EB; ee++)
#pragma unroll
                        for (int m = 0; m < 4; m++) {
#pragma unroll
                            for (int a = 0; a < ACC; a++) av[a][ee * 4 + m] = unit_elem(buf[j][a * 4 + m], e0 + ee);
                            bv[ee * 4 + m] = unit_elem(buf[j][ACC * 4 + m], e0 + ee);
                        }
                    if (e0 + EB == 8) {
                        // every register of the slot has been read: refill it.  The unpacked operands are pinned in front of the refill --
                        // otherwise hipcc sinks unpack ops below the asm that reloads the slot and keeps the old value alive through a
                        // register copy made BEFORE the wait (seen in the ISA of the first version: v_mov of in-flight registers)
#pragma unroll
                        for (int q = 0; q < 4 * EB; q++) {
#pragma unroll
                            for (int a = 0; a < ACC; a++) asm volatile("" : "+v"(av[a][q]));
                            asm volatile("" : "+v"(bv[q]));
                        }
                        issue_next(buf[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4 * EB; q++)         // k-groups g = 4 (e0 + ee) + m ascending: the reference's k order (operations_lineartransform.go:46-65)
#pragma unroll
                        for (int a = 0; a < ACC; a++) {
                            // the first k-group of a job starts from C = 0 IN the instruction (an inline constant): zeroing the accumulator
                            // registers after the epilogue instead made them a loop-carried phi of {matrix-core result, 0} and hipcc moved all
                            // of them AGPR -> VGPR -> AGPR at every chunk boundary
                            if (e0 == 0 && q == 0 && c == 0) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][q], bv[q], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][q], bv[q], acc[a], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (++c == nchunks) {                        // end of this job's chains: D layout = column lane & 15, rows (lane >> 4) * 4 + r
                    const int s = lane & 15, tc0 = job * ACC;
                    if (EPI == EPI_SILU_MUL) {
                        if (tc0 + 1 <= last_chain) stream_epilogue<EPI>(p, acc[0], acc[ACC - 1], s, (tc0 / 2) * 16 + (lane >> 4) * 4);
                    } else {
#pragma unroll
                        for (int a = 0; a < ACC; a++)
                            if (tc0 + a <= last_chain) stream_epilogue<EPI>(p, acc[a], acc[a], s, (tc0 + a) * 16 + (lane >> 4) * 4);
                    }
                    c = 0; job += TW;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
    if (p.dbg && lane == 0) p.dbg[gw] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
}

// ------------------------------------------------------------------------------------------------
// mfma_pair_kernel<EPI> (round 4): the thin matrices of a batch of up to 16 sequences (wq|wk|wv, wo, w2: one 16-row tile per SIMD, ONE
// k-ordered chain of K / 4 dependent matrix instructions each).  In mfma_stream_kernel the wave that owns the chain also unpacks its
// operands (64 vector ops per 128 k-steps) and issues its loads (8 x 1 KiB): 1554 cycles per chunk of which the 32 matrix instructions are
// 1037 -- 12.1 cycles per k-step.  Here the role split of the single stream's GEMVs: wave w (0, 1) only issues matrix instructions; wave
// w + 2 -- another SIMD -- streams the weights and activations through the hand-counted ring, unpacks them and leaves the f32
// operands in the LDS in consumption order ([read r][lane][A 2r, A 2r+1, B 2r, B 2r+1]: sixteen ds_read_b128 per chunk, each feeding two
// matrix instructions as it lands); two LDS buffers per pair, one workgroup barrier per chunk.
// Two tiles per CU (the thin matrices have 256-384 tiles: 128-192 of the 256 CUs).  grid.x = min(ceil(n_jobs / 2), CUs), block 256, dynamic LDS 64 KB.  Same chains, same epilogues, same bits as mfma_stream_kernel<1, EPI>.
// ------------------------------------------------------------------------------------------------
constexpr int MP_BUF = 16 * 64 * 16;                        // one chunk's operands of one pair: 16 KB
template <int EPI>
__global__ __launch_bounds__(256) void mfma_pair_kernel(StreamParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = 4, L = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pr = wave & 1;                                 // the pair: its chain wave is wave pr, its helper wave pr + 2 -- on ANOTHER SIMD (waves 0..3 of a
                                                             // workgroup sit on four different SIMDs): a vector op of the helper between two matrix instructions of the
                                                             // chain costs the accumulator forwarding just like one of its own (measured: same-SIMD pairs, w2 132 us)
    const int gw = blockIdx.x * 2 + pr, TW = gridDim.x * 2;
    const int nchunks = p.K >> 7;
    // every wave of the workgroup runs the SAME number of chunks (one barrier per chunk): that of its first pair; a pair without a job left
    // streams a valid chain and drops the result
    const int G = p.n_groups > 1 ? p.n_groups : 1, n_units = p.n_jobs * G;       // unit u = (column group u / n_jobs, tile u % n_jobs): the groups of a tile run on different CUs at the same time
    const int T = ((n_units - (int)blockIdx.x * 2 + TW - 1) / TW) * nchunks;
    const size_t chain_bytes = (size_t)nchunks * 4096;
    const int last_chain = p.n_chains - 1;
    char* const lds = smem + (size_t)pr * 2 * MP_BUF + (size_t)lane * 16;      // this lane's 16 bytes of read r: + r * 1024 (+ MP_BUF: the other buffer)
    if (wave >= 2) {
        // ================================ helper: HBM -> registers -> unpack -> LDS ================================
        const unsigned aoff = (unsigned)(((lane & 15) * 4 + (lane >> 4)) * 16), boff = (unsigned)lane * 16u;
        u32x4 buf[R][L];
        int ij = gw, ic = 0, issued = 0;
        auto issue_next = [&](u32x4 (&dst)[L]) {
            const int uj = ij < n_units ? ij : n_units - 1;                         // (a pair without a unit left streams a valid one and drops the result)
            const char* xb = (const char*)p.xt + ((size_t)(uj / p.n_jobs) * 16 * (size_t)p.K) * 2 + (size_t)ic * 4096;
            int tc = uj % p.n_jobs; tc = tc < last_chain ? tc : last_chain;
            const char* wb = (const char*)p.w + (size_t)tc * chain_bytes + (size_t)ic * 4096;
            ld_unit_nt<0>(dst[0], aoff, wb); ld_unit_nt<1>(dst[1], aoff, wb); ld_unit_nt<2>(dst[2], aoff, wb); ld_unit_nt<3>(dst[3], aoff, wb);
            ld_unit<0>(dst[4], boff, xb); ld_unit<1>(dst[5], boff, xb); ld_unit<2>(dst[6], boff, xb); ld_unit<3>(dst[7], boff, xb);
            if (issued + 1 < T) { issued++; if (++ic == nchunks) { ic = 0; ij += TW; } }
        };
#pragma unroll
        for (int j = 0; j < R; j++) issue_next(buf[j]);
        // chunk t -> LDS buffer t & 1, k-groups g = 4 e + m in order: read r holds groups 2r, 2r + 1
        for (int t0 = 0; t0 < T + 1; t0 += R) {              // (T chunks; the helper runs one barrier ahead of the chain wave's first chunk)
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int t = t0 + j;
                if (t < T) {
                    wait_chunk<(R - 1) * L, L>(buf[j]);
                    float4 op[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int g0 = 2 * r, g1 = 2 * r + 1;
                        op[r] = make_float4(unit_elem(buf[j][g0 & 3], g0 >> 2), unit_elem(buf[j][g1 & 3], g1 >> 2),
                                            unit_elem(buf[j][4 + (g0 & 3)], g0 >> 2), unit_elem(buf[j][4 + (g1 & 3)], g1 >> 2));
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(op[r].x), "+v"(op[r].y), "+v"(op[r].z), "+v"(op[r].w));      // pinned in front of the refill (see mfma_stream_kernel)
                    issue_next(buf[j]);
                    char* dst = lds + (size_t)(t & 1) * MP_BUF;
#pragma unroll
                    for (int r = 0; r < 16; r++) *(float4*)(dst + r * 1024) = op[r];
                }
                if (t <= T) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); }   // barrier t: chunk t is in the LDS (t == T: the last chunk is being consumed)
            }
        }
        asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
        return;
    }
    // ==================================== chain wave: matrix instructions only ====================================
    __builtin_amdgcn_s_setprio(3);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int c = 0, job = gw;
    for (int t = 0; t < T; t++) {
        __builtin_amdgcn_s_barrier();                         // barrier t: chunk t is in buffer t & 1 (and everybody is done reading chunk t-1's buffer... of the step before)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const char* src = lds + (size_t)(t & 1) * MP_BUF;
        float4 op[16];
#pragma unroll
        for (int r = 0; r < 16; r++) op[r] = *(const float4*)(src + r * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; r++) {                        // k-groups ascending: the reference's k order (operations_lineartransform.go:46-65)
            if (r == 0 && c == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op[r].x, op[r].z, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);     // (C = 0 inside the instruction: see mfma_stream_kernel)
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op[r].x, op[r].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op[r].y, op[r].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++c == nchunks) {                                 // end of this job's chain: D layout = column lane & 15, rows (lane >> 4) * 4 + r
            if (job < n_units && job % p.n_jobs <= last_chain) stream_epilogue<EPI>(p, acc, acc, (job / p.n_jobs) * 16 + (lane & 15), (job % p.n_jobs) * 16 + (lane >> 4) * 4);
            c = 0; job += TW;
        }
    }
    __builtin_amdgcn_s_barrier();                             // barrier T (pairs with the helper's last one)
}

// ------------------------------------------------------------------------------------------------
// batch_rmsnorm_xt_kernel: RMSNorm (llamatransformer.go:633-660) of the batch's rows, one workgroup per sequence, with the EXACT parallel
// evaluation of the reference's serial sum of squares -- the prologue of the norm-fused GEMVs (rms_fold / rms_scale_wide, lnb_seqsum.h) --
// written out in the B-operand layout of the following product.  grid = nseq, block = (1 + NH) * 64, dynamic LDS = scratch + (kpad + 8) f32.
// ------------------------------------------------------------------------------------------------
constexpr int BN_NH = 6;
__host__ __device__ inline int bn_kpad(int K) { return ((K + 7) & ~7) + 320; }
__host__ __device__ inline size_t bn_scratch() { return (rms_scratch_bytes(BN_NH) + 255) & ~(size_t)255; }
// XT false: the same norm with plain row-major bf16 output [row][K] -- the prefill's rows (rmsnorm_rows_kernel walks the serial sum with
// one wave per row, 24 us per launch at K = 4096 whatever the row count; this evaluation takes 9).
template <bool XT>
__global__ __launch_bounds__((1 + BN_NH) * 64) void batch_rmsnorm_xt_kernel(const uint16_t* x, const uint16_t* norm_w, float eps, uint16_t* xt, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NH = BN_NH, NS = 1 + NH, CW = 3;
    char* scratch = smem;
    float* xs = (float*)(smem + bn_scratch());
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = blockIdx.x, kpad = bn_kpad(K);
    GemvParams p{}; p.K = K; p.norm_w = norm_w; p.eps = eps; p.dbg = nullptr;
    const uint16_t* xrow = x + (size_t)s * K;
    long long t_aux = 0;
    uint4 xv[XCh<true>::value], nv[XCh<true>::value];
    if (wave != CW) {
        const int hw = wave < CW ? wave : wave - 1;
        x_issue<true, NS>(p, xrow, 1 + hw, lane, xv, nv);
        x_store<true, NS>(p, xs, kpad, 1 + hw, lane, xv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();                        // B1: the squares are in LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        rms_fold<rms_nf(NH)>(p, xs, scratch, hw < rms_nf(NH) ? hw : -1, lane, t_aux);       // X1, X2 inside
        __builtin_amdgcn_s_barrier();                        // B2: r published
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        x_normalize<NS>(p, xs, kpad, xs[kpad], 1 + hw, lane, xv, nv);
    } else {
        x_issue<true, NS>(p, xrow, 0, lane, xv, nv);
        x_store<true, NS>(p, xs, kpad, 0, lane, xv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();                        // B1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float r = rms_scale_wide<rms_nf(NH)>(p, xs, scratch, lane, t_aux);     // X1, X2 inside
        if (lane == 0) xs[kpad] = r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();                        // B2
        x_normalize<NS>(p, xs, kpad, r, 0, lane, xv, nv);
    }
    __syncthreads();                                         // B3: xs holds trunc(trunc(x*r)*w) as f32 (bf16-exact)
    if constexpr (!XT) {
        for (int u = threadIdx.x; u < (K >> 3); u += (1 + NH) * 64) {
            const float* v = xs + 8 * u;
            *(uint4*)(xt + (size_t)s * K + 8 * u) = make_uint4((uint32_t)bf_trunc(v[0]) | ((uint32_t)bf_trunc(v[1]) << 16), (uint32_t)bf_trunc(v[2]) | ((uint32_t)bf_trunc(v[3]) << 16),
                                                               (uint32_t)bf_trunc(v[4]) | ((uint32_t)bf_trunc(v[5]) << 16), (uint32_t)bf_trunc(v[6]) | ((uint32_t)bf_trunc(v[7]) << 16));
        }
        return;
    }
    // 16 B units of the B-operand layout: unit (C, m, kk) of sequence s = the eight k = 128C + 16e + 4m + kk, e = 0..7
    for (int u = threadIdx.x; u < (K >> 3); u += (1 + NH) * 64) {
        const int C = u >> 4, m = (u >> 2) & 3, kk = u & 3, kb = 128 * C + 4 * m + kk;
        uint32_t w4[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) w4[e >> 1] = (uint32_t)bf_trunc(xs[kb + 16 * e]) | ((uint32_t)bf_trunc(xs[kb + 16 * (e + 1)]) << 16);
        *(uint4*)(xt + xt_group(s, K) + xt_index(s & 15, kb)) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
}

// Fwd_Get_Rows (operations_impl.go:142-173) for the batch: row s = embedding of sequence s's current token (its context's device word)
__global__ void batch_embed_kernel(const uint16_t* emb, const BatchTab* tab, uint16_t* x, int dim, int vocab, int* err) {
    const int s = blockIdx.x;
    const int t = *tab->dtok[s];
    if (t < 0 || t >= vocab) { if (threadIdx.x == 0) atomicExch(err, 1 + s); return; }
    const uint4* src = (const uint4*)(emb + (size_t)t * dim);
    uint4* dst = (uint4*)(x + (size_t)s * dim);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) dst[i] = src[i];
}
// start of a batched run: every sequence's token word (tokens == nullptr: keep what its context holds) and position (its own context's
// StepState); ring = the batch's contiguous token words (the pipeline's token exchange), kept equal to the contexts' words
// pos[s] < 0: sequence s has ENDED (a stop id in an earlier chunk of the run): it stays frozen -- token word, position and caches as they are, no
// tokens logged -- while its column goes on recomputing its last step (the pass over the weights is shared)
__global__ void batch_set_state_kernel(const BatchTab* tab, const int32_t* tokens, const int32_t* pos, int32_t* ring, int honour_stop) {
    const int s = threadIdx.x;
    if (s >= tab->n) return;
    const bool frozen = pos[s] < 0;
    if (tokens && !frozen) *tab->dtok[s] = tokens[s];
    if (ring) ring[s] = *tab->dtok[s];
    StepState* st = tab->st[s];
    if (!frozen) st->pos = pos[s];
    st->n_out = 0; st->finished = frozen ? 1 : 0; st->honour_stop = honour_stop;
}
// pipeline, first stage: the tokens the last stage sent (contiguous) -> every sequence's own token word
__global__ void batch_scatter_ring_kernel(const BatchTab* tab, const int32_t* ring) {
    const int s = threadIdx.x;
    if (s < tab->n) *tab->dtok[s] = ring[s];
}
// pipeline, a stage without the head: the step is done, every sequence moves on by one position (the last stage's argmax does it there)
__global__ void batch_advance_kernel(const BatchTab* tab) {
    const int s = threadIdx.x;
    if (s < tab->n) tab->st[s]->pos = tab->st[s]->pos + 1;
}

// ------------------------------------------------------------------------------------------------
// gemm_stream_kernel<EPI, NCH, NTW>: the prefill product (S >= 16 rows of ONE sequence) on the matrix-core feed of mfma_stream_kernel.
// gemm_mfma_kernel (lnb_kernels.hip) stages BOTH operand tiles through the LDS (62-66 % of the f32 matrix rate at S >= 2048, 38 % at
// S = 128: 30 ms for the 128-row prompt); here the weights never touch the LDS -- M16 units HBM -> VGPR -> A operand, one unpack op per
// matrix instruction and chain, shared by all the batch tiles of the wave -- and the activations are staged once per 128-step chunk and
// workgroup as f32 rows (pitch 130 floats: the 16 rows x 2 k of a 32-lane read group hit 32 different banks), so a B operand is one
// 4-byte LDS read, no unpack.  A wave owns NCH chains of one 16-row weight tile x NTW batch tiles of 16 rows (NCH * NTW independent
// accumulators: the matrix instructions of a k-group issue back to back at the pipe's 32 cycles, the LDS reads and the unpack op ride
// in between -- the single chain's forwarding cliff, 5.11, does not exist here); the four waves of a workgroup take four neighbouring
// weight tiles and share the 16 * NTW activation rows.  One barrier per chunk (double-buffered tile).
// Staging map: lane l of wave w loads, for every batch tile u, the 16 bytes (row 16u + (l & 15), k-unit 4w + (l >> 4)): a 16-lane write
// group of ds_write_b64 is then 16 ROWS of one k-unit = bank pairs 2 * row: conflict-free (the row-major map -- 16 k-units of one row per
// write group -- was 4-way conflicted: tools/gemmstream_bench.hip, staging alone cost 10-12 % of the kernel).
// NTW <= 4: two workgroups per CU (LDS 2 x 33 KB, <= 256 registers): one's staging / barrier hides under the other's matrix instructions.
// grid (persistent workgroups over the weight tiles, ceil(S / (16 * NTW))), block 256, dynamic LDS = 2 * 16 * NTW * 130 * 4 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int GS_PITCH = 130;
#ifndef GS_DBG
#define GS_DBG 0                                             // tools/gemmstream_bench.hip: 1 = stage only the first two chunks, 2 = no barriers, 4 = LDS operands unused,
                                                             // 8 = activation loads always hit one line, 16 = weight loads always re-read chunk 0, 32 = no unpack, 64 = no LDS reads issued
#endif
template <int I, int N, class F> DEVINL void static_for(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); } }
template <int N, int L> DEVINL void wait_slot(u32x4 (&b)[L]) {           // any number of loads per slot: one counted wait, then every register of the slot is handed back
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");            // (volatile asm statements keep their order; every use of b[i] hangs off its own retire)
#pragma unroll
    for (int i = 0; i < L; i++) asm volatile("; RING_RETIRE %0" : "+v"(b[i]));
}
DEVINL void ld_plain(u32x4& d, const void* a) { asm volatile("global_load_dwordx4 %0, %1, off ; RING_LOAD" : "=&v"(d) : "v"(a) : "memory"); }
#ifndef GS_W_NT
#define GS_W_NT 0                                            // weight loads: default cache policy (0) -- the row groups of a launch share every weight tile through the L2:
                                                             // 1-5 % faster than non-temporal loads (1) at 128-4096 rows, 3.6 against 4.7 GB of fabric reads per gate|up launch at 4096
#endif
#ifndef GS_X1
#define GS_X1 1                                              // chain layouts at four batch tiles per wave: the one-k matrix instruction with CBSZ / ABID (no transpose); 0 = the row-swap transpose
#endif
#ifndef GS_RX
#define GS_RX 2                                              // ring depth of the two-chain product on the chain layouts at four batch tiles per wave
#endif
#ifndef GS_R1
#define GS_R1 3                                              // ring depth at NTW = 1 / NTW = 2 (one chain)
#define GS_R2 3
#endif
#ifndef GS_RT
#define GS_RT 2                                              // ring depth with two weight tiles per wave (TT): a chunk is 256 matrix instructions of lookahead
#endif
#ifndef GS_OCC1
#define GS_OCC1 2                                            // waves per SIMD the register budget is set for at NTW = 1 / NTW = 2 (tools/gemmstream_bench.hip sweeps them)
#define GS_OCC2 2
#endif
// SRC: where the A operand comes from.  0 = the M16 copy (lnb_model_enable_batch).  Round 5 -- the RESIDENT layouts, no second copy:
//   1 = the row-broadcast layout of wo / w2 (tag RW 4): its 16-byte units ARE M16 units (the eight k of one row with the same k % 16), only their order
//       in memory differs -- lane (i, kk) finds unit (m) of chunk C at ((4 tile + i / 4) * K / 128 + C) * 1024 + ((i % 4) * 16 + 4 m + kk) * 16: same loads,
//       same unpack, different address arithmetic;
//   2 = the chain layouts [N/RW][K/8][NCH][RW][8] (wq|wk|wv, w1|w3, output): a unit holds EIGHT CONSECUTIVE k of one row.  The four lanes that feed one
//       row of the A operand -- (i, kk = 0..3) = lanes i, i + 16, i + 32, i + 48: one in each 16-lane row of the wave -- load the four units of 32
//       consecutive k (load j of a chunk: k = 128 C + 32 j + 8 kk ...), and a 4 x 4 transpose over the wave's ROWS hands every lane the element it owns of
//       each of the eight k-groups: v_permlane16_swap_b32 + v_permlane32_swap_b32, the row swaps new in gfx950 (tools/permlane_probe.hip prints what they
//       do), 12 of them + 8 v_perm_b32 half-selects per 8 matrix instructions and chain instead of 8 shifts -- shared by the NTW batch tiles of the wave,
//       hidden behind the matrix pipe.  (A DPP quad is four ADJACENT lanes, i.e. four different rows of the operand: quad_perm cannot do this.)
// The k-groups are consumed in the same ascending order in all three, so the chains -- and the bits -- are the same (tests/test_gpu_batch.py).
DEVINL void ct_ops(float (&op)[8], const u32x4& v, unsigned sel) {      // chain layouts: the eight operands (k-groups g' = 0..7 of a load's 32 k) of this lane
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // source row r (the lane kk = r of the quartet) holds k = 8 r + 4 h + {0, 1} in P and + {2, 3} in Q; k-group g' = 2 r + h wants, in target row t,
        // element t of those four: rows 0, 1 take P's halves, rows 2, 3 take Q's
        const unsigned P = v[2 * h], Q = v[2 * h + 1];
        const auto pp = __builtin_amdgcn_permlane16_swap(P, P, false, false);              // [P0 P0 P2 P2], [P1 P1 P3 P3]   (rows of the wave)
        const auto qq = __builtin_amdgcn_permlane16_swap(Q, Q, false, false);
        const auto a = __builtin_amdgcn_permlane32_swap(pp[0], qq[0], false, false);       // [P0 P0 Q0 Q0], [P2 P2 Q2 Q2]
        const auto b = __builtin_amdgcn_permlane32_swap(pp[1], qq[1], false, false);       // [P1 P1 Q1 Q1], [P3 P3 Q3 Q3]
        op[0 + h] = __uint_as_float(__builtin_amdgcn_perm(0u, a[0], sel)); op[2 + h] = __uint_as_float(__builtin_amdgcn_perm(0u, b[0], sel));
        op[4 + h] = __uint_as_float(__builtin_amdgcn_perm(0u, a[1], sel)); op[6 + h] = __uint_as_float(__builtin_amdgcn_perm(0u, b[1], sel));
    }
}
// a byte offset that IS wave-uniform, told to the compiler (its divergence analysis loses track of the issue cursor through the lambdas: the "s" operand
// of the asm loads would be handed a VGPR pair)
DEVINL size_t uniform_off(size_t off) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)off), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(off >> 32));
    return ((size_t)hi << 32) | lo;
}
DEVINL void ld_w_plain(u32x4& d, unsigned voff, const char* sb) { asm volatile("global_load_dwordx4 %0, %1, %2 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory"); }
template <int M> DEVINL void ld_rc_unit(u32x4& d, unsigned voff, const char* sb) {     // row-broadcast layout: unit m of the lane's row and chunk sits 64 m bytes further
    static_assert(M >= 0 && M < 4, "unit");
    if constexpr (M == 0) asm volatile("global_load_dwordx4 %0, %1, %2 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:64 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:128 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
    if constexpr (M == 3) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory");
}
template <int EPI, int NCH, int NTW, int SRC = 0, int TT = 0>
__global__ __launch_bounds__(256, NTW == 1 ? GS_OCC1 : NTW == 2 ? GS_OCC2 : NTW == 4 ? 2 : 1) void gemm_stream_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // TT (round 6): a wave of a ONE-chain product (wo, w2: M16 copy or row-broadcast layout) carries TWO neighbouring weight tiles x NTW batch tiles, the shape the gate|up
    // product has by construction: every B operand read from the LDS feeds two matrix instructions instead of one, staging and the chunk barrier are paid once per 256 instead of
    // 128 of them.  The unit a wave owns is then a tile PAIR (2 u, 2 u + 1); an odd last tile is computed twice and stored once.
    static_assert(!TT || (NCH == 1 && SRC != 2 && NTW == 4), "two tiles per wave: one-chain products from the M16 copy or the row-broadcast layout, four batch tiles");
    constexpr int NC = TT ? 2 : NCH;                         // chains (accumulator sets) per wave
    constexpr int L = NC * 4 + NTW;                          // loads per chunk and lane: NC * 4 weight units + NTW activation units
    constexpr int R = TT ? GS_RT : NCH == 1 ? (NTW == 1 ? GS_R1 : NTW == 2 ? GS_R2 : 3) : (SRC == 2 && NTW == 4) ? GS_RX : 3;   // chunks in flight (two chains x four batch tiles from a chain
                                                             // layout: 2 -- a chunk is then 256 matrix instructions = 3.6 us of lookahead each, and the third slot's 48
                                                             // registers spilled).  A chunk of one batch tile is 32 matrix instructions = ~1.5k
                                                             // cycles of a wave: three of them in flight are less than HBM's latency under load
    static_assert(R * L <= 60, "vmcnt is a 6-bit counter");
    constexpr int D = NTW == 1 ? 3 : NTW == 2 ? 2 : 1;       // B-operand reads run this many steps (of four k-groups) ahead of their matrix instructions
    static_assert(D * 2 * NTW <= 15, "lgkmcnt is a 4-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int rows_wg = 16 * NTW;                        // batch rows of this workgroup
    // dispatch order -> (weight-tile group bx, row group by): tile groups fastest, or row groups fastest (launch_gemm_stream decides, lnb_kernels.hip)
    const int lin_wg = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int by = p.rows_fastest ? lin_wg % (int)gridDim.y : (int)blockIdx.y, bx = p.rows_fastest ? lin_wg / (int)gridDim.y : (int)blockIdx.x;
    const int m0 = by * rows_wg;
    const int nchunks = p.K >> 7, n_tiles = (p.n_rows + 15) >> 4, n_units = TT ? (n_tiles + 1) >> 1 : n_tiles;       // units: what a wave owns -- a tile, or a tile pair
    const int rounds = (n_units + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);
    const int T = rounds * nchunks;
    const size_t chain_bytes = (size_t)nchunks * 4096;
    const unsigned aoff = (unsigned)(((lane & 15) * 4 + (lane >> 4)) * 16);
    // resident layouts: per-lane byte offsets of this lane's (row i = lane & 15, kk = lane >> 4)
    const unsigned rc_voff = (unsigned)(((lane & 15) >> 2) * nchunks * 1024 + (((lane & 15) & 3) * 16 + (lane >> 4)) * 16);
    const int ct_rw = p.rw > 0 ? p.rw : 16;                   // (chain layouts only)
    const unsigned ct_s8 = (unsigned)(NCH * ct_rw * 16);     // bytes from one 8-wide k chunk of a row block to the next
    const unsigned ct_sel = ((lane >> 4) & 1) ? 0x03020c0cu : 0x01000c0cu;      // v_perm_b32: the low (kk even) / high (kk odd) half of a dword, widened to f32
    unsigned ct_voff = 0; size_t ct_base = 0; int ct_tile = -1;
    float* Bs = (float*)smem;                                // [2][rows_wg][GS_PITCH]
    constexpr size_t bs_stride = (size_t)rows_wg * GS_PITCH;
    const int srow = lane & 15, scol = wave * 4 + (lane >> 4);                     // staging: row inside a batch tile, 16-byte k-unit of the chunk
    const uint16_t* xrow[NTW];
#pragma unroll
    for (int u = 0; u < NTW; u++) { int row = m0 + u * 16 + srow; row = row < p.S ? row : p.S - 1; xrow[u] = p.x + (size_t)row * p.K + scol * 8; }
    u32x4 buf[R][L];
    int ir = 0, ic = 0, issued = 0;                          // issue cursor: (round, chunk)
    auto tile_of = [&](int round) { int t = (round * (int)gridDim.x + bx) * 4 + wave; return t < n_units ? t : n_units - 1; };
    auto tile_c = [&](int unit, int c) { int t = TT ? unit * 2 + c : unit; return t < n_tiles ? t : n_tiles - 1; };        // chain c's weight tile
    auto issue_next = [&](u32x4 (&dst)[L]) {
        const int tile = tile_of(ir);
        if constexpr (SRC == 2) {
            if (tile != ct_tile) {                           // (wave-uniform; once per weight tile) the lane's row: block b and row r inside it; b0 = the tile's first block
                int n = tile * 16 + (lane & 15); n = n < p.n_rows ? n : p.n_rows - 1;
                const int b0 = (tile * 16) / ct_rw, b = n / ct_rw, r = n - b * ct_rw;
                ct_voff = (unsigned)(b - b0) * (unsigned)(p.K >> 3) * ct_s8 + (unsigned)r * 16u + (unsigned)(lane >> 4) * ct_s8;
                ct_base = (size_t)b0 * (size_t)(p.K >> 3) * ct_s8;
                ct_tile = tile;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if constexpr (SRC == 1) {
                const char* wb = (const char*)p.w + uniform_off(((size_t)tile_c(tile, c) * 4 * nchunks + (size_t)ic) * 1024);
                ld_rc_unit<0>(dst[c * 4 + 0], rc_voff, wb); ld_rc_unit<1>(dst[c * 4 + 1], rc_voff, wb);
                ld_rc_unit<2>(dst[c * 4 + 2], rc_voff, wb); ld_rc_unit<3>(dst[c * 4 + 3], rc_voff, wb);
                continue;
            }
            if constexpr (SRC == 2) {
                const size_t o0 = uniform_off(ct_base + (size_t)c * (size_t)(ct_rw * 16) + (size_t)(16 * ic) * ct_s8), oj = uniform_off((size_t)4 * ct_s8);
#pragma unroll
                for (int jj = 0; jj < 4; jj++) ld_w_plain(dst[c * 4 + jj], ct_voff, (const char*)p.w + o0 + (size_t)jj * oj);
                continue;
            }
            const char* wb = (const char*)p.w16 + (size_t)(TT ? tile_c(tile, c) : tile * NCH + c) * chain_bytes + ((GS_DBG & 16) ? 0 : (size_t)ic * 4096);
            if constexpr (GS_W_NT) {
                ld_unit_nt<0>(dst[c * 4 + 0], aoff, wb); ld_unit_nt<1>(dst[c * 4 + 1], aoff, wb);
                ld_unit_nt<2>(dst[c * 4 + 2], aoff, wb); ld_unit_nt<3>(dst[c * 4 + 3], aoff, wb);
            } else {
                ld_unit<0>(dst[c * 4 + 0], aoff, wb); ld_unit<1>(dst[c * 4 + 1], aoff, wb);
                ld_unit<2>(dst[c * 4 + 2], aoff, wb); ld_unit<3>(dst[c * 4 + 3], aoff, wb);
            }
        }
#pragma unroll
        for (int u = 0; u < NTW; u++) ld_plain(dst[NC * 4 + u], (GS_DBG & 8) ? p.x : xrow[u] + (size_t)ic * 128);
        if (issued + 1 < T) { issued++; if (++ic == nchunks) { ic = 0; ir++; } }
    };
#pragma unroll
    for (int j = 0; j < R; j++) issue_next(buf[j]);
    f32x4 acc[NC][NTW];
#pragma unroll
    for (int cc = 0; cc < NC; cc++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[cc][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // X1 (chain layouts, four batch tiles per wave): the four BLOCKS of v_mfma_f32_16x16x1_f32 are the four batch tiles, one k per instruction, and CBSZ = 2
    // broadcasts the A operand of block ABID to all of them -- lane (i, r) holds k = 32 j + 8 r + e of weight row i (exactly what the chain-layout loads put
    // there), so ABID = r walks the k of a load in ascending order with NO transpose: one unpack op per element, shared by the four ABID steps that use the
    // register.  Bit-identical to the chain and at the 16x16x4 instruction's rate (tools/mfma_x1_probe.hip, profiles/r05_mfma_x1_probe.log).
    constexpr bool X1 = GS_X1 && SRC == 2 && NTW == 4;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc16[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; cc++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc16[cc][q] = 0.f;
    int c = 0, round = 0;
    const int bcol = lane & 15, bk = lane >> 4;
    for (int t0 = 0; t0 < T; t0 += R) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (t0 + j < T) {
                wait_slot<(R - 1) * L, L>(buf[j]);
                float* bw = Bs + (size_t)((t0 + j) & 1) * bs_stride;
                if (!(GS_DBG & 1) || t0 + j < 2) {
#pragma unroll
                    for (int u = 0; u < NTW; u++) {          // widen once per element: bf16 -> f32 is a shift / a mask
                        const u32x4 v = buf[j][NC * 4 + u];
                        float* d = bw + (size_t)(u * 16 + srow) * GS_PITCH + scol * 8;
                        *(float2*)(d) = make_float2(bf_lo(v[0]), bf_hi(v[0])); *(float2*)(d + 2) = make_float2(bf_lo(v[1]), bf_hi(v[1]));
                        *(float2*)(d + 4) = make_float2(bf_lo(v[2]), bf_hi(v[2])); *(float2*)(d + 6) = make_float2(bf_lo(v[3]), bf_hi(v[3]));
                    }
                }
                if (!(GS_DBG & 2)) __syncthreads();          // the chunk's activations are in the LDS (the other buffer is still being read by slower waves)
                // B operands: hand-issued ds_read2_b32 (two k-groups of one batch tile each), D steps of four k-groups ahead of the matrix
                // instructions that consume them (left to hipcc, a read at NTW = 1 / 2 is issued right in front of its use: ~100 cycles of LDS
                // latency per two matrix instructions, 55 % of the matrix rate at 128 rows); counted lgkmcnt waits, registers handed back
                // through the same RING markers tools/isa_audit.py checks
                if constexpr (X1) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const unsigned ba1 = (unsigned)(size_t)(bw + (size_t)lane * GS_PITCH);       // LDS byte address of this lane's activation row: batch tile lane >> 4, row lane & 15
                    f32x2 bq1[2][4];                         // the B operands of a group of eight steps (k = 8 g .. 8 g + 7 of the chunk), read one group ahead
                    auto lds_issue1 = [&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value;
#pragma unroll
                        for (int h = 0; h < 4; h++) {        // (a local result and an address EXPRESSION: hipcc does not see a captured variable named directly in an asm operand of a generic lambda)
                            f32x2 v;
                            asm volatile("ds_read_b64 %0, %1 offset:%2 ; RING_LOAD" : "=v"(v) : "v"(ba1 + 0u), "n"(32 * g + 8 * h) : "memory");
                            bq1[g & 1][h] = v;
                        }
                    };
                    lds_issue1(std::integral_constant<int, 0>{});
                    float av1[NCH][8];
                    static_for<0, 16>([&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value, jj = g >> 2, r = g & 3;       // load jj of the chunk, lane row r: k = 128 C + 32 jj + 8 r + e
                        if constexpr (g + 1 < 16) lds_issue1(std::integral_constant<int, g + 1>{});
                        if constexpr (r == 0) {
#pragma unroll
                            for (int cc = 0; cc < NCH; cc++)
#pragma unroll
                                for (int e = 0; e < 8; e++) { const unsigned w = buf[j][cc * 4 + jj][e >> 1]; av1[cc][e] = (e & 1) ? bf_hi(w) : bf_lo(w); }
                            if constexpr (jj == 3) {         // every weight register of the slot has been read (the x registers were consumed above): refill
#pragma unroll
                                for (int cc = 0; cc < NCH; cc++)
#pragma unroll
                                    for (int e = 0; e < 8; e++) asm volatile("" : "+v"(av1[cc][e]));
                                issue_next(buf[j]);
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(g + 1 < 16 ? 4 : 0) : "memory");
#pragma unroll
                        for (int h = 0; h < 4; h++) asm volatile("; RING_RETIRE %0" : "+v"(bq1[g & 1][h]));
#pragma unroll
                        for (int e = 0; e < 8; e++)           // ascending k (operations_lineartransform.go:46-65)
#pragma unroll
                            for (int cc = 0; cc < NCH; cc++)
                                acc16[cc] = __builtin_amdgcn_mfma_f32_16x16x1f32(av1[cc][e], bq1[g & 1][e >> 1][e & 1], acc16[cc], 2, r, 0);
                    });
                    if (++c == nchunks) {                    // block t of the result = batch tile t: column lane & 15 of it, rows (lane >> 4) * 4 + q of the weight tile
                        const int tile = (round * (int)gridDim.x + bx) * 4 + wave;
#pragma unroll
                        for (int t = 0; t < NTW; t++) {
                            const f32x4 a0 = {acc16[0][4 * t], acc16[0][4 * t + 1], acc16[0][4 * t + 2], acc16[0][4 * t + 3]};
                            const f32x4 a1 = {acc16[NCH - 1][4 * t], acc16[NCH - 1][4 * t + 1], acc16[NCH - 1][4 * t + 2], acc16[NCH - 1][4 * t + 3]};
                            if (tile < n_tiles) gemm_epilogue4<EPI>(p, a0, a1, m0 + t * 16 + (lane & 15), tile * 16 + (lane >> 4) * 4);
                        }
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++)
#pragma unroll
                            for (int q = 0; q < 16; q++) acc16[cc][q] = 0.f;
                        c = 0; round++;
                    }
                    continue;
                }
                const unsigned ba = (unsigned)(size_t)(bw + (size_t)bcol * GS_PITCH + bk);       // LDS byte address of this lane's (row, kk) at k-group 0, batch tile 0
                float bq[D + 1][2 * NTW][2];
                auto lds_issue = [&](auto ec) __attribute__((always_inline)) {
                    constexpr int e = decltype(ec)::value;
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int t = 0; t < NTW; t++) {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            f32x2 v;
                            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3 ; RING_LOAD" : "=v"(v) : "v"(ba + (unsigned)(t * 16 * GS_PITCH * 4)), "n"(16 * e + 8 * h), "n"(16 * e + 8 * h + 4) : "memory");
                            bq[e % (D + 1)][h * NTW + t][0] = v[0]; bq[e % (D + 1)][h * NTW + t][1] = v[1];
                        }
                };
                if constexpr (!(GS_DBG & 64)) static_for<0, D>(lds_issue);
                float opq[NC][8];                           // (chain layouts) the operands of the current load's eight k-groups
                static_for<0, 8>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int e = decltype(ec)::value;
                    if constexpr (e + D < 8 && !(GS_DBG & 64)) lds_issue(std::integral_constant<int, e + D>{});
                    float av[NC][4];
                    if constexpr (SRC == 2 && (e & 1) == 0) {      // chain layouts: load e / 2 of the chunk carries the k-groups 4 e .. 4 e + 7; all eight operands at once
#pragma unroll
                        for (int cc = 0; cc < NC; cc++) ct_ops(opq[cc], buf[j][cc * 4 + (e >> 1)], ct_sel);
                    }
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int cc = 0; cc < NC; cc++) {
                            if constexpr (SRC == 2) av[cc][m] = opq[cc][4 * (e & 1) + m];
                            else av[cc][m] = (GS_DBG & 32) ? __uint_as_float(buf[j][cc * 4 + m][e >> 1]) : unit_elem(buf[j][cc * 4 + m], e);
                        }
                    if constexpr (e == 7) {                  // every weight register of the slot has been read (the x registers were consumed above): refill
#pragma unroll
                        for (int m = 0; m < 4; m++)
#pragma unroll
                            for (int cc = 0; cc < NC; cc++) asm volatile("" : "+v"(av[cc][m]));
                        issue_next(buf[j]);
                    }
                    constexpr int ahead = (e + D < 8 ? D : 7 - e) * 2 * NTW;             // ds_read2 instructions issued after the ones of step e
                    if constexpr (!(GS_DBG & 64)) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(ahead) : "memory");
#pragma unroll
                    for (int q = 0; q < ((GS_DBG & 64) ? 0 : 2 * NTW); q++) { asm volatile("; RING_RETIRE %0" : "+v"(bq[e % (D + 1)][q][0])); asm volatile("; RING_RETIRE %0" : "+v"(bq[e % (D + 1)][q][1])); }
#pragma unroll
                    for (int m = 0; m < 4; m++) {            // k-group g = 4e + m: k = 128C + 4g + kk, ascending (operations_lineartransform.go:46-65)
#pragma unroll
                        for (int t = 0; t < NTW; t++) {
                            const float b = (GS_DBG & 4) ? __int_as_float(0x3f800000 + lane + m + t) : bq[e % (D + 1)][(m >> 1) * NTW + t][m & 1];
#pragma unroll
                            for (int cc = 0; cc < NC; cc++) {
                                if (e == 0 && m == 0 && c == 0) acc[cc][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][m], b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                                else acc[cc][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][m], b, acc[cc][t], 0, 0, 0);
                            }
                        }
                    }
                });
                if (++c == nchunks) {                        // D layout: column lane & 15 of the batch tile, rows (lane >> 4) * 4 + r of the weight tile
                    const int tile = (round * (int)gridDim.x + bx) * 4 + wave;
                    if constexpr (TT) {                      // two tiles, each with the one-chain epilogue
#pragma unroll
                        for (int cc = 0; cc < 2; cc++) {
                            const int tt = tile * 2 + cc;
                            if (tile < n_units && tt < n_tiles) {
#pragma unroll
                                for (int t = 0; t < NTW; t++) gemm_epilogue4<EPI>(p, acc[cc][t], acc[cc][t], m0 + t * 16 + (lane & 15), tt * 16 + (lane >> 4) * 4);
                            }
                        }
                    } else if (tile < n_tiles) {
#pragma unroll
                        for (int t = 0; t < NTW; t++)
                            gemm_epilogue4<EPI>(p, acc[0][t], acc[NCH - 1][t], m0 + t * 16 + (lane & 15), tile * 16 + (lane >> 4) * 4);
                    }
                    c = 0; round++;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// gemm_blgp_kernel<EPI, NCH> (round 6): the prefill / batch product for FEW batch rows straight from the CHAIN layouts, no transpose and no second copy.
//
// At one or two batch tiles per wave gemm_stream_kernel's chain-layout source (SRC 2) turns every load of eight consecutive k into 16x16x4 A operands with
// 12 row swaps + 8 half-selects per 8 matrix instructions (wq|wk|wv of a 128-token prompt: 91 us against 67 from the M16 copy; batches of up to 32 sequences want
// the 15 GB copy for that reason).  v_mfma_f32_16x16x1_f32 computes FOUR 16x16 blocks with one k per instruction, and BLGP = 4 + r broadcasts the B operand of lane row r
// to all four (tools/mfma_x1_probe.hip: bit-identical to the sequential chain, 32 cycles per instruction even as one dependent chain).  With the four blocks = four
// (weight tile, chain) pairs -- lane (i, b) = row i of block b -- every lane consumes exactly what a chain-layout unit holds, eight consecutive k of ITS row, in order:
// one shift / mask per element, nothing crosses lanes.  The B operand of steps k0 + 4m + r is lane (n, r)'s activation x[n][k0 + 4m + r]: one ds_read2_b32 per eight steps.
//   NCH = 1: a wave owns 4 weight tiles x 16 batch rows;  NCH = 2 (gate|up): 2 tiles x (gate, up) -- block 2t = gate, 2t + 1 = up of tile t, so SiLU*up stays lane-local.
// The four waves of a workgroup take four neighbouring tile groups and share the batch tile's activations (staged per 128-step chunk as f32 rows, pitch GS_PITCH:
// gemm_stream_kernel's conflict-free map); weights through the hand-counted register ring (two chunks of 16 units in flight).  k ascends inside a unit, across the
// units of a chunk and across chunks: the reference's chain (operations_lineartransform.go:46-65), the same bits as every other form (tests/test_gpu_round6.py).
// NOT the default (launch_gemm_stream, lnb_kernels.hip: measured slower than the transpose -- one dependent chain per wave, a quarter as many waves); opt-in LNB_GEMM_BLGP=1|2.
// grid (ceil(tile groups / 4), batch tiles), block 256, dynamic LDS 2 * 16 * GS_PITCH * 4.
// ------------------------------------------------------------------------------------------------
template <int EPI, int NCH>
__global__ __launch_bounds__(256, 2) void gemm_blgp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int L = 17, R = 2;                             // loads per chunk and lane: 16 weight units + 1 activation unit; chunks in flight
    constexpr int TPG = 4 / NCH;                             // weight tiles per wave
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4, n_tg = (n_tiles + TPG - 1) / TPG, nchunks = p.K >> 7;
    const int tg_raw = (int)blockIdx.x * 4 + wave, tg = tg_raw < n_tg ? tg_raw : n_tg - 1;       // (a wave past the end re-walks the last group and stores nothing: it still stages and meets the barriers)
    const int m0 = (int)blockIdx.y * 16;
    // this lane's weight row: block q of the wave
    const int my_tile = tg * TPG + (NCH == 1 ? q : (q >> 1)), chain = NCH == 1 ? 0 : (q & 1);
    int n = my_tile * 16 + i; n = n < p.n_rows ? n : p.n_rows - 1;
    const int rw = p.rw, b0 = (tg * TPG * 16) / rw, b = n / rw, r = n - b * rw;
    const unsigned s8 = (unsigned)(NCH * rw * 16);           // bytes from one 8-wide k unit of a row block to the next
    const unsigned voff = (unsigned)(b - b0) * (unsigned)(p.K >> 3) * s8 + (unsigned)chain * (unsigned)(rw * 16) + (unsigned)r * 16u;
    const size_t wbase = (size_t)b0 * (size_t)(p.K >> 3) * s8;
    float* Bs = (float*)smem;                                // [2][16][GS_PITCH]
    constexpr size_t bs_stride = (size_t)16 * GS_PITCH;
    const int srow = tid & 15, scol = tid >> 4;              // staging: row of the batch tile, 16-byte k-unit of the chunk
    int xr = m0 + srow; xr = xr < p.S ? xr : p.S - 1;
    const uint16_t* xrow = p.x + (size_t)xr * p.K + scol * 8;
    u32x4 buf[R][L];
    int ic = 0;
    auto issue_next = [&](u32x4 (&dst)[L]) {
        const char* cb = (const char*)p.w + uniform_off(wbase + (size_t)(16 * ic) * s8);
        const size_t su = uniform_off((size_t)s8);
#pragma unroll
        for (int u = 0; u < 16; u++) ld_w_plain(dst[u], voff, cb + (size_t)u * su);
        ld_plain(dst[16], xrow + (size_t)ic * 128);
        if (ic + 1 < nchunks) ic++;                          // past the last chunk: stays put, re-reads
    };
#pragma unroll
    for (int j = 0; j < R; j++) issue_next(buf[j]);
    f32x16 acc;
#pragma unroll
    for (int z = 0; z < 16; z++) acc[z] = 0.f;
    for (int t0 = 0; t0 < nchunks; t0 += R) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (t0 + j < nchunks) {
                wait_slot<(R - 1) * L, L>(buf[j]);
                float* bw = Bs + (size_t)((t0 + j) & 1) * bs_stride;
                {   // widen the chunk's activations once per element
                    const u32x4 v = buf[j][16];
                    float* d = bw + (size_t)srow * GS_PITCH + scol * 8;
                    *(float2*)(d) = make_float2(bf_lo(v[0]), bf_hi(v[0])); *(float2*)(d + 2) = make_float2(bf_lo(v[1]), bf_hi(v[1]));
                    *(float2*)(d + 4) = make_float2(bf_lo(v[2]), bf_hi(v[2])); *(float2*)(d + 6) = make_float2(bf_lo(v[3]), bf_hi(v[3]));
                }
                __syncthreads();                             // the chunk's activations are in the LDS (the other buffer may still be read by slower waves)
                const float* bl = bw + (size_t)i * GS_PITCH + q;     // this lane's B operands: x[row i][k0 + 4 m + q]
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const float b0v = bl[u * 8], b1v = bl[u * 8 + 4];
                    const u32x4 w = buf[j][u];
                    float a[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) a[e] = (e & 1) ? bf_hi(w[e >> 1]) : bf_lo(w[e >> 1]);
                    // k ascending: e = 0..7 (operations_lineartransform.go:46-65); step e takes its B operand from lane row e & 3
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[0], b0v, acc, 0, 0, 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[1], b0v, acc, 0, 0, 5);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[2], b0v, acc, 0, 0, 6);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[3], b0v, acc, 0, 0, 7);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[4], b1v, acc, 0, 0, 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[5], b1v, acc, 0, 0, 5);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[6], b1v, acc, 0, 0, 6);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[7], b1v, acc, 0, 0, 7);
                }
                asm volatile("" : "+v"(acc));                // every weight register of the slot has been read: refill
                issue_next(buf[j]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
    // D layout: register 4 * blk + rr of lane (n, q) = block blk, row 4 q + rr of its tile, batch column n
    if (tg_raw < n_tg) {
        const int m = m0 + i;
        if constexpr (NCH == 1) {
#pragma unroll
            for (int blk = 0; blk < 4; blk++) {
                const int tile = tg * 4 + blk;
                const f32x4 a0 = {acc[4 * blk], acc[4 * blk + 1], acc[4 * blk + 2], acc[4 * blk + 3]};
                if (tile < n_tiles) gemm_epilogue4<EPI>(p, a0, a0, m, tile * 16 + q * 4);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int tile = tg * 2 + t;
                const f32x4 g = {acc[8 * t], acc[8 * t + 1], acc[8 * t + 2], acc[8 * t + 3]}, uu = {acc[8 * t + 4], acc[8 * t + 5], acc[8 * t + 6], acc[8 * t + 7]};
                if (tile < n_tiles) gemm_epilogue4<EPI>(p, g, uu, m, tile * 16 + q * 4);
            }
        }
    }
}
