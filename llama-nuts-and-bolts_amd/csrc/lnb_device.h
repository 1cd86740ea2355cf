// lnb_device.h -- shared host/device definitions for the MI355X (gfx950) LlamaTransformer.Forward path.
//
// Numerics contract (reference: adalkiran/llama-nuts-and-bolts):
//   * every bf16 store is a TRUNCATION of the f32 bit pattern   (src/dtype/bfloat16.go:31-33)
//   * every matmul output is ONE f32 chain, k ascending          (src/ml/operations_lineartransform.go:46-65)
//   * softmax in f64 without max subtraction                     (src/ml/operations_impl.go:492-508)
// The kernels keep those chains intact: one lane owns one output element and walks k in order.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define LNB_HD __host__ __device__ __forceinline__
#else
#define LNB_HD inline
#endif

// ---- tiled weight layout ------------------------------------------------------------------------
// A logical [N,K] bf16 matrix in the reference's [out_features,in_features] row-major layout is stored
// in HBM as   [N/RW][K/8][NCH][RW][8]   so that the RW output rows owned by the RW lanes of one
// consumer wave are contiguous for every 8-wide k chunk: one 16-byte load per lane, 1 KiB (RW=64) of
// perfectly sequential stream per wave instruction, and each lane still sees its own row in k order.
// NCH = number of independent chains per lane (2 for the fused w1|w3 gate/up matrix).
struct TiledDesc {
    uint16_t* w;      // device pointer
    int n_rows;       // logical lane-rows (outputs per chain)
    int k;            // in_features
    int rw;           // rows per wave block: 16 / 32 / 64
    int nch;          // chains per lane: 1 / 2
    int n_blocks;     // ceil(n_rows / rw)
    uint16_t* w16;    // the same matrix in the M16 layout (below), once lnb_model_enable_batch has built it; else nullptr
};
//
// RW == 4 selects the ROW-BROADCAST layout of rowcast_kernel (thin matrices, K % 128 == 0, NCH == 1):
//   [N/4 wave tiles][K/128 chunks][row%4][k%16][(k%128)/16]  -- a wave streams 1 KiB per 128-step chunk, lane (q, j) of the
// wave holds, in k order, the eight weights of row 4t+q whose k is congruent to j modulo 16.
LNB_HD size_t tiled_index(int n, int k, int c, int K, int RW, int NCH) {
    if (RW == 4) return ((((size_t)(n >> 2) * (size_t)(K >> 7) + (size_t)(k >> 7)) * 64 + (size_t)((n & 3) * 16 + (k & 15))) << 3) + (size_t)((k & 127) >> 4);
    size_t b = (size_t)(n / RW), r = (size_t)(n % RW), kc = (size_t)(k >> 3), e = (size_t)(k & 7);
    return ((((b * (size_t)(K >> 3) + kc) * (size_t)NCH + (size_t)c) * (size_t)RW + r) << 3) + e;
}
LNB_HD size_t tiled_elems(int n_rows, int K, int RW, int NCH) {
    if (RW == 4) return (size_t)((n_rows + 15) / 16) * 16 * (size_t)K;       // whole 16-row workgroups
    return (size_t)((n_rows + RW - 1) / RW) * (size_t)RW * (size_t)NCH * (size_t)K;
}

// ---- synthetic weights (same integer-exact generator as oracle/lnb_oracle.c; spec in DESIGN.md) ----
LNB_HD uint64_t lnb_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
LNB_HD int32_t lnb_synth_isum(uint64_t base, uint64_t idx) {
    uint64_t a = lnb_splitmix64(base ^ idx);
    uint64_t b = lnb_splitmix64(a);
    int32_t s = 0;
    for (int i = 0; i < 4; i++) { s += (int32_t)((a >> (16 * i)) & 0xFFFF); s += (int32_t)((b >> (16 * i)) & 0xFFFF); }
    return s - 262140;
}

// per-decode-step device state: lets one captured hipGraph be replayed for every position
constexpr int LNB_MAX_STOP_IDS = 8;
struct StepState {
    int32_t pos;        // start position of the current call (tokens already in the KV cache)
    int32_t n_out;      // tokens appended to out_tokens so far
    int32_t finished;   // 1 once a generated token was one of the stop ids (inference.go:233-252): position, token word and token log are frozen from
                        // then on -- the remaining steps of an enqueued run recompute the same step into the same KV row and change nothing visible
    int32_t n_stop;     // stop ids in use (0: never finishes)
    int32_t stop[LNB_MAX_STOP_IDS];
    int32_t honour_stop; // set per call: 1 = the entry point reports how far the run got (lnb_decode_greedy_until, lnb_batch_decode_until, pipeline ticks of a
                        // one-stage pipe) and the stop ids are compared; 0 = lnb_decode_greedy / lnb_batch_decode, which promise n_steps tokens: the ids are ignored
};

enum { EPI_STORE = 0, EPI_QKV_ROPE = 1, EPI_RESID = 2, EPI_SILU_MUL = 3 };

struct GemvParams {
    const uint16_t* w;          // tiled weights
    const uint16_t* x;          // [S][K] bf16 activations (pre-norm when norm_w != nullptr)
    const uint16_t* norm_w;     // [K] bf16 RMSNorm weight, or nullptr (no fused norm)
    float eps;
    int K;                      // in_features
    int n_rows;                 // logical lane-rows
    int S;                      // rows in this call
    int n_blocks;               // row blocks of RW rows
    int n_wg;                   // workgroups per x row; workgroup w owns blocks w, w+n_wg, ...
    const StepState* st;        // start position lives on the device
    // EPI_STORE / EPI_RESID / EPI_SILU_MUL
    uint16_t* out;              // [S][n_rows] bf16
    const uint16_t* res;        // [S][n_rows] bf16 residual (EPI_RESID)
    const float* silu;          // f32[65536] SiLU table (EPI_SILU_MUL)
    // EPI_QKV_ROPE
    const float* cis;           // [rows][head_dim/2][2] f32
    uint16_t* q_out;            // [S][n_heads*head_dim]
    uint16_t* cache_k;          // [n_kv][head_dim/8][seq_len][8]  (position-contiguous per 8-dim chunk: coalesced score reads)
    uint16_t* cache_v;          // [seq_len][n_kv*head_dim]
    int seq_len;
    int q_dim, kv_dim, head_dim;
    long long* dbg;             // optional per-wave timing dump (nullptr in production)
    int dbg_full;               // with dbg: 1 = also time every barrier and ring wait (two s_memtime reads + an lgkmcnt(0) each: the launch runs ~15 % slower,
                                // LNB_GEMV_TIMING=1), 0 = only the phase stamps and the exit record (bench.py's measured model: the launch runs as in production)
    int* norm_fb;               // optional counter: rows whose norm sum left the branch-free item walk for the record walk (counted by workgroup 0)
    int lds_pad;                // host side only: extra dynamic LDS requested for the launch (co-residency experiments: forces one workgroup per CU)
    int prio;                   // measurement aid (tools/ffn_overlap.py): != 0 raises the wave priority of rowcast_kernel's (self-feeding) chain waves
    int sched;                  // host side only: 0 = latency forms (one stream owns the chip: every CU, eight or nine waves, up to 124 KB of LDS per
                                // workgroup), 1 = throughput forms of the same arithmetic (lnb_ctx_set_schedule: at most 57 KB of LDS per workgroup, so
                                // that a chain-bound launch of one context shares a CU with the HBM-bound gate|up launch of another)
};

// exact-order prefill GEMM on the f32 matrix cores (gemm_mfma_kernel): Y[m][n] for S >= 16 rows per call
struct GemmParams {
    const uint16_t* w;          // tiled weights (any layout of tiled_index)
    int rw, nch;                // layout tag of w
    const uint16_t* x;          // [S][K] bf16 (already normalised when the GEMV would fuse the norm)
    int K, n_rows, S;
    const StepState* st;
    uint16_t* out; const uint16_t* res; const float* silu;            // EPI_STORE / EPI_RESID / EPI_SILU_MUL
    const float* cis; uint16_t* q_out; uint16_t* cache_k; uint16_t* cache_v; int seq_len, q_dim, kv_dim, head_dim;   // EPI_QKV_ROPE
    // gemm_stream_kernel (lnb_batch_kernels.h): the same product with the weights streamed from their M16 copy straight into the A operand
    const uint16_t* w16;        // M16 copy of w (lnb_model_enable_batch), or nullptr
    // rows = the one new token of each SEQUENCE of a batch (lnb_batch_*, more than 16 sequences): EPI_QKV_ROPE takes row m's position, caches
    // and cache length from the batch tables instead of st / cache_k / cache_v / seq_len
    const struct BatchTab* btab; const struct BatchKV* bkv;
    uint16_t* out_xt;           // EPI_SILU_MUL of a 17..32-sequence batch: the result in the B-operand layout, column groups of 16 (xt_group), for mfma_pair_kernel; else nullptr
    int rows_fastest;           // dispatch order of gemm_stream_kernel's workgroups: 1 = row groups fastest (set by the launcher for many row groups), 0 = weight-tile groups fastest
};

struct AttnParams {
    const uint16_t* q;          // [S][H*hd]
    const uint16_t* cache_k;    // [KVH][hd/8][seq_len][8]
    const uint16_t* cache_v;    // [seq_len][KVH*hd]
    uint16_t* out;              // [S][H*hd]
    const StepState* st;
    int S, H, KVH, hd, seq_len;
    int lds_T;                  // positions the LDS arrays of attn_exact_kernel are sized for (min(seq_len, what fits 160 KB)); the call's T must not exceed it
    int host_T;                 // start_pos + S of this call as the HOST knows it (0: unknown -- a replayed graph); lets the launcher refuse a call beyond lds_T
    float divisor;              // wide(trunc(f32(sqrt(hd))))  (llamatransformer.go:464)
    long long* dbg;             // LNB_GEMV_TIMING: phase stamps of workgroup (0,0)
    int mfma;                   // S >= 16: 16-row tiles on the f32 matrix cores (attn_mfma_kernel), same bits
    const double* exp_tab;      // f64[65536]: exp(trunc(s / divisor)) of every raw bf16 score s (attn_mfma_kernel)
    // long-context decode (S == 1, attn_long_scores_kernel + attn_long_pv_kernel): scores over all CUs, PV per (head, 16-dim slice)
    int longctx;                // 1: the two-kernel form; 2: ONE launch (attn_one_kernel: the slice workgroups score their share and exchange in the launch); 3: the same
                                // with every workgroup taking the bounded poll's time-out path (it scores every block itself: tests)
    int force_zseq;             // 1: always take the sequential-Z path of attn_long_pv_kernel (tests)
    double* e_buf;              // [H][seq_len] f64: exp of every score of the current token
    double* z_part;             // [H][ceil(seq_len / 256)] f64: per-block tree sums of e (only an ESTIMATE of Z, see the kernel)
    int* zseq_count;            // counts workgroups that had to fall back to the sequential Z chain (diagnostics)
    unsigned* cnt;              // attn_one_kernel (longctx >= 2): [H] arrival counters of the in-launch exchange (never reset: a launch's generation is old / slices) + [H] = polls that timed out
    // batched decode (attn_exact_kernel, S = 1 per sequence): query row i belongs to sequence i of the batch -- its own position, caches and
    // cache length come from the tables; the output goes to out_xt in the B-operand layout of the wo product (lnb_batch_kernels.h)
    const struct BatchTab* btab; const struct BatchKV* bkv; uint16_t* out_xt;
    uint64_t* score_idx;         // round 6, prefill (attn_mfma3_kernel): [H][ceil(S/16)][sidx_jt][64 lanes] x 8 bytes -- the sixteen-bit exp-table indices of pass 1, read back by pass 2; nullptr: attn_mfma_kernel (scores twice)
    int sidx_jt;                // position tiles per (head, query tile) strip of score_idx
    int touch;                  // round 6, long-context decode: 1 = the scores launch touches its layer's V rows for the PV launch that follows, from a workgroup on the XCD whose L2 the PV workgroups read
    int head_major;             // batched dense grid: 1 = head-major dispatch order inside an XCD (the round-3 order), 0 = sequence-major (the heads of a KV head back to back)
};

// ---- batched exact decode: up to 16 independent sequences per pass over the weights (lnb_batch_kernels.h) --------------------------
// v_mfma_f32_16x16x4_f32 IS the reference's k-ordered chain (NOTES.md 5.6); its 16 batch columns carry 16 SEQUENCES' decode tokens,
// so one pass over the weights serves all of them, each with its own bit-exact chains.  The matrix cores are fed straight from HBM:
// M16 weight layout of a logical [N, K] matrix (K % 128 == 0), NCH chains per 16-row tile:
//   [tile t = n / 16][chain c][chunk C = k / 128][m = (k % 16) / 4][i = n % 16][kk = k % 4][e = (k % 128) / 16]      (bf16)
// -- the element mapping of the row-broadcast layout above (a 16 B unit = the eight k of one row with the same k % 16), in 16-row
// tiles: a wave-wide 16 B-per-lane load of unit (C, m) is 1 KiB contiguous and matrix-core lane (i, kk) finds in it, as elements
// e = 0..7, its A operands of the k-groups g = 4e + m of the chunk (k = 128C + 4g + kk).  Activations of the batch ("xt"):
//   [C][m][kk][s = sequence 0..15][e]   -- lane (s, kk) loads its B operands of the same k-groups with the same instruction shape.
constexpr int LNB_BATCH_MAX = 128;          // sequences per batch
constexpr int LNB_STREAM_COLS = 16;         // ... of which mfma_stream_kernel carries up to 16 as the columns of ONE matrix instruction; larger batches are rows of gemm_stream_kernel
LNB_HD size_t m16_index(int n, int k, int c, int K, int NCH) {
    const int t = n >> 4, i = n & 15, C = k >> 7, e = (k >> 4) & 7, m = (k >> 2) & 3, kk = k & 3;
    return ((((((size_t)t * NCH + c) * (size_t)(K >> 7) + C) * 4 + m) * 16 + i) * 4 + kk) * 8 + e;
}
LNB_HD size_t m16_elems(int n_rows, int K, int NCH) { return (size_t)((n_rows + 15) / 16) * 16 * (size_t)NCH * (size_t)K; }
// more than 16 sequences in the B-operand layout: group s / 16 is a layout of its own, 16 * K elements further (xt_group)
LNB_HD size_t xt_group(int s, int K) { return (size_t)(s >> 4) * 16 * (size_t)K; }
LNB_HD size_t xt_index(int s, int k) {
    const int C = k >> 7, e = (k >> 4) & 7, m = (k >> 2) & 3, kk = k & 3;
    return ((((size_t)C * 4 + m) * 4 + kk) * 16 + s) * 8 + e;
}
struct BatchTab {                  // device-resident: the sequences of a batch = the contexts they belong to (column s of every product)
    StepState* st[LNB_BATCH_MAX];  // each context's own position / token counter
    int32_t* dtok[LNB_BATCH_MAX];  // ... next-token word (read by the embedding gather, written by the argmax)
    int32_t* dout[LNB_BATCH_MAX];  // ... generated-token log
    int32_t dout_cap[LNB_BATCH_MAX];
    int32_t seq_len[LNB_BATCH_MAX];   // ... KV-cache length (the K cache layout depends on it)
    int32_t n, pad[3];
};
struct BatchKV { uint16_t* ck[LNB_BATCH_MAX]; uint16_t* cv[LNB_BATCH_MAX]; };   // one per layer: every sequence's caches of that layer

struct StreamParams {              // mfma_stream_kernel: Y[s][n] = trunc(sum_k x_s[k] W[n][k]) for the nseq sequences of a batch
    const uint16_t* w;             // M16 weights
    const uint16_t* xt;            // activations in the B-operand layout (already normalised where the GEMV would fuse the norm)
    int K, n_rows, nch;            // n_rows: logical rows per chain
    int n_chains;                  // tile-chains = ceil(n_rows / 16) * nch
    int n_jobs;                    // jobs of ACC tile-chains each
    int nseq;
    int n_groups;                  // mfma_pair_kernel: column groups of 16 sequences (0 / 1: one); group g reads xt + g * 16 * K and writes columns 16 g ..
    uint16_t* out; const uint16_t* res;            // EPI_STORE / EPI_RESID: [nseq][n_rows] bf16
    uint16_t* out_xt; const float* silu;           // EPI_SILU_MUL: gate*up activations in the B-operand layout of the next product (K' = n_rows)
    const float* cis; uint16_t* q_out; const BatchTab* tab; const BatchKV* kv; int q_dim, kv_dim, head_dim;   // EPI_QKV_ROPE
    long long* dbg;
};

// ---- launch plan of gemm_stream_kernel (host logic; lnb_kernels.hip: launch_gemm_stream) ------------------------------------------------
// Batch tiles of 16 rows per wave (tools/gemmstream_bench.hip over the 8B shapes, profiles/r03_gemmstream_bench.log): 4 -- four matrix
// instructions per unpack op, two workgroups per CU -- as long as the grid still has two workgroups per CU (1.5 for the gate|up pairs, whose
// waves carry two chains); fewer for short prompts / thin matrices, down to 1 (one chain per wave, two or three waves per SIMD: 55-60 % of
// the f32 matrix rate at 128 rows, where a 16 x 16 output tile's one k-ordered chain leaves only 2-3 chains per SIMD to interleave).
// n_tiles = ceil(output rows / 16), ct = ceil(batch rows / 16).
LNB_HD int lnb_gemm_stream_ntw(int n_tiles, int ct, int nch, int num_cus) {
    const long need = nch == 2 ? 3L * num_cus / 2 : 2L * num_cus;
    int ntw = 4;
    while (ntw > 1 && (ntw > ct || (long)((n_tiles + 3) / 4) * ((ct + ntw - 1) / ntw) < need)) ntw >>= 1;
    return ntw;
}
// Dispatch order: row groups fastest once there are more than 8 of them (NOTES.md 5.12, traffic)
LNB_HD int lnb_gemm_stream_rows_fastest(int row_groups) { return row_groups > 8 ? 1 : 0; }
