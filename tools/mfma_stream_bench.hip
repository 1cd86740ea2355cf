// mfma_stream_bench.hip -- round-3 measurements behind the batched exact decode (sequences in flight on the f32 matrix cores).
//
//   part 1  issue-rate questions the design depends on (cycles from s_memtime, one workgroup per CU):
//           a. dependent v_mfma_f32_16x16x4_f32 chain, back to back; with VALU ops between the MFMAs; VALU in batches ahead of runs of MFMAs
//           b. two interleaved accumulator chains (with the unpack VALU of a real stream)
//           c. two DPP chain waves on ONE SIMD against one (the exact GEMVs keep one chain wave per SIMD: is that still right?)
//           d. an MFMA chain wave and a DPP chain wave sharing a SIMD (could the w2 chain run under an MFMA-fed w1|w3?)
//   part 2  the streaming kernel itself in isolation: weights in the 16-row matrix-core layout ("M16"), activations of up to 16 sequences
//           in the matching B-operand layout, HBM -> VGPR -> MFMA A operand, no LDS, no barrier.  Bit-compared with a naive k-ordered fmaf
//           chain per output, then timed per shape (HIP events over launches that cycle through several weight copies).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_stream_bench.hip -o tools/mfma_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define DEVINL __device__ __forceinline__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ part 1
// a/b: NM MFMAs per iteration on ACC accumulators; VB plain VALU ops (independent shifts) placed in front of every run of RUN MFMAs
template <int ACC, int RUN, int VB>
__global__ __launch_bounds__(64) void k_mfma_chain(float* out, long long* cyc, int iters, uint32_t seed) {
    f32x4 acc[ACC];
    for (int a = 0; a < ACC; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t v[16];
    for (int i = 0; i < 16; i++) v[i] = seed * (threadIdx.x + 1 + i) | 0x3f000000u;
    float op[16];
    for (int i = 0; i < 16; i++) op[i] = __uint_as_float(v[i] & 0x3fff0000u);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RUN) {
#pragma unroll
            for (int q = 0; q < VB; q++) op[q & 15] = __uint_as_float((v[q & 15] + (uint32_t)(it + r0)) << 16);    // "unpack": one VALU op each
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RUN; r++)
#pragma unroll
                for (int a = 0; a < ACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[(r + a) & 15], op[(r + 2 * a + 1) & 15], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < ACC; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

DEVINL void chain16(float& acc, const float& p) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(acc) : "v"(p));
}
// c/d: waves of a workgroup by role.  role of wave w = (roles >> (4*w)) & 15: 0 idle, 1 DPP chain (8 x 16 dependent adds per iteration + 8 products),
// 2 dependent MFMA chain with its two unpack ops per MFMA (32 per iteration), 3 plain dependent v_add chain (128 per iteration).
// Waves w and w+4 of a workgroup share a SIMD.
__global__ __launch_bounds__(512) void k_roles(float* out, long long* cyc, int iters, unsigned roles, uint32_t seed) {
    const int wave = threadIdx.x >> 6;
    const int role = (roles >> (4 * wave)) & 15;
    float acc = 0.f; f32x4 macc = {0.f, 0.f, 0.f, 0.f};
    uint32_t v = seed * (threadIdx.x + 3) | 0x3f000000u;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; it++) {
            float pr[8];
#pragma unroll
            for (int i = 0; i < 8; i++) pr[i] = __uint_as_float(((v + it) << 16) & 0x3fff0000u) * 1.0001f;
            asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
#pragma unroll
            for (int i = 0; i < 8; i++) chain16(acc, pr[i]);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int g0 = 0; g0 < 32; g0 += 8) {
                float a[8], b[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { a[q] = __uint_as_float((v + g0 + q) << 16); b[q] = __uint_as_float((v + it + q) & 0x3fff0000u); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) macc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], macc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (role == 4) {
        float p0 = __uint_as_float(v & 0x3fff0000u), a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) {               // 4 independent VALU ops per step, 128 per iteration (an issue-hungry wave)
                const uint32_t w = v + i + it;
                a0 = a0 * 0.5f + __uint_as_float(w << 16); a1 = a1 * 0.5f + __uint_as_float(w & 0xffff0000u);
            }
        }
        acc = a0 + a1 + a2 + a3 + p0;
    } else if (role == 3) {
        const float p = __uint_as_float(v & 0x3fff0000u);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 128; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc) : "v"(p));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc + macc[0] + macc[1] + macc[2] + macc[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ part 2
// M16 weight layout of a logical [N, K] matrix (K % 128 == 0, N % 16 == 0 here), NCH chains per tile:
//   [tile t = n / 16][chain c][chunk C = k / 128][m = (k % 16) / 4][i = n % 16][kk = k % 4][e = (k % 128) / 16]   (bf16)
// i.e. the element mapping of the row-broadcast layout (a 16 B unit = the eight k of one row with the same k % 16), in 16-row tiles: a
// wave-wide 16 B-per-lane load of unit (C, m) is 1 KiB contiguous, and matrix-core lane (i, kk) finds in it, as elements e = 0..7, its A
// operands of the k-groups g = 4e + m of the chunk (k = 128C + 4g + kk).  xt, the activations of up to 16 sequences:
//   [C][m][kk][n = sequence][e]   -- lane (n, kk) loads ITS B operands of the same k-groups with the same instruction shape.
__host__ __device__ inline size_t m16_index(int n, int k, int c, int K, int NCH) {
    const int t = n >> 4, i = n & 15, C = k >> 7, e = (k >> 4) & 7, m = (k >> 2) & 3, kk = k & 3;
    return ((((((size_t)t * NCH + c) * (size_t)(K >> 7) + C) * 4 + m) * 16 + i) * 4 + kk) * 8 + e;
}
__host__ __device__ inline size_t xt_index(int n, int k) {
    const int C = k >> 7, e = (k >> 4) & 7, m = (k >> 2) & 3, kk = k & 3;
    return ((((size_t)C * 4 + m) * 4 + kk) * 16 + n) * 8 + e;
}
DEVINL uint64_t splitmix64(uint64_t z) { z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
// random bf16 with a spread of exponents (signed, |v| in 2^-6 .. 2^2) as a function of (seed, logical index)
DEVINL uint16_t rnd_bf16(uint64_t seed, uint64_t idx) {
    const uint64_t r = splitmix64(seed ^ (idx * 0x9E3779B97F4A7C15ULL));
    const uint32_t e = 121 + (uint32_t)(r % 9), m = (uint32_t)(r >> 8) & 0x7F, s = (uint32_t)(r >> 20) & 1;
    return (uint16_t)((s << 15) | (e << 7) | m);
}
__global__ void k_fill_w(uint16_t* Wm, uint16_t* Wlin, int N, int K, int NCH, uint64_t seed) {     // Wlin may be null
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)N * NCH * K;
    if (idx >= total) return;
    const int k = (int)(idx % K); const size_t rc = idx / K; const int c = (int)(rc % NCH), n = (int)(rc / NCH);
    const uint16_t v = rnd_bf16(seed, idx);
    Wm[m16_index(n, k, c, K, NCH)] = v;
    if (Wlin) Wlin[idx] = v;                                  // [n][c][k]
}
__global__ void k_fill_x(uint16_t* xt, uint16_t* xlin, int nseq, int K, uint64_t seed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16 * K) return;
    const int n = idx / K, k = idx % K;
    const uint16_t v = n < nseq ? rnd_bf16(seed, (uint64_t)idx) : (uint16_t)0;
    xt[xt_index(n, k)] = v; xlin[idx] = v;
}
// the reference's loop per output (operations_lineartransform.go:46-65): acc = acc + x_k * w_k, k ascending (product exact: fmaf)
__global__ void k_naive(const uint16_t* Wlin, const uint16_t* xlin, float* ref, int rows, int K, int nseq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nseq) return;
    const int r = idx / nseq, s = idx % nseq;
    float acc = 0.f;
    for (int k = 0; k < K; k++) acc = fmaf(__uint_as_float((uint32_t)xlin[(size_t)s * K + k] << 16), __uint_as_float((uint32_t)Wlin[(size_t)r * K + k] << 16), acc);
    ref[(size_t)s * rows + r] = acc;
}

// ring loads: asm (hipcc's waitcnt pass drains vmcnt(0) around loop-carried register prefetch), retired by hand-counted waits
DEVINL void ld_w(u32x4& d, unsigned voff, const char* sb, int imm) {
    // (the immediate must be a literal in the asm text: four variants)
    if (imm == 0) asm volatile("global_load_dwordx4 %0, %1, %2 nt ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else if (imm == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else if (imm == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 nt ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 nt ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
}
DEVINL void ld_x(u32x4& d, unsigned voff, const char* sb, int imm) {
    if (imm == 0) asm volatile("global_load_dwordx4 %0, %1, %2 ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else if (imm == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else if (imm == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 ; RING_LOAD" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
}
template <int N, int L> DEVINL void wait_slot(u32x4 (&b)[L]) {
    static_assert(L == 8 || L == 12, "loads per chunk");
    if constexpr (L == 8) asm volatile("s_waitcnt vmcnt(%8) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "n"(N) : "memory");
    if constexpr (L == 12) asm volatile("s_waitcnt vmcnt(%12) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]) : "n"(N) : "memory");
}
DEVINL float unpack(const u32x4& v, int e) { const uint32_t d = v[e >> 1]; return __uint_as_float((e & 1) ? (d & 0xFFFF0000u) : (d << 16)); }

// ACC accumulator chains per wave (the two chains of a gate|up tile, or two tiles, or one tile), R chunks in flight, EB e-values (4 k-groups each)
// unpacked ahead of each run of MFMAs.  out[s][chain_row] = trunc_bf16(acc); chain a of job j is tile-chain j*ACC + a, rows 16*(j*ACC+a) ..
template <int ACC, int R, int EB, int MODE = 0>
__global__ __launch_bounds__(256) void k_stream(const uint16_t* __restrict__ Wm, const uint16_t* __restrict__ xt, uint16_t* __restrict__ out,
                                                float* __restrict__ outf, int n_jobs, int K, int rows_total, int nseq, long long* dbg) {
    constexpr int L = ACC * 4 + 4;
    static_assert(R * L <= 60, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = blockIdx.x * 4 + wave, TW = gridDim.x * 4;
    const int nchunks = K >> 7;
    if (gw >= n_jobs) return;
    const int njobs_mine = (n_jobs - gw + TW - 1) / TW;
    const int T = njobs_mine * nchunks;
    const size_t chain_bytes = (size_t)nchunks * 4096;
    const unsigned aoff = (unsigned)(((lane & 15) * 4 + (lane >> 4)) * 16), boff = (unsigned)lane * 16u;
    const long long t_begin = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    u32x4 buf[R][L];
    int ij = gw, ic = 0, issued = 0;
    auto issue_next = [&](u32x4 (&dst)[L]) {
        const char* wb = (const char*)Wm + (size_t)ij * ACC * chain_bytes + (size_t)ic * 4096;
        const char* xb = (const char*)xt + (size_t)ic * 4096;
#pragma unroll
        for (int a = 0; a < ACC; a++)
#pragma unroll
            for (int m = 0; m < 4; m++) ld_w(dst[a * 4 + m], aoff, wb + (size_t)a * chain_bytes, m);
#pragma unroll
        for (int m = 0; m < 4; m++) ld_x(dst[ACC * 4 + m], boff, xb, m);
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
                wait_slot<(R - 1) * L, L>(buf[j]);          // (never skipped, not even for an experiment: a register with a load in flight that the
                                                             // compiler believes dead gets reused, and the late data then lands in somebody's pointer)
#pragma unroll
                for (int e0 = 0; e0 < 8; e0 += EB) {
                    float av[ACC][4 * EB], bv[4 * EB];
#pragma unroll
                    for (int ee = 0; ee < EB; ee++)
#pragma unroll
                        for (int m = 0; m < 4; m++) {
#pragma unroll
                            for (int a = 0; a < ACC; a++) av[a][ee * 4 + m] = unpack(buf[j][a * 4 + m], e0 + ee);
                            bv[ee * 4 + m] = unpack(buf[j][ACC * 4 + m], e0 + ee);
                        }
                    if (e0 + EB == 8) {                      // every register of the slot has been read: refill it (chunk t + R)
                        // pin the unpacked operands in front of the refill: otherwise hipcc sinks unpack ops below the asm that reloads the
                        // slot and keeps the old value alive through a register copy made BEFORE the wait (found in the ISA: 16 v_mov_b64 of
                        // in-flight registers at the loop head)
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
                    for (int q = 0; q < 4 * EB; q++)         // k-groups g = 4 (e0 + ee) + m ascending: the reference's k order
#pragma unroll
                        for (int a = 0; a < ACC; a++) {
                            if (MODE == 2) { if ((q & 3) == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[a][0]) : "v"(av[a][q] * bv[q])); }
                            else acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][q], bv[q], acc[a], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (++c == nchunks) {
                    const int s = lane & 15;
                    if (s < nseq) {
#pragma unroll
                        for (int a = 0; a < ACC; a++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = (job * ACC + a) * 16 + (lane >> 4) * 4 + r;
                                if (row < rows_total) {
                                    out[(size_t)s * rows_total + row] = (uint16_t)(__float_as_uint(acc[a][r]) >> 16);
                                    if (outf) outf[(size_t)s * rows_total + row] = acc[a][r];
                                }
                            }
                    }
#pragma unroll
                    for (int a = 0; a < ACC; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
                    c = 0; job += TW;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
    if (dbg && lane == 0) dbg[gw] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
}

static double max_of(const long long* h, int n) { double m = 0; for (int i = 0; i < n; i++) if ((double)h[i] > m) m = (double)h[i]; return m; }

template <int ACC, int R, int EB, int MODE = 0>
static void run_shape(const char* name, int N, int K, int NCH, int nseq, int copies, int iters, bool check) {
    // rows of the logical matrix = N * NCH tile-chain rows; jobs = (N / 16) * NCH / ACC
    const int tiles = N / 16, n_jobs = tiles * NCH / ACC, rows_total = N * NCH;
    const size_t welems = (size_t)N * NCH * K;
    uint16_t *Wm = nullptr, *Wlin = nullptr, *xt, *xlin, *out; float *outf = nullptr, *ref = nullptr;
    CHK(hipMalloc(&Wm, welems * 2 * copies));
    if (check) { CHK(hipMalloc(&Wlin, welems * 2)); CHK(hipMalloc(&outf, (size_t)16 * rows_total * 4)); CHK(hipMalloc(&ref, (size_t)16 * rows_total * 4)); }
    CHK(hipMalloc(&xt, (size_t)16 * K * 2)); CHK(hipMalloc(&xlin, (size_t)16 * K * 2)); CHK(hipMalloc(&out, (size_t)16 * rows_total * 2));
    for (int cp = 0; cp < copies; cp++)
        hipLaunchKernelGGL(k_fill_w, dim3((unsigned)((welems + 255) / 256)), dim3(256), 0, 0, Wm + welems * cp, cp == 0 ? Wlin : nullptr, N, K, NCH, 77ull + cp);
    hipLaunchKernelGGL(k_fill_x, dim3((16 * K + 255) / 256), dim3(256), 0, 0, xt, xlin, nseq, K, 5ull);
    CHK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int grid = (n_jobs + 3) / 4; if (grid > cus) grid = cus;
    if (check) {
        // NCH = 2: Wlin is [n][c][k] = tile-chain rows in (n, c) order; the kernel's chain order is [t][c][i]: row id = (t*NCH + c)*16 + i
        hipLaunchKernelGGL((k_stream<ACC, R, EB, MODE>), dim3(grid), dim3(256), 0, 0, Wm, xt, out, outf, n_jobs, K, rows_total, nseq, nullptr);
        hipLaunchKernelGGL(k_naive, dim3((unsigned)(((size_t)rows_total * nseq + 255) / 256)), dim3(256), 0, 0, Wlin, xlin, ref, rows_total, K, nseq);
        CHK(hipDeviceSynchronize());
        std::vector<float> ho((size_t)16 * rows_total), hr((size_t)16 * rows_total);
        CHK(hipMemcpy(ho.data(), outf, ho.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int s = 0; s < nseq; s++)
            for (int n = 0; n < N; n++)
                for (int c = 0; c < NCH; c++) {
                    const int krow = ((n >> 4) * NCH + c) * 16 + (n & 15), lrow = n * NCH + c;
                    if (memcmp(&ho[(size_t)s * rows_total + krow], &hr[(size_t)s * rows_total + lrow], 4)) {
                        if (bad < 6) printf("    differs: seq %d row %d chain %d: kernel %a reference %a\n", s, n, c, ho[(size_t)s * rows_total + krow], hr[(size_t)s * rows_total + lrow]);
                        bad++;
                    }
                }
        printf("  [check] %-22s N=%d K=%d NCH=%d nseq=%d ACC=%d R=%d EB=%d: %ld / %ld outputs differ from the k-ordered fmaf chain\n", name, N, K, NCH, nseq, ACC, R, EB, bad, (long)nseq * N * NCH);
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    long long* dbg; CHK(hipMalloc(&dbg, (size_t)cus * 4 * 8)); CHK(hipMemset(dbg, 0, (size_t)cus * 4 * 8));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_stream<ACC, R, EB, MODE>), dim3(grid), dim3(256), 0, 0, Wm + welems * (i % copies), xt, out, (float*)nullptr, n_jobs, K, rows_total, nseq, (long long*)nullptr);
    CHK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_stream<ACC, R, EB, MODE>), dim3(grid), dim3(256), 0, 0, Wm + welems * (i % copies), xt, out, (float*)nullptr, n_jobs, K, rows_total, nseq, (long long*)nullptr);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL((k_stream<ACC, R, EB, MODE>), dim3(grid), dim3(256), 0, 0, Wm, xt, out, (float*)nullptr, n_jobs, K, rows_total, nseq, dbg);
    CHK(hipDeviceSynchronize());
    std::vector<long long> hd((size_t)cus * 4); CHK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
    const double us = 1e3 * ms / iters, cyc = max_of(hd.data(), (int)hd.size());
    printf("  %-22s N=%6d K=%5d NCH=%d nseq=%2d ACC=%d R=%d EB=%d grid=%3d jobs=%5d: %8.2f us  %7.1f GB/s  (slowest wave %.0f cycles = %.2f per k-step per chain-round)\n",
           name, N, K, NCH, nseq, ACC, R, EB, grid, n_jobs, us, (double)welems * 2 / us / 1e3, cyc, cyc / ((double)K * ((n_jobs + grid * 4 - 1) / (grid * 4))));
    (void)hipFree(Wm); if (Wlin) (void)hipFree(Wlin); if (outf) (void)hipFree(outf); if (ref) (void)hipFree(ref); (void)hipFree(xt); (void)hipFree(xlin); (void)hipFree(out); (void)hipFree(dbg);
}

template <int ACC, int RUN, int VB> static void bench_chain(const char* what, float* out, long long* cyc) {
    const int iters = 400;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k_mfma_chain<ACC, RUN, VB>), dim3(256), dim3(64), 0, 0, out, cyc, iters, 12345u); CHK(hipDeviceSynchronize()); }
    long long h[256]; CHK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    printf("  mfma 16x16x4 f32, %d chain(s), %2d VALU ahead of every run of %2d MFMA(s) per chain: %6.1f cycles per MFMA (%s)\n", ACC, VB, RUN, max_of(h, 256) / (iters * 32.0 * ACC), what);
}
static void bench_roles(const char* what, unsigned roles, float* out, long long* cyc) {
    const int iters = 300;
    for (int rep = 0; rep < 2; rep++) { CHK(hipMemset(cyc, 0, 256 * 8 * 8)); hipLaunchKernelGGL(k_roles, dim3(256), dim3(512), 0, 0, out, cyc, iters, roles, 777u); CHK(hipDeviceSynchronize()); }
    std::vector<long long> h(256 * 8); CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    printf("  %-58s", what);
    for (int w = 0; w < 8; w++) {
        const int role = (roles >> (4 * w)) & 15;
        if (!role) continue;
        double mx = 0; for (int b = 0; b < 256; b++) if ((double)h[b * 8 + w] > mx) mx = (double)h[b * 8 + w];
        printf(" w%d:%s %.2f", w, role == 1 ? "dpp/step" : role == 2 ? "mfma/instr" : role == 4 ? "valu/op" : "add/step", mx / (iters * (role == 2 ? 32.0 : 128.0)));
    }
    printf("\n");
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const bool skip1 = argc > 1 && !strcmp(argv[1], "part2");
    setvbuf(stdout, nullptr, _IONBF, 0);                     // (a GPU fault must not take buffered lines with it)
    float* out; long long* cyc;
    CHK(hipMalloc(&out, 256 * 512 * 4)); CHK(hipMalloc(&cyc, 256 * 8 * 8));
    if (!skip1) {
    printf("part 1a: dependent MFMA chains (s_memtime cycles; 32 issue / 40 dependent expected)\n");
    bench_chain<1, 32, 0>("bare, back to back", out, cyc);
    bench_chain<1, 1, 1>("1 VALU between", out, cyc);
    bench_chain<1, 1, 2>("2 VALU between", out, cyc);
    bench_chain<1, 4, 8>("8 VALU, then 4 MFMA", out, cyc);
    bench_chain<1, 8, 16>("16 VALU, then 8 MFMA", out, cyc);
    bench_chain<1, 16, 32>("32 VALU, then 16 MFMA", out, cyc);
    printf("part 1b: two interleaved chains\n");
    bench_chain<2, 32, 0>("bare", out, cyc);
    bench_chain<2, 4, 12>("12 VALU, then 4 + 4 MFMA", out, cyc);
    bench_chain<2, 8, 24>("24 VALU, then 8 + 8 MFMA", out, cyc);
    printf("part 1c/d: waves by role (w and w+4 share a SIMD); cycles per dependent step\n");
    bench_roles("one DPP chain wave per SIMD (waves 0-3)", 0x00001111u, out, cyc);
    bench_roles("two DPP chain waves per SIMD (waves 0-7)", 0x11111111u, out, cyc);
    bench_roles("one plain v_add chain wave per SIMD", 0x00003333u, out, cyc);
    bench_roles("two plain v_add chain waves per SIMD", 0x33333333u, out, cyc);
    bench_roles("DPP chain (0-3) + dependent MFMA chain (4-7) per SIMD", 0x22221111u, out, cyc);
    bench_roles("dependent MFMA chain alone (4-7)", 0x22220000u, out, cyc);
    bench_roles("two dependent MFMA chains per SIMD", 0x22222222u, out, cyc);
    bench_roles("DPP chain (0-3) + VALU-heavy helper wave (4-7) per SIMD", 0x44441111u, out, cyc);
    bench_roles("DPP chain (4-7, younger) + VALU-heavy helper (0-3)", 0x11114444u, out, cyc);
    }

    printf("part 2: the streaming kernel (weights in the 16-row matrix-core layout, up to 16 sequences)\n");
    // exactness on small shapes (ragged job counts, both chain modes), then the 8B shapes
    run_shape<1, 4, 2>("check one chain", 1024, 512, 1, 16, 1, 2, true);
    run_shape<2, 3, 2>("check gate|up pairs", 2048, 1024, 2, 5, 1, 2, true);
    run_shape<2, 3, 2>("check two tiles", 4096 + 32, 256, 1, 16, 1, 2, true);
    run_shape<2, 3, 2>("check two tiles K=512", 4096 + 32, 512, 1, 16, 1, 2, true);
    run_shape<2, 3, 2>("check two tiles N=4096", 4096, 256, 1, 16, 1, 2, true);
    run_shape<1, 4, 2>("check one chain K=256", 1024, 256, 1, 16, 1, 2, true);
    run_shape<1, 4, 8>("check EB 8", 512, 4096, 1, 3, 1, 2, true);
    if (quick) return 0;
    for (int nseq = 16; nseq >= 1; nseq -= 15) {
        run_shape<2, 3, 2>("w1|w3 (gate|up pairs)", 14336, 4096, 2, nseq, 4, 40, false);
        run_shape<2, 4, 2>("w1|w3 R=4", 14336, 4096, 2, nseq, 4, 40, false);
        run_shape<2, 3, 4>("w1|w3 EB=4", 14336, 4096, 2, nseq, 4, 40, false);
        run_shape<2, 3, 2>("output (two tiles)", 128256, 4096, 1, nseq, 2, 10, false);
        run_shape<1, 4, 2>("wq|wk|wv one tile/wave", 6144, 4096, 1, nseq, 8, 40, false);
        run_shape<1, 6, 2>("wq|wk|wv R=6", 6144, 4096, 1, nseq, 8, 40, false);
        run_shape<1, 4, 8>("wq|wk|wv EB=8", 6144, 4096, 1, nseq, 8, 40, false);
        run_shape<1, 4, 2>("wo one tile/wave", 4096, 4096, 1, nseq, 8, 40, false);
        run_shape<1, 4, 2>("w2 one tile/wave", 4096, 14336, 1, nseq, 4, 40, false);
        if (nseq == 16) {                                   // where the time of a chunk goes: the same stream without its matrix instructions
            run_shape<1, 4, 2, 2>("w2 NO mfma", 4096, 14336, 1, nseq, 4, 40, false);
            run_shape<2, 3, 4, 2>("w1|w3 EB=4 NO mfma", 14336, 4096, 2, nseq, 4, 40, false);
        }
        run_shape<1, 4, 4>("w2 EB=4", 4096, 14336, 1, nseq, 4, 40, false);
        run_shape<1, 4, 8>("w2 EB=8", 4096, 14336, 1, nseq, 4, 40, false);
    }
    return 0;
}
