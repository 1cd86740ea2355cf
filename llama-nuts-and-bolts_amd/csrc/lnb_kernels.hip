// lnb_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for LlamaTransformer.Forward.
//
// Reference behaviour restated (never copied): adalkiran/llama-nuts-and-bolts
//   src/model/llamatransformer.go:145-180 (Forward), :215-254 (block), :289-527 (attention),
//   :593-624 (SwiGLU), :633-660 (RMSNorm), :753-790 (RoPE); src/ml/operations_lineartransform.go:37-70.
//
// Design: "exact-order" kernels.  The reference defines every output as ONE sequential f32 chain over k followed by a
// bf16 truncation, so a kernel never splits k; it only decides which lane walks a chain and how that lane is fed.
// bf16 x bf16 products are exact in f32, so everything except the dependent add can be done by other lanes / waves:
//   * gemv_chain_kernel (norm-fused and fat matrices): weights re-tiled ONCE at load time into [N/RW][K/8][NCH][RW][8]
//     (one linear byte stream per row block); helper waves stream them HBM -> VGPR ring (hand-counted vmcnt), multiply by x
//     and leave f32 products in an LDS ring; ONE chain wave per CU only adds (v_pk_add_f32 for the two-chain w1|w3);
//   * rowcast_kernel (thin wo / w2): products stay in registers; 16 lanes of a DPP row hold the products of 16 consecutive
//     k and every lane adds them in order with v_add_f32_dpp row_newbcast -- no LDS round trip in the chain;
//   * RMSNorm is fused into the consuming GEMV; its sequential f32 sum of squares is evaluated EXACTLY in parallel with
//     composable parity maps (lnb_seqsum.h); RoPE + KV-cache append are fused into the QKV epilogue, SiLU*up and the
//     residual adds are epilogues too: 5 launches per transformer block;
//   * attn_exact_kernel: one position per lane for the scores, f64 softmax in the reference's order, exact-product PV.
// Every register prefetch ring is inline asm with hand-counted waits; tools/isa_audit.py checks the compiled code.
// Compile with -ffp-contract=off: every a*b+c below is either an explicit fmaf (product exact) or
// must stay two roundings.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "lnb_device.h"
#include "lnb_seqsum.h"

#define DEVINL __device__ __forceinline__

DEVINL float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
DEVINL float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
DEVINL float bf_wide(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
DEVINL uint16_t bf_trunc(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }   // bfloat16.go:31-33

DEVINL uint4 ld_nt_u4(const void* p) {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    u4v v = __builtin_nontemporal_load((const u4v*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// one 8-wide k chunk of one chain: acc <- acc + x_k*w_k, k ascending (operations_lineartransform.go:46-65).
// bf16*bf16 is exact in f32, so fmaf == mul-then-add unless the product underflows 2^-126.
DEVINL float mac8(float acc, const float4& xa, const float4& xb, const uint4& w) {
    acc = fmaf(xa.x, bf_lo(w.x), acc); acc = fmaf(xa.y, bf_hi(w.x), acc);
    acc = fmaf(xa.z, bf_lo(w.y), acc); acc = fmaf(xa.w, bf_hi(w.y), acc);
    acc = fmaf(xb.x, bf_lo(w.z), acc); acc = fmaf(xb.y, bf_hi(w.z), acc);
    acc = fmaf(xb.z, bf_lo(w.w), acc); acc = fmaf(xb.w, bf_hi(w.w), acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// The serial chains.  Measured on gfx950 (tools/microbench.hip, profiles/): ONE wave issues one
// instruction per ~4.4 cycles whatever it is, and a dependent v_add_f32 / v_fmac_f32 costs 4.33
// cycles, so a k-ordered chain runs at its latency floor only if the wave that owns it executes ~1
// instruction per k step.  bf16*bf16 products are EXACT in f32, so the chain is split by role:
//   helper waves (other SIMDs of the CU) unpack the bf16 weights, multiply by x and leave the exact f32
//   products in LDS;  the chain wave only does  acc = acc + p_k  (one v_add_f32 per step, operands
//   fetched 4 steps per ds_read_b128), with all 64 lanes active (a partially masked wave issues slower).
// mul-then-add is also literally what the reference does (operations_lineartransform.go:60-64).
// ------------------------------------------------------------------------------------------------
DEVINL float add4(float acc, const float4& p) { acc += p.x; acc += p.y; acc += p.z; acc += p.w; return acc; }
// zero-instruction "use" of 16 loaded registers: makes hipcc place ONE counted s_waitcnt lgkmcnt(N) for the whole
// group (LDS returns in order) instead of one wait per ds_read_b128 result in front of the adds
DEVINL void touch16(const float4& a, const float4& b, const float4& c, const float4& d) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w),
                 "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVINL float4 mul4(const float4& x, float w0, float w1, float w2, float w3) {   // packed f32 multiplies (exact products)
    f32x2 a = {x.x, x.y}, b = {x.z, x.w}, u = {w0, w1}, v = {w2, w3};
    a = a * u; b = b * v;
    return make_float4(a.x, a.y, b.x, b.y);
}


// ---- x staging (cooperative: the chain wave and the helper waves are the NS "stagers") -----------------
// x row -> LDS as f32, zero padded up to kpad (a whole number of stages + 64 floats of slack, so the
// software-pipelined readers may run one group ahead), optionally through the fused RMSNorm.
// The global loads of x are ISSUED BEFORE the loaders start their LDS-DMA burst (barrier B0) -- otherwise the
// 8 KB of x queue behind ~56 KiB of weight prefetch per CU (measured: +4 us per launch).
// 512-element chunks per stager wave held in registers: 12 for the plain kernels (K up to 18k with 3 stagers), 6 for the
// RMSNorm kernels (K = dim <= 8192; they also hold the norm weights and must stay clear of the 256-VGPR budget)
template <bool NORM> struct XCh { static constexpr int value = NORM ? 6 : 12; };
// X_CH (deduced from the register arrays) is a compile-time property of the kernel instance: the launchers pick the smallest count that covers the
// row (2 for K = 4096 -- round 5: the norm-fused kernels used to carry 6 chunks' worth of clamped loads, lane-varying compares and registers
// through x_issue / x_store / x_normalize; of the 2 k cycles x_normalize took, ~1.7 k were control flow around five empty rounds)
template <bool NORM, int NS, int X_CH>
DEVINL void x_issue(const GemvParams& p, const uint16_t* xrow, int sidx, int lane, uint4 (&xv)[X_CH], uint4 (&nv)[X_CH]) {
#pragma unroll
    for (int i = 0; i < X_CH; i++) {
        // UNCONDITIONAL loads (address clamped, value zeroed afterwards): a predicated load would make hipcc
        // wait for each one before issuing the next (measured: 12 serialized round trips)
        const int k = ((i * NS + sidx) * 64 + lane) * 8;
        const int kc = k < p.K ? k : p.K - 8;
        uint4 a = *(const uint4*)(xrow + kc);
        uint4 b = make_uint4(0, 0, 0, 0);
        if (NORM) b = *(const uint4*)(p.norm_w + kc);
        const bool in = k < p.K;
        xv[i] = make_uint4(in ? a.x : 0u, in ? a.y : 0u, in ? a.z : 0u, in ? a.w : 0u);
        nv[i] = b;
    }
}
// squares (NORM: Pow(x,2), exact in f32, operations_impl.go:197-217) or plain values; zeros in [K, kpad)
template <bool NORM, int NS, int X_CH>
DEVINL void x_store(const GemvParams& p, float* xs, int kpad, int sidx, int lane, const uint4 (&xv)[X_CH]) {
#pragma unroll
    for (int i = 0; i < X_CH; i++) {
        const int k = ((i * NS + sidx) * 64 + lane) * 8;
        if (k < kpad) {
            float4 a = make_float4(bf_lo(xv[i].x), bf_hi(xv[i].x), bf_lo(xv[i].y), bf_hi(xv[i].y));
            float4 c = make_float4(bf_lo(xv[i].z), bf_hi(xv[i].z), bf_lo(xv[i].w), bf_hi(xv[i].w));
            if (NORM) {
                a.x = a.x * a.x; a.y = a.y * a.y; a.z = a.z * a.z; a.w = a.w * a.w;
                c.x = c.x * c.x; c.y = c.y * c.y; c.z = c.z * c.z; c.w = c.w * c.w;
            }
            *(float4*)(xs + k) = a; *(float4*)(xs + k + 4) = c;
        }
    }
}
// RMSNorm.doNormalization (llamatransformer.go:641-660): Mean = serial f32 sum of the squares in xs, k ascending
// (operations_impl.go:236-251), /K, +eps (f32), f32(1/sqrt(f64)).
//
// 4096 dependent roundings cost >= 7.4 us as a plain chain (measured ~15 us with its LDS reads) in front of EVERY
// workgroup of the RMSNorm-fused GEMVs.  The sum is evaluated EXACTLY in parallel instead (lnb_seqsum.h): inside one
// binade an f32 add of a non-negative term is the integer map M -> M + c_{M&1}, and such maps compose.
//   helpers (NH waves): lane = one leaf of 8 consecutive squares; approximate prefix sums (wave scan + wave totals) guess
//                       the leaf's binade; the leaf map is folded; a segmented inclusive wave scan composes every run of
//                       valid equal-binade leaves; run ends and invalid leaves are flagged in a 64-bit item mask;
//   chain wave (walker): adds the first RMS_HEAD squares one by one (the sum changes binade every few terms there) while
//                       the helpers fold, then visits only the items: a run's composed map is applied when it starts
//                       exactly at the walker's position, its binade matches the running sum and the sum stays inside it
//                       (a wrong guess can never be used); anything else is replayed term by term.
// Bit-identical to the sequential loop for every input (fuzzed on the CPU: tests/test_seqsum.py emulates this lane by
// lane; on the GPU: tests/test_gpu_parity.py through lnb_op_rmsnorm_linear); ~17 items + ~5 replayed leaves for gaussian x.
// A leaf is seq_leaf_size(K, NH*64) terms (one leaf per folding lane; the last one may run into the zero padding).
constexpr int RMS_HEAD = 256;
__host__ __device__ inline size_t rms_scratch_bytes(int NH) { return (size_t)NH * (512 + 8 + 4 + 4 + 2048 + 8) + 64; }
// LDS scratch (the idle product ring): SeqNode rec[NH][64] | uint64 items[NH] | float wtot[NH] | uint32 scan_failed[NH] |
//                                      SeqItem list[NH][128] (each wave's items of the branch-free walk, in leaf order) | uint32 {count, uncovered}[NH]
// Folding waves: every helper (-DLNB_RMS_NF_MAX=4 restricts the fold to four waves on four different SIMDs -- waves w and w+4 of a workgroup
// share one: HW_ID read per wave, round 4 -- with leaves of 16 terms instead of 12: measured SLOWER, wq|wk|wv 21.2 vs 20.1 us; the fold's time
// follows its instruction count per wave, not the number of waves on a SIMD).  Helpers past the limit only pass the fold's two barriers.
// ring stages a norm-fused kernel's helper issues IN FRONT of the fold (the rest behind it): wq|wk|wv, w1|w3 (two chains), output.
// All R in front stalls the folding waves ~3 k cycles in the issue queue (round 4, first half: everything behind the fold); FOUR of w1|w3's
// eight do not, and the HBM pipe is already streaming when the prologue ends: w1|w3 44.1 -> 43.0 us, block 127.4 -> 126.1 (0 / 2 / 4 / 6 stages:
// 44.1 / 43.6 / 43.0 / 43.8); wq|wk|wv (20.2 -> 20.6 at four) and the output product (159.0 -> 159.9) want none.
#ifndef LNB_EARLY_QUAD
#define LNB_EARLY_QUAD 0
#endif
#ifndef LNB_EARLY_W13
#define LNB_EARLY_W13 4
#endif
#ifndef LNB_EARLY_HEAD
#define LNB_EARLY_HEAD 0
#endif
#ifndef LNB_RMS_NF_MAX
#define LNB_RMS_NF_MAX 8
#endif
#ifndef LNB_QUAD_R
#define LNB_QUAD_R 5                                        // ring stages in flight per helper of gemv_quad_kernel<24, 256, ...> (A/B builds: 4, 6)
#endif
__host__ __device__ constexpr int rms_nf(int NH) { return NH > LNB_RMS_NF_MAX ? LNB_RMS_NF_MAX : NH; }
__host__ __device__ inline size_t rms_list_off(int NH) { return (size_t)NH * 528; }
__host__ __device__ inline size_t rms_meta_off(int NH) { return (size_t)NH * (528 + 2048); }

// wave scans on DPP data movement (VALU, no LDS round trip like ds_bpermute): in-row shifts by 1, 2, 4, 8, then lane 15 of a row
// into the next row (rows 1, 3) and lane 31 into rows 2, 3 -- the classic gfx9 inclusive-scan order.  `old` is what lanes
// without a source keep.
template <int CTRL, int ROWMASK> DEVINL int dpp_mov(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xf, false); }
template <int D> DEVINL int dpp_row_shr(int v, int old) { return dpp_mov<0x110 + D, 0xf>(v, old); }       // lane i <- lane i-D inside its row of 16
DEVINL int dpp_bcast15(int v, int old) { return dpp_mov<0x142, 0xa>(v, old); }                             // rows 1,3 <- lane 15 of the row before
DEVINL int dpp_bcast31(int v, int old) { return dpp_mov<0x143, 0xc>(v, old); }                             // rows 2,3 <- lane 31
DEVINL int dpp_wave_shr1(int v, int old) { return dpp_mov<0x138, 0xf>(v, old); }                           // lane i <- lane i-1 (whole wave)
DEVINL int dpp_wave_shl1(int v, int old) { return dpp_mov<0x130, 0xf>(v, old); }                           // lane i <- lane i+1
DEVINL float wave_inclusive_sum(float v) {
#define LNB_SUM_STEP(EXPR) { const float o_ = __int_as_float(EXPR); v += o_; }
    LNB_SUM_STEP(dpp_row_shr<1>(__float_as_int(v), 0)) LNB_SUM_STEP(dpp_row_shr<2>(__float_as_int(v), 0))
    LNB_SUM_STEP(dpp_row_shr<4>(__float_as_int(v), 0)) LNB_SUM_STEP(dpp_row_shr<8>(__float_as_int(v), 0))
    LNB_SUM_STEP(dpp_bcast15(__float_as_int(v), 0)) LNB_SUM_STEP(dpp_bcast31(__float_as_int(v), 0))
#undef LNB_SUM_STEP
    return v;
}

// sum of a double over the wave, every lane gets it: six DPP steps on the two 32-bit halves (VALU data movement) + two v_readlane, instead of the twelve
// ds_bpermute round trips a __shfl_xor butterfly of doubles costs (~1.2 k cycles in front of attn_exact_kernel's certified p_j).  Only ever used for
// ESTIMATES with a rigorous error bound (the softmax denominator's tree sum): the order of the additions is free.
DEVINL double wave_sum_f64(double v) {
#define LNB_DSUM_STEP(MOV) { const int lo_ = MOV(__double2loint(v), 0), hi_ = MOV(__double2hiint(v), 0); v += __hiloint2double(hi_, lo_); }
    LNB_DSUM_STEP(dpp_row_shr<1>) LNB_DSUM_STEP(dpp_row_shr<2>) LNB_DSUM_STEP(dpp_row_shr<4>) LNB_DSUM_STEP(dpp_row_shr<8>)
    LNB_DSUM_STEP(dpp_bcast15) LNB_DSUM_STEP(dpp_bcast31)
#undef LNB_DSUM_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

DEVINL void rms_load16(float (&tv)[16], const float* q, int c, int LEAF) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 4 * i < LEAF) v = *(const float4*)(q + c + 4 * i);       // (wave-uniform)
        tv[4 * i] = v.x; tv[4 * i + 1] = v.y; tv[4 * i + 2] = v.z; tv[4 * i + 3] = v.w;
    }
}
DEVINL float rms_tree16(const float (&tv)[16]) {
    return (((tv[0] + tv[1]) + (tv[2] + tv[3])) + ((tv[4] + tv[5]) + (tv[6] + tv[7]))) + (((tv[8] + tv[9]) + (tv[10] + tv[11])) + ((tv[12] + tv[13]) + (tv[14] + tv[15])));
}
template <int NH> DEVINL void rms_fold(const GemvParams& p, const float* xs, char* scratch, int hw, int lane, long long& t_dbg) {
    if (hw < 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); return; }      // a helper that does not fold: X1, X2
    const long long tf0_ = p.dbg ? clock64() : 0;
    const int LEAF = seq_leaf_size(p.K, NH * 64), nleaf = (p.K + LEAF - 1) / LEAF;
    SeqNode* rec = (SeqNode*)scratch;
    unsigned long long* items = (unsigned long long*)(scratch + (size_t)NH * 512);
    float* wtot = (float*)(scratch + (size_t)NH * 520);
    const int headleaf = (p.K < RMS_HEAD ? p.K : RMS_HEAD) / LEAF;
    const int b = hw * 64 + lane;
    const float* q = xs + (size_t)(b < nleaf ? b : nleaf - 1) * LEAF;    // (xs is zero padded past K: the last leaf may over-read)
    // the leaf's terms are read from LDS ONCE, into registers (16 per pass; what lies past the leaf is replaced by +0, which changes no sum):
    // the three passes below were bound by their LDS reads (a loop of dependent ds_read_b128 -> wait -> add; lanes 48 or 64 bytes apart
    // conflict 3- or 4-way), not by their arithmetic.  Leaves of more than 16 terms (K > 4096) re-read chunk by chunk.
    float tv[16];
    rms_load16(tv, q, 0, LEAF);
    float bsum = rms_tree16(tv);                                         // only feeds the guess
    for (int c = 16; c < LEAF; c += 16) { rms_load16(tv, q, c, LEAF); bsum += rms_tree16(tv); }
    bsum = b < nleaf ? bsum : 0.0f;
    const float incl = wave_inclusive_sum(bsum);             // (x + 0.0f for lanes without a source: only feeds the guess)
    if (lane == 63) wtot[hw] = incl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();                                        // X1: wave totals published
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float base = 0.0f;
#pragma unroll
    for (int w = 0; w < NH; w++) { const float v = wtot[w]; base += w < hw ? v : 0.0f; }
    SeqNode n; n.a = 0u; n.b = 0u;
    const float lo = base + (incl - bsum), hi = base + incl;           // approximate running sum in front of / behind this leaf
    {   // the leaf's parity map from two simulated f32 running sums (lnb_seqsum.h: seq_leaf): 2 adds per term
        const int32_t e = b < nleaf ? seq_guess_tight(lo, hi) : 0;
        float s0, s1; seq_sim_init(e, s0, s1);
        for (int c = 0; c < LEAF; c += 16) {
            if (LEAF > 16) rms_load16(tv, q, c, LEAF);
#pragma unroll
            for (int i = 0; i < 8; i++) { s0 = s0 + tv[i]; s1 = s1 + tv[i]; }
            if (LEAF > 8) {                                              // (wave-uniform: the terms past the leaf are zeros, skipped instead of added)
#pragma unroll
                for (int i = 8; i < 12; i++) { s0 = s0 + tv[i]; s1 = s1 + tv[i]; }
            }
            if (LEAF > 12) {
#pragma unroll
                for (int i = 12; i < 16; i++) { s0 = s0 + tv[i]; s1 = s1 + tv[i]; }
            }
        }
        if (b < nleaf) n = seq_sim_node(e, s0, s1);
    }
    if (b < nleaf && bsum == 0.0f && (n.a >> 24) == 0u) n.a = SEQ_ZERO_LEAF;      // nothing to add, whatever the running sum is
    // A leaf that enters the next binade: split at the crossing term (lnb_seqsum.h: seq_split_leaf, here branch-free for the whole wave --
    // only waves that hold such a leaf run it): map of the terms before x*, x*, map of the terms after it.
    SeqItem sa, sb2; sa.x = sa.c0 = 0u; sa.d = 0; sa.e = 0u; sb2 = sa;
    bool split_ok = false;
    {
        const bool cand = b < nleaf && (n.a >> 24) == 0u && n.a != SEQ_ZERO_LEAF && seq_split_candidate(lo, hi);
        if (__ballot(cand)) {
            const SeqSplit sp = LEAF <= 12 ? seq_split_leaf(tv, 12, lo, hi) : LEAF <= 16 ? seq_split_leaf(tv, 16, lo, hi) : seq_split_leaf(q, LEAF, lo, hi);
            split_ok = cand && sp.ok; sa = sp.a; sb2 = sp.b;
        }
    }
    SeqNode left; left.a = (uint32_t)dpp_wave_shr1((int)n.a, 0); left.b = 0u;
    int f = seq_is_start(lane, n, left, b == headleaf);
    const int fnext = dpp_wave_shl1(f, 1);
    const unsigned long long mask = __ballot((n.a >> 24) == 0u || lane == 63 || fnext);
    if (lane == 0) items[hw] = mask;
    int start = lane;
    // segmented inclusive scan of the leaf maps (associative: a run absorbs the run that ends right in front of it), six DPP steps
    int failed = 0;                                                      // a composition that did not go through (cannot happen inside a verified binade)
#define LNB_SEG_STEP(MOV, VALID) { SeqNode o_; o_.a = (uint32_t)MOV((int)n.a, 0); o_.b = (uint32_t)MOV((int)n.b, 0); const int ofs_ = MOV((f << 8) | start, 0); \
                                   if ((VALID) && !f) failed |= seq_scan_step(n, f, start, o_, ofs_ >> 8, ofs_ & 0xFF) ^ 1; }
    LNB_SEG_STEP(dpp_row_shr<1>, (lane & 15) >= 1) LNB_SEG_STEP(dpp_row_shr<2>, (lane & 15) >= 2)
    LNB_SEG_STEP(dpp_row_shr<4>, (lane & 15) >= 4) LNB_SEG_STEP(dpp_row_shr<8>, (lane & 15) >= 8)
    LNB_SEG_STEP(dpp_bcast15, (lane & 16) != 0) LNB_SEG_STEP(dpp_bcast31, lane >= 32)
#undef LNB_SEG_STEP
    {   // the wave's items of the branch-free walk, in leaf order: the end of every run (its composed map), the two items of every split leaf;
        // leaves of exact zeros emit nothing; a leaf that nothing covers marks the row for the old walk
        const bool pick = ((mask >> lane) & 1ull) && b >= headleaf && b < nleaf && n.a != SEQ_ZERO_LEAF;
        const bool valid = (n.a >> 24) != 0u;
        // round 6: a leaf that no guess covers -- too close to a binade edge to call (SEQ_MARGIN), neither a run member nor cleanly split -- is replayed INSIDE the
        // branch-free walk as LEAF single-term items (exact adds, no binade assumed) instead of sending the whole row to the record walk (2.7 % of gaussian rows at
        // K = 4096, 6.8 % at 8192 -> ~0 / < 1 %: tests/test_seqsum.py emulates it).  Non-finite squares stay the record walk's business.
        const bool single = pick && !valid && !split_ok && bsum <= 3.4028234e38f;
        const unsigned long long m1 = __ballot(pick && (valid || split_ok)), m2 = __ballot(pick && !valid && split_ok), m3 = __ballot(single);
        const unsigned cntw = (unsigned)(__builtin_popcountll(m1) + __builtin_popcountll(m2)) + (unsigned)LEAF * (unsigned)__builtin_popcountll(m3);
        const bool fits = cntw <= 128u;                                  // (the wave's list segment; the walker takes at most 64 items per row anyway)
        const unsigned long long mu = __ballot(pick && !valid && !split_ok && !single) | (fits ? 0ull : 1ull);
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u)) +
                              __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u)) +
                              (unsigned)LEAF * __builtin_amdgcn_mbcnt_hi((unsigned)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m3, 0u));
        uint4* list = (uint4*)(scratch + rms_list_off(NH)) + hw * 128;
        if (fits) {
            if (pick && valid) { const SeqItem it = seq_item_of_node(n); list[rank] = make_uint4(it.x, it.c0, (uint32_t)it.d, it.e); }
            if (pick && !valid && split_ok) {
                list[rank] = make_uint4(sa.x, sa.c0, (uint32_t)sa.d, sa.e);
                list[rank + 1] = make_uint4(sb2.x, sb2.c0, (uint32_t)sb2.d, sb2.e);
            }
            if (single) for (int i = 0; i < LEAF; i++) list[rank + i] = make_uint4(__float_as_uint(q[i]), 0u, 0u, SEQ_ANY_BINADE);      // (terms from the LDS: rare path)
        }
        if (lane == 0) {
            uint32_t* meta = (uint32_t*)(scratch + rms_meta_off(NH)) + hw * 2;
            meta[0] = fits ? cntw : 0u;
            meta[1] = mu ? 1u : 0u;
        }
    }
    n.b |= (uint32_t)start << 24;                                        // c1 < 2^24: the run's first leaf rides in the top byte
    rec[hw * 64 + lane] = n;
    { const unsigned long long fm = __ballot(failed != 0); if (lane == 0) ((uint32_t*)(scratch + (size_t)NH * 524))[hw] = fm ? 1u : 0u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (p.dbg) t_dbg = clock64() - tf0_;                                 // fold time incl. the X1 wait
    __builtin_amdgcn_s_barrier();                                        // X2: records published
}

// the items of one wave's 64 leaves: rc = this lane's record, from leaf `pos` up to `nloc`; sq = the wave's first square.
// Scalar code (SGPRs + v_readlane); the accept test is evaluated branch-free, the only branch is the rare replay.
DEVINL uint32_t rms_walk_heap(uint32_t sb, const SeqNode& rc, unsigned long long mask, int pos, int nloc, const float* sq, int LEAF) {
    mask &= ~0ull << pos;                                                // items in front of the head are done
    if (nloc < 64) mask &= ~(~0ull << nloc);                             // leaves past the end of the row
    // leaves of exact zeros that no binade could be guessed for (a row that starts with zeros, or is all zeros): not visited, stepped
    // over -- x + 0 == x.  (Visiting them one by one made an all-zero row 1.8x, replaying them 2.7x the kernel time.)
    const unsigned long long zm = __ballot(rc.a == SEQ_ZERO_LEAF);
    mask &= ~zm;
    while (mask) {
        const int i = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)rc.a, i), nb = (uint32_t)__builtin_amdgcn_readlane((int)rc.b, i);
        const int st = (int)(nb >> 24);
        if (st > pos && ((((1ull << st) - 1ull) & (~0ull << pos)) & ~zm) == 0ull) pos = st;       // only zeros between the walker and this run
        const uint32_t e = na >> 24, es = sb >> 23;
        const uint32_t M = (sb & 0x7FFFFFu) | 0x800000u;
        const uint32_t Mn = M + (((M & 1u) ? nb : na) & 0xFFFFFFu);
        const uint32_t ok = (uint32_t)((nb >> 24) == (uint32_t)pos) & (uint32_t)(e == es) & (uint32_t)(e != 0u) & (uint32_t)(Mn < 0x1000000u);
        const uint32_t snew = (es << 23) | (Mn & 0x7FFFFFu);
        if (__builtin_expect(!ok, 0)) {                                  // replay leaves pos..i term by term
            float f = __uint_as_float(sb);
            const float* qe = sq + (size_t)(i + 1) * LEAF;
            for (const float* q = sq + (size_t)pos * LEAF; q < qe; q += 4) f = add4(f, *(const float4*)q);
            sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f));
        } else sb = snew;
        pos = i + 1;
    }
    return sb;
}

// The same walk without the per-item bookkeeping (round 4: 280 -> ~90 cycles per item, 6.8 k -> ~2.5 k cycles per norm).  The items of a wave
// TILE its leaves by construction -- every non-zero leaf is either invalid (an item of its own) or belongs to exactly one run, and a run's
// record covers [its first leaf, the item's lane] -- unless a composition inside the segmented scan failed, which the folding wave reports
// (scan_failed: then rms_walk_heap above, which checks every record's start, is used).  So an item only has to match the running sum's
// binade and keep the sum inside it; anything else is replayed from the leaf behind the previous item (leaves of exact zeros in between
// add nothing).  A replay fetches up to 16 terms at once (one LDS round trip per 16 terms instead of one per 4).
// tq: this wave-set's x^2 terms, lane l holding leaf l's (up to 12) terms -- loaded by the walker while it waits for the fold; `regs` says
// whether they are there (LEAF <= 12).  A replayed leaf then costs a v_readlane + v_add per term (~120 cycles) instead of an LDS round trip
// behind a cold branch (~750 measured).
DEVINL uint32_t rms_walk_fast(uint32_t sb, const SeqNode& rc, unsigned long long mask, int pos, int nloc, const float* sq, int LEAF, int& cnt) {
    mask &= ~0ull << pos;
    if (nloc < 64) mask &= ~(~0ull << nloc);
    mask &= ~__ballot(rc.a == SEQ_ZERO_LEAF);
    // (tools/chainbench4.hip: what this loop pays for is the hand-offs between the vector and the scalar unit -- a dynamic v_readlane
    //  between scalar ops ~50 cycles, a branch on a vector result ~40 even when not taken, a taken branch ~22 -- not the arithmetic:
    //  the same recurrence on the vector unit measured no faster)
    while (mask) {
        const int i = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)rc.a, i), nb = (uint32_t)__builtin_amdgcn_readlane((int)rc.b, i);
        // M + c on the f32 BITS of the running sum: no carry into the exponent field <=> the sum stays inside the binade
        const uint32_t e = na >> 24;
        const uint32_t t = sb + (((sb & 1u) ? nb : na) & 0xFFFFFFu);
        const uint32_t bad = ((t ^ sb) >> 23) | (e ^ (sb >> 23)) | ((e - 1u) >> 31);      // (e == 0 as integer arithmetic: hipcc sent the comparison through the vector unit and back)
        cnt += 0x10000;                                                  // (diagnostics: items << 16 | replays)
        if (__builtin_expect(bad == 0u, 1)) sb = t;
        else {                                                           // replay leaves pos..i term by term
            cnt += 1;
            float f = __uint_as_float(sb);
            {
                const float* qe = sq + (size_t)(i + 1) * LEAF;
                for (const float* q = sq + (size_t)pos * LEAF; q < qe; q += 16) {
                    // (reads past qe stay inside the zero-padded x buffer; what lies past the range is replaced by +0, and f + 0 == f)
                    float4 v0 = *(const float4*)q, v1 = *(const float4*)(q + 4), v2 = *(const float4*)(q + 8), v3 = *(const float4*)(q + 12);
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q + 4 >= qe) v1 = z;
                    if (q + 8 >= qe) v2 = z;
                    if (q + 12 >= qe) v3 = z;
                    f = add4(f, v0); f = add4(f, v1); f = add4(f, v2); f = add4(f, v3);
                }
            }
            sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f));
        }
        pos = i + 1;
    }
    return sb;
}

template <int NH> DEVINL float rms_scale_wide(const GemvParams& p, const float* xs, const char* scratch, int lane, long long& t_dbg) {
    const int K = p.K, LEAF = seq_leaf_size(K, NH * 64), nleaf = (K + LEAF - 1) / LEAF;
    const SeqNode* rec = (const SeqNode*)scratch;
    const unsigned long long* items = (const unsigned long long*)(scratch + (size_t)NH * 512);
    __builtin_amdgcn_s_barrier();                                        // X1 (the helpers' wave totals)
    // head of the sum, term by term, while the helpers fold
    const int headleaf = (K < RMS_HEAD ? K : RMS_HEAD) / LEAF, head = headleaf * LEAF;
    float sum = 0.0f;
    for (int k0 = 0; k0 < head; k0 += 4) sum = add4(sum, *(const float4*)(xs + k0));
    __builtin_amdgcn_s_barrier();                                        // X2 (the records)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum));
    const long long tw0_ = p.dbg ? clock64() : 0;
    int cnt = 0;
    // ---- the row's item list (rms_fold), walked WITHOUT a branch per item -------------------------------------------------------
    // lane w < NH: wave w's item count and "something uncovered" flag; lane g: global item g, fetched from its wave's list segment
    bool done = false;
    int why = 0;
    {
        const uint2 meta = ((const uint2*)(scratch + rms_meta_off(NH)))[lane < NH ? lane : 0];
        const uint32_t failv = ((const uint32_t*)(scratch + (size_t)NH * 524))[lane < NH ? lane : 0];
        const unsigned long long badm = __ballot(lane < NH && (meta.y != 0u || failv != 0u));
        int tot = 0, wsel = 0, below = 0;                                // items in front of lane g's wave
#pragma unroll
        for (int w = 0; w < NH; w++) {
            const int cw = __builtin_amdgcn_readlane((int)meta.x, w);
            tot += cw;
            if (w + 1 < NH) { const bool past = lane >= tot; wsel += past ? 1 : 0; below = past ? tot : below; }
        }
        why = (badm ? 1 << 28 : 0) | (tot > 64 ? 2 << 28 : 0) | ((tot & 0xFF) << 8) | (int)((badm & 0xFF) << 20);
        if (badm == 0ull && tot <= 64) {
            const uint4 it = ((const uint4*)(scratch + rms_list_off(NH)))[wsel * 128 + (lane < tot ? lane - below : 0)];
            const float ix = __uint_as_float(lane < tot ? it.x : 0u);    // lanes past the list: identity items (never reached anyway)
            const uint32_t ic0 = lane < tot ? it.y : 0u;
            const int id = lane < tot ? (int)it.z : 0;
            // systolic recurrence: the running sum hops one lane per step (DPP wave_shr:1); lane g holds its true input at step g, computes
            // u = bits(f32(s) + x), t = u + c0 + (u & 1) d, latches t and hands it on.  Seven vector instructions per item, no scalar work.
            // (inline asm: the compiler turns (u & 1) * d into v_cmp + v_cndmask and the latch into v_cmp + v_cndmask -- two SGPR round trips
            // with their wait states per item; here: add (DPP), bfe, and, add3, latch through a constant lane mask in SGPRs, and the two
            // wait states a DPP read of a just-written VGPR needs -- s_nop 1: whatever the compiler put in front of the first step)
            uint32_t tv, lat = 0u, tmp;
            {   // item 0: every lane from the head's sum
                const uint32_t u0 = __float_as_uint(__uint_as_float(sb) + ix);
                tv = u0 + ic0 + (uint32_t)(-(int)(u0 & 1u) & id);
                lat = lane == 0 ? tv : 0u;
            }
#define LNB_ITEM_STEP(G) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                                      "v_bfe_i32 %2, %0, 0, 1\n\tv_and_b32 %2, %2, %4\n\tv_add3_u32 %0, %0, %5, %2\n\t" \
                                      "v_cndmask_b32_e64 %1, %1, %0, %6" \
                                      : "+v"(tv), "+v"(lat), "=&v"(tmp) : "v"(ix), "v"(id), "v"(ic0), "s"(1ull << (G)));
#define LNB_ITEM_STEP4(G) LNB_ITEM_STEP(G) LNB_ITEM_STEP((G) + 1) LNB_ITEM_STEP((G) + 2) LNB_ITEM_STEP((G) + 3)
            if (tot > 1) { LNB_ITEM_STEP(1) LNB_ITEM_STEP(2) LNB_ITEM_STEP(3) }
#pragma unroll
            for (int G = 4; G < 64; G += 4) if (G < tot) { LNB_ITEM_STEP4(G) }
#undef LNB_ITEM_STEP4
#undef LNB_ITEM_STEP
            // every item's check at once: lane g redoes its step from the latched output of lane g-1 (lane 0: the head's sum).  (The DPP move
            // is NOT inside a conditional: a lane masked off by the condition would be read as "no source lane".)
            const uint32_t lprev = (uint32_t)dpp_wave_shr1((int)lat, 0);
            const uint32_t sin = lane == 0 ? sb : lprev;
            const uint32_t u = __float_as_uint(__uint_as_float(sin) + ix);
            const uint32_t t = u + ic0 + (uint32_t)(-(int)(u & 1u) & id);
            const uint32_t bad = lane < tot ? (((t ^ u) >> 23) | (it.w != SEQ_ANY_BINADE ? (it.w ^ (u >> 23)) : 0u) | (t ^ lat)) : 0u;      // (single-term items assume no binade)
            if (__ballot(bad != 0u) == 0ull) {
                if (tot > 0) sb = (uint32_t)__builtin_amdgcn_readlane((int)lat, tot - 1);
                done = true;
                cnt = tot << 16;
            } else why |= 4 << 28;
        }
    }
    if (!done && p.norm_fb && blockIdx.x == 0 && lane == 0) atomicAdd(p.norm_fb, 1);       // (diagnostics: every workgroup walks the same row, one counts)
    if (!done) {
        // ---- the old walk (every record's binade verified item by item, crossing leaves replayed term by term): rows with a leaf that is
        // neither a run member nor cleanly split (non-finite terms, jumps of several binades, a crossing too close to call), with a guess
        // that did not hold, or with more than 64 items
        SeqNode rc[NH];
        unsigned long long mk[NH];
#pragma unroll
        for (int w = 0; w < NH; w++) { rc[w] = rec[w * 64 + lane]; mk[w] = items[w]; }
        const uint32_t badv = ((const uint32_t*)(scratch + (size_t)NH * 524))[lane < NH ? lane : 0];      // scan_failed[w] in lane w
        const unsigned long long badm = __ballot(badv != 0u);
        cnt = 1;
#pragma unroll
        for (int w = 0; w < NH; w++) {
            int nloc = nleaf - w * 64; nloc = nloc < 0 ? 0 : (nloc > 64 ? 64 : nloc);
            int pos = headleaf - w * 64; pos = pos < 0 ? 0 : pos;
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mk[w]);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mk[w] >> 32));
            if (__builtin_expect((badm >> w) & 1ull, 0)) sb = rms_walk_heap(sb, rc[w], ((unsigned long long)hi32 << 32) | lo32, pos, nloc, xs + (size_t)w * 64 * LEAF, LEAF);
            else sb = rms_walk_fast(sb, rc[w], ((unsigned long long)hi32 << 32) | lo32, pos, nloc, xs + (size_t)w * 64 * LEAF, LEAF, cnt);
        }
    }
    if (p.dbg) t_dbg = clock64() - tw0_;                                 // walk time
    if (!done) cnt = why;
    if (p.dbg && lane == 0) p.dbg[(size_t)4096 * 8 * 4 + ((size_t)blockIdx.x * 8 + 0) * 8 + 7] = cnt;   // stamp slot 7 of the walker: items << 16 | (old walk: 1 + ...)
    float mean = __fdiv_rn(__uint_as_float(sb), (float)K);
    mean = mean + p.eps;
    return (float)(1.0 / sqrt((double)mean));
}
// trunc(x*r) then trunc(.*w): two truncations (llamatransformer.go:656,638), from the registers loaded by x_issue
template <int NS, int X_CH>
DEVINL void x_normalize(const GemvParams& p, float* xs, int kpad, float r, int sidx, int lane, const uint4 (&xv)[X_CH], const uint4 (&nv)[X_CH]) {
#pragma unroll
    for (int i = 0; i < X_CH; i++) {
        const int k = ((i * NS + sidx) * 64 + lane) * 8;
        if (k >= p.K && k < kpad) {   // the (padded) squares may have spilled past K: the stage walkers need zeros there
            *(float4*)(xs + k) = make_float4(0.f, 0.f, 0.f, 0.f); *(float4*)(xs + k + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (k < p.K) {
            const uint4 v = xv[i], wv = nv[i];
            float4 a, c;
            a.x = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_lo(v.x) * r)) * bf_lo(wv.x)));
            a.y = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_hi(v.x) * r)) * bf_hi(wv.x)));
            a.z = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_lo(v.y) * r)) * bf_lo(wv.y)));
            a.w = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_hi(v.y) * r)) * bf_hi(wv.y)));
            c.x = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_lo(v.z) * r)) * bf_lo(wv.z)));
            c.y = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_hi(v.z) * r)) * bf_hi(wv.z)));
            c.z = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_lo(v.w) * r)) * bf_lo(wv.w)));
            c.w = bf_wide(bf_trunc(bf_wide(bf_trunc(bf_hi(v.w) * r)) * bf_hi(wv.w)));
            *(float4*)(xs + k) = a; *(float4*)(xs + k + 4) = c;
        }
    }
}

// PX: lane distance of the RoPE partner row (2i <-> 2i+1): 1 when a lane is a row, 4 in gemv_quad_kernel (four lanes per row)
template <int NCH, int EPI, int PX = 1>
DEVINL void gemv_epilogue(const GemvParams& p, const float (&acc)[NCH], int m, int n, bool valid) {
    if (EPI == EPI_STORE) {
        if (valid) p.out[(size_t)m * p.n_rows + n] = bf_trunc(acc[0]);
    } else if (EPI == EPI_RESID) {
        // ml.Add (operations_impl.go:320-332): trunc(wide(x) + wide(y)), y = trunc(acc)
        if (valid) {
            size_t o = (size_t)m * p.n_rows + n;
            p.out[o] = bf_trunc(bf_wide(p.res[o]) + bf_wide(bf_trunc(acc[0])));
        }
    } else if (EPI == EPI_SILU_MUL) {
        // Silu table lookup on the raw bf16 bits (activations.go:36-39) then MultiplyElementwise (:614)
        if (valid) {
            uint16_t g = bf_trunc(acc[0]), u = bf_trunc(acc[NCH - 1]);
            uint16_t gs = bf_trunc(p.silu[g]);
            p.out[(size_t)m * p.n_rows + n] = bf_trunc(bf_wide(gs) * bf_wide(u));
        }
    } else if (EPI == EPI_QKV_ROPE) {
        // rows [0,q_dim) = xq, [q_dim,q_dim+kv_dim) = xk, rest = xv (llamatransformer.go:297-384)
        const int pos = p.st->pos + m;
        const uint16_t mine = bf_trunc(acc[0]);
        const uint16_t other = (uint16_t)__shfl_xor((int)mine, PX);      // RoPE partner (2i <-> 2i+1)
        if (valid) {
            if (n < p.q_dim + p.kv_dim) {
                // applyRotaryEmbeddings (:753-790): complex64 product evaluated in f64, narrowed, truncated
                const int d = n % p.head_dim, i = d >> 1;
                const float2 cs = *(const float2*)(p.cis + ((size_t)pos * (p.head_dim >> 1) + i) * 2);
                const double cr = (double)cs.x, ci = (double)cs.y;
                uint16_t r16;
                if ((n & 1) == 0) { double a = (double)bf_wide(mine), bb = (double)bf_wide(other); r16 = bf_trunc((float)(a * cr - bb * ci)); }
                else              { double a = (double)bf_wide(other), bb = (double)bf_wide(mine); r16 = bf_trunc((float)(a * ci + bb * cr)); }
                if (n < p.q_dim) p.q_out[(size_t)m * p.q_dim + n] = r16;
                else {                                                                // :402, K cache layout [kv head][d/8][position][8]
                    const int kc = n - p.q_dim, kh = kc / p.head_dim;
                    p.cache_k[(((size_t)kh * (p.head_dim >> 3) + (d >> 3)) * p.seq_len + pos) * 8 + (d & 7)] = r16;
                }
            } else {
                p.cache_v[(size_t)pos * p.kv_dim + (n - p.q_dim - p.kv_dim)] = mine;  // :403
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Exact-order streaming GEMV / skinny GEMM:   y[m, n] = trunc( sum_{k ascending} x[m,k] * W[n,k] )
//
// workgroup = 1 + NH waves, ONE workgroup per CU (every wave on its own SIMD: two issue-hungry waves on
// one SIMD halve each other's rate), ONE raw s_barrier per stage:
//   waves 1..NH  helpers: stream the block's bf16 weights HBM -> VGPR (16 B/lane non-temporal loads, a
//                         register ring of R stages in flight, COUNTED s_waitcnt vmcnt(N) -- never 0 in
//                         steady state), multiply by x (exact products) and leave f32 products
//                         [k4][chain][row][4] in a double-buffered LDS ring;
//   wave 0       chain  : RW rows (lane & (RW-1)), NCH chains per lane: per 16 k-steps 4 ds_read_b128 +
//                         16 v_add_f32 per chain.
// iteration t: helpers produce stage t | chain walks stage t-1.
// (An LDS-DMA loader wave + separate multiplier waves was measured first: the extra LDS round trip of the
//  bf16 bytes and the DMA's arbitration against ds traffic capped a CU at ~12 GB/s; see DESIGN.md.)
// A workgroup is PERSISTENT over its row blocks (b = wg, wg + n_wg, ...): x is staged (and RMS-normalised)
// once, the load pipeline runs across block boundaries, the epilogue runs at every block end.
// "thin" matrices (wq|wk|wv, wo, w2: latency bound) use RW=16/32 with one block per workgroup on all CUs;
// "fat" ones (w1|w3, output: HBM bound) use RW=64.
// grid.x = S * n_wg (m fastest so that the S workgroups sharing a weight block run together)
// dynamic LDS: [2 * 2*SA product ring][kpad f32 x + 16 B]
// ------------------------------------------------------------------------------------------------
// optional per-wave timing (GemvParams.dbg != nullptr): [wg][wave][4] = {total, barrier wait, x staging / vm wait, -} in s_memtime ticks
#define TIMED_BARRIER() do { if (p.dbg_full) { long long tb_ = clock64(); __builtin_amdgcn_s_barrier(); t_wait += clock64() - tb_; } else __builtin_amdgcn_s_barrier(); } while (0)
// phase stamp i (0..7) of this wave, cycles since the kernel started: a second region behind the [4096][8][4] totals (lnb_api.cpp prints the averages)
#define LNB_STAMP(i) do { if (p.dbg && lane == 0) p.dbg[(size_t)4096 * 8 * 4 + ((size_t)blockIdx.x * 8 + wave) * 8 + (i)] = clock64() - t_begin; } while (0)
// ... and stamp 7 = the same interval on the constant-rate wall clock (s_memrealtime; hipDeviceAttributeWallClockRate): the shader clock of THIS launch
// under ITS load = total / wall, which is what turns the cycle counts into microseconds (bench.py: roofline.measured_model)
#define LNB_WALL_EXIT() do { if (p.dbg && lane == 0) p.dbg[(size_t)4096 * 8 * 4 + ((size_t)blockIdx.x * 8 + wave) * 8 + 7] = (long long)wall_clock64() - t_wall0; } while (0)
#define DBG_EXIT() do { if (p.dbg && lane == 0) { long long* d_ = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4; d_[0] = clock64() - t_begin; d_[1] = t_wait; d_[2] = t_x; d_[3] = t_aux; } LNB_WALL_EXIT(); } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// asm loads are invisible to hipcc's s_waitcnt bookkeeping: the ring below is waited for by hand (wait_ring)
// saddr form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset -> no per-load VALU address arithmetic
DEVINL void ld_nt_asm(u32x4& dst, unsigned voff, const char* sbase) { asm volatile("global_load_dwordx4 %0, %1, %2 nt ; RING_LOAD" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory"); }
// the "; RING_RETIRE ..." comment names the registers this wait retires: tools/isa_audit.py checks on the compiled
// code that hipcc touches no ring register between its asm load and the wait that retires it
template <int N, int NP> DEVINL void wait_ring(u32x4 (&b)[NP]) {
    static_assert(NP == 1 || NP == 2 || NP == 4 || NP == 8, "loads per stage per helper");
    if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(%1) ; RING_RETIRE %0" : "+v"(b[0]) : "n"(N) : "memory");
    if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(%2) ; RING_RETIRE %0 %1" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    if constexpr (NP == 4) asm volatile("s_waitcnt vmcnt(%4) ; RING_RETIRE %0 %1 %2 %3" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
    if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(%8) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "n"(N) : "memory");
}

template <int RW, int NCH, int SA, int NH, int R, int EPI, bool NORM, int XC = XCh<NORM>::value>
__global__ __launch_bounds__((1 + NH) * 64) void gemv_chain_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long t_begin = p.dbg ? clock64() : 0, t_wait = 0, t_x = 0, t_aux = 0;
    const long long t_wall0 = p.dbg ? (long long)wall_clock64() : 0;
    constexpr int KC = SA / (NCH * RW * 16);              // 8-wide k chunks per stage
    constexpr int GS = KC / 2;                            // 16-step groups per stage
    constexpr int SB = 2 * SA;                            // f32 product stage
    constexpr int UNITS = SA / 16;                        // 16-byte (8 x bf16) units per stage
    constexpr int NP = UNITS / (64 * NH);                 // 16 B loads per lane per stage per helper
    static_assert(KC >= 2 && (KC & 1) == 0, "bad stage geometry");
    static_assert(UNITS % (64 * NH) == 0 && R * NP <= 60, "vmcnt is a 6-bit counter");
    static_assert(NH >= 2, "x staging is sized for >= 3 stager waves");
    char* ringB = smem;
    float* xs = (float*)(smem + 2 * SB);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, S = p.S;
    const int m = (S == 1) ? 0 : (int)(blockIdx.x % (unsigned)S);
    const int wg = (S == 1) ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned)S);
    const size_t stream_bytes = (size_t)K * NCH * RW * 2;
    const int nstages = (int)((stream_bytes + SA - 1) / SA);
    const int nb_mine = (p.n_blocks - wg + p.n_wg - 1) / p.n_wg;       // row blocks wg, wg+n_wg, ...
    const int T = nb_mine * nstages;                                   // global stage count of this workgroup
    constexpr int NS = 1 + NH;                                         // x stagers: chain wave + helpers
    const int kpad = nstages * KC * 8 + 320;                           // launcher guarantees kpad <= X_CH*NS*512
    const uint16_t* xrow = p.x + (size_t)m * K;

    // wave roles: with more than 3 helpers the chain wave is wave 3 -- waves w and w+4 share a SIMD, so waves 0..2
    // and 4..6 (six helpers) pair up on three SIMDs and the chain wave keeps the fourth one to itself
    constexpr int CW = (NH > 3) ? 3 : 0;
    if (wave != CW) {
        // ================================ helper waves ============================================
        const int hw = wave < CW ? wave : wave - 1;
        u32x4 buf[R][NP];
        {
            uint4 xv[XC], nv[XC];
            x_issue<NORM, NS>(p, xrow, 1 + hw, lane, xv, nv);
            __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): x (and the norm weights) have landed; hipcc needs no
                                                                       // wait of its own in x_store, which would drain the ring below too
        // the weight stream starts only now, BEHIND the x loads in this CU's memory queue (in front of them, or right behind them
        // with a counted wait, it delays x: measured), and the LDS writes of x_store run in its shadow
        // issue cursor: next global stage to load = (block ib, stage is); past the end it re-reads the last stage
        const size_t last16 = stream_bytes - 16;
        // unit of load i:  q = (kc*NCH + c)*RW + r.  One chain: q = (i*NH + hw)*64 + lane.  Two chains (w1|w3): the lane takes BOTH
        // chains of (kc, r) = ((hw*64 + lane) / RW, (hw*64 + lane) % RW) -- load i is chain i -- so that it can leave the
        // products pair-interleaved (gate_k, up_k) for the chain wave's v_pk_add_f32
        static_assert(NCH == 1 || (NCH == 2 && NP % 2 == 0), "two chains: an even number of loads per lane (pairs)");
        unsigned loff[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int pq = ((i >> 1) * NH + hw) * 64 + lane;               // pair index of load i (two chains)
            loff[i] = NCH == 2 ? (unsigned)((((pq / RW) * 2 + (i & 1)) * RW + (pq % RW)) * 16) : (unsigned)(((i * NH + hw) * 64 + lane) * 16);
        }
        int ib = wg, is = 0, issued = 0;
        auto issue_next = [&](u32x4 (&dst)[NP]) {
            const size_t soff = (size_t)is * SA;
            const char* sb = (const char*)p.w + (size_t)ib * stream_bytes + soff;     // wave-uniform
            // branch-free clamp (only the last, partial stage of a block ever clamps): one v_min_u32 per load
            const unsigned lim = (soff + SA <= stream_bytes) ? 0xFFFFFFFFu : (unsigned)(last16 - soff);
#pragma unroll
            for (int i = 0; i < NP; i++) ld_nt_asm(dst[i], loff[i] < lim ? loff[i] : lim, sb);
            if (issued + 1 < T) { issued++; if (++is == nstages) { is = 0; ib += p.n_wg; } }
        };
            // norm-fused kernels start the weight stream BEHIND the norm's fold (round 4): issuing R stages per helper into a cold memory
            // pipeline stalls the issuing waves ~3 k cycles -- the waves that fold; behind the fold they only wait for the walker anyway
            // (measured: wq|wk|wv -1.0 us, w1|w3 -2.3 us, output -1 us)
            constexpr bool late = NORM;
            constexpr int EARLY_ = NCH == 2 ? LNB_EARLY_W13 : LNB_EARLY_HEAD;
            constexpr int EARLY = late ? (EARLY_ < R ? EARLY_ : R) : R;      // stages issued in front of the fold
#pragma unroll
            for (int j = 0; j < EARLY; j++) issue_next(buf[j]);
            x_store<NORM, NS>(p, xs, kpad, 1 + hw, lane, xv);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            TIMED_BARRIER();                                           // B1: xs (or the squares) are in LDS
            if (NORM) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                rms_fold<rms_nf(NH)>(p, xs, ringB, hw - (NH - rms_nf(NH)), lane, t_aux);      // X1, X2 inside; the LAST four helpers fold (waves 4..7 of 8: four SIMDs)
                {                                                      // the walker walks now: the issue stall costs nothing here
#pragma unroll
                    for (int j = EARLY; j < R; j++) issue_next(buf[j]);
                }
                TIMED_BARRIER();                                       // B2: r published
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if constexpr (NORM) x_normalize<NS>(p, xs, kpad, xs[kpad], 1 + hw, lane, xv, nv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                TIMED_BARRIER();                                       // B3: xs normalised
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int st = 0;                                        // stage-in-block of the stage being converted
        for (int it0 = 0; it0 <= T; it0 += R) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int t = it0 + j;
                if (t <= T) {
                    if (t < T) {
                        const long long tb_ = p.dbg_full ? clock64() : 0;
                        wait_ring<(R - 1) * NP, NP>(buf[j]);           // stage t landed; R-1 younger stages stay in flight
                        if (p.dbg_full) t_x += clock64() - tb_;
                        char* dst = ringB + (t & 1) * SB;
                        const float* xst = xs + (size_t)st * (KC * 8);
                        if constexpr (NCH == 2) {
                            // both chains of (kc, r): products written pair-interleaved, [k4][half][r] x (gate_s, up_s, gate_s+1, up_s+1)
#pragma unroll
                            for (int pi = 0; pi < NP / 2; pi++) {
                                const int pq = (pi * NH + hw) * 64 + lane, kc = pq / RW, r = pq % RW;
                                const float4 xa = *(const float4*)(xst + kc * 8), xb = *(const float4*)(xst + kc * 8 + 4);
                                const u32x4 vg = buf[j][2 * pi], vu = buf[j][(2 * pi + 1) % NP];
                                const float4 ga = mul4(xa, bf_lo(vg.x), bf_hi(vg.x), bf_lo(vg.y), bf_hi(vg.y)), gb = mul4(xb, bf_lo(vg.z), bf_hi(vg.z), bf_lo(vg.w), bf_hi(vg.w));
                                const float4 ua = mul4(xa, bf_lo(vu.x), bf_hi(vu.x), bf_lo(vu.y), bf_hi(vu.y)), ub = mul4(xb, bf_lo(vu.z), bf_hi(vu.z), bf_lo(vu.w), bf_hi(vu.w));
                                char* d = dst + r * 16;
                                *(float4*)(d + ((4 * kc + 0) * RW) * 16) = make_float4(ga.x, ua.x, ga.y, ua.y);
                                *(float4*)(d + ((4 * kc + 1) * RW) * 16) = make_float4(ga.z, ua.z, ga.w, ua.w);
                                *(float4*)(d + ((4 * kc + 2) * RW) * 16) = make_float4(gb.x, ub.x, gb.y, ub.y);
                                *(float4*)(d + ((4 * kc + 3) * RW) * 16) = make_float4(gb.z, ub.z, gb.w, ub.w);
                            }
                        } else {
                        float4 xa[NP], xb[NP];
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            const int q = (i * NH + hw) * 64 + lane;       // q = (kc*NCH + c)*RW + r
                            const int kc = q / (NCH * RW);
                            xa[i] = *(const float4*)(xst + kc * 8); xb[i] = *(const float4*)(xst + kc * 8 + 4);
                        }
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            const int q = (i * NH + hw) * 64 + lane;
                            const int kc = q / (NCH * RW), u = q % (NCH * RW);
                            const u32x4 v = buf[j][i];
                            // exact products (8-bit x 8-bit significands): val1F32 * val2F32, operations_lineartransform.go:60
                            *(float4*)(dst + ((2 * kc) * (NCH * RW) + u) * 16) = mul4(xa[i], bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
                            *(float4*)(dst + ((2 * kc + 1) * (NCH * RW) + u) * 16) = mul4(xb[i], bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
                        }
                        }
                        if (++st == nstages) st = 0;
                        __builtin_amdgcn_sched_barrier(0);             // refill AFTER the slot has been consumed (no register copies)
                        issue_next(buf[j]);                            // refill this register slot with stage t+R
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // ds_writes complete before the barrier
                    TIMED_BARRIER();
                }
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
        DBG_EXIT();
        return;
    }
    // ==================================== chain wave ==================================================
    {
        uint4 xv[XC], nv[XC];
        x_issue<NORM, NS>(p, xrow, 0, lane, xv, nv);
        x_store<NORM, NS>(p, xs, kpad, 0, lane, xv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        TIMED_BARRIER();                                               // B1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (NORM) {
            const float r = rms_scale_wide<rms_nf(NH)>(p, xs, ringB, lane, t_aux);     // X1, X2 inside
            if (lane == 0) xs[kpad] = r;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            TIMED_BARRIER();                                           // B2
            if constexpr (NORM) x_normalize<NS>(p, xs, kpad, r, 0, lane, xv, nv);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            TIMED_BARRIER();                                           // B3
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (p.dbg) t_x = clock64() - t_begin;
    LNB_STAMP(6);                                                      // (stamp 6 marks a chain wave: the cycle its main loop starts)
    float acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[c] = 0.0f;
    const int row = (RW & (RW - 1)) == 0 ? (lane & (RW - 1)) : (lane % RW);   // RW 56 / 28: the lanes past RW shadow rows 0.. (every lane stays active: a partially masked wave issues slower)
    int st = 0, blk = wg;
    // every stage is walked in full: beyond K the x values (hence the products) are +0, and acc + 0 == acc
    // because acc is never -0
    for (int it = 0; it <= T; it++) {
        const int t = it - 1;
        if (t >= 0) {
            const char* src = ringB + (t & 1) * SB + row * 16;
            // software pipeline: group g+1's 16 products per chain are in flight while group g is added
            float4 pb[2][NCH][4];
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int j = 0; j < 4; j++) pb[0][c][j] = *(const float4*)(src + ((j * NCH + c) * RW) * 16);
#pragma unroll
            for (int g = 0; g < GS; g++) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g + 1 < GS) {
#pragma unroll
                    for (int c = 0; c < NCH; c++)
#pragma unroll
                        for (int j = 0; j < 4; j++) pb[nxt][c][j] = *(const float4*)(src + (((4 * (g + 1) + j) * NCH + c) * RW) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the adds (hipcc otherwise sinks it next to its use)
#pragma unroll
                for (int c = 0; c < NCH; c++) touch16(pb[cur][c][0], pb[cur][c][1], pb[cur][c][2], pb[cur][c][3]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NCH == 2) {
                    // pb[cur][half][j] = (gate_s, up_s, gate_s+1, up_s+1): both chains advance with one v_pk_add_f32 per k-step
                    // (6.0 cycles per dependent op against 2 x 4.33; written as separate adds hipcc paired them anyway, behind two
                    // v_mov per step to build the register pairs)
                    f32x2 a2 = {acc[0], acc[1]};
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float4 q4 = pb[cur][h][j];
                            a2 = a2 + f32x2{q4.x, q4.y};
                            a2 = a2 + f32x2{q4.z, q4.w};
                        }
                    acc[0] = a2.x; acc[1] = a2.y;
                } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int c = 0; c < NCH; c++) acc[c] = add4(acc[c], pb[cur][c][j]);   // valDstF32 += p, k ascending (:63)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++st == nstages) {                                     // end of a row block: write it out, start the next
                gemv_epilogue<NCH, EPI>(p, acc, m, blk * RW + lane, (lane < RW) && (blk * RW + lane < p.n_rows));
#pragma unroll
                for (int c = 0; c < NCH; c++) acc[c] = 0.0f;
                st = 0; blk += p.n_wg;
            }
        }
        TIMED_BARRIER();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    DBG_EXIT();
}

// ------------------------------------------------------------------------------------------------
// gemv_quad_kernel (round 4): the norm-fused thin product wq|wk|wv on ALL CUs, its chains fed through the LDS AND by DPP.
//
// gemv_chain_kernel gives a chain lane its own products (one ds_read_b128 per 4 steps: 8.7-9.0 cycles per step, the returning data
// occupies the SIMD's register write port) and 6144 rows in 32-row blocks fill 192 of the 256 CUs.  A DPP operand serves several
// lanes at once: with quad_perm a row lives in FOUR lanes, lane j of the quad receives the products k = 16g + 4j .. 4j + 3 of a
// 16-step group in one ds_read_b128 and every lane of the quad adds all sixteen in k order,
//     v_add_f32_dpp acc, p_e, acc quad_perm:[j,j,j,j]     e = 0..3 inside j = 0..3,
// i.e. one LDS read per 16 steps and 16 rows per chain wave: 6.8 cycles per step (tools/chainbench3.hip; 7.5 with helpers hammering
// the LDS).  A block is RW = n_rows / 256 rows (24 for the 8B shape): ceil(RW / 16) chain waves, each on its own SIMD.
//   waves 0 .. NCW-1    chain  : rows 16w .. 16w+15 of the block (lanes past the last row shadow earlier ones: every lane stays active);
//   waves NCW .. NCW+NH-1 helper: stream the block's bf16 weights (layout [N/RW][K/8][RW][8], the register ring of gemv_chain_kernel),
//                                  multiply by x and leave the exact f32 products in the LDS as [16-step group][row][4 x 16 B], the
//                                  16-byte slots of a row XOR-swizzled with (row >> 1) & 3 so that both the helpers' ds_write_b128
//                                  (eight consecutive rows of one k-chunk) and the chain waves' ds_read_b128 are conflict-free;
// stages of KS steps in a three-slot ring, one s_barrier per stage, the chain waves two stages behind the helpers so that their
// prefetch (three groups ahead) runs across stage boundaries (as in rowcast_lds_kernel).  The fused RMSNorm prologue is
// gemv_chain_kernel's: the helpers fold, chain wave 0 walks.
// grid.x = S * n_wg, block = (NCW + NH) * 64, dynamic LDS = 3 * RW * KS * 4 + x.
// ------------------------------------------------------------------------------------------------
#define LNB_QP(j) " quad_perm:[" #j "," #j "," #j "," #j "] row_mask:0xf bank_mask:0xf\n\t"
#define LNB_QADD4(j) "v_add_f32_dpp %0, %1, %0" LNB_QP(j) "v_add_f32_dpp %0, %2, %0" LNB_QP(j) "v_add_f32_dpp %0, %3, %0" LNB_QP(j) "v_add_f32_dpp %0, %4, %0" LNB_QP(j)
// the 16 dependent adds of one group in ONE asm statement (separate statements get an s_nop each from hipcc's hazard recognizer);
// the leading s_nop covers the VGPR-write -> DPP-read wait states for whatever was scheduled last
DEVINL void chain16q(float& acc, const float4& q) {
    asm volatile("s_nop 1\n\t" LNB_QADD4(0) LNB_QADD4(1) LNB_QADD4(2) LNB_QADD4(3) : "+v"(acc) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
}
constexpr int GQ_SLOTS = 3;
__host__ __device__ constexpr int gq_ncw(int RW) { return (RW + 15) / 16; }
template <int RW, int KS, int NH, int R, int EPI, bool NORM, int XC = XCh<NORM>::value>
__global__ __launch_bounds__((gq_ncw(RW) + NH) * 64) void gemv_quad_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long t_begin = p.dbg ? clock64() : 0, t_wait = 0, t_x = 0, t_aux = 0;
    const long long t_wall0 = p.dbg ? (long long)wall_clock64() : 0;
    constexpr int NCW = gq_ncw(RW), NW = NCW + NH;
    constexpr int KC = KS / 8;                            // 8-wide k chunks per stage
    constexpr int GS = KS / 16;                           // 16-step groups per stage
    constexpr int SA = RW * KS * 2;                       // bf16 bytes per stage
    constexpr int SB = 2 * SA;                            // f32 product bytes per stage
    constexpr int UNITS = RW * KC;                        // 16-byte weight units per stage
    constexpr int NP = UNITS / (64 * NH);                 // loads per lane per stage per helper
    static_assert(RW % 8 == 0 && KS % 16 == 0, "rows in eights (conflict-free product writes), whole 16-step groups");
    static_assert(UNITS % (64 * NH) == 0 && R * NP <= 60, "vmcnt is a 6-bit counter");
    static_assert(NH >= 2 && GS >= 4 && GS % 4 == 0, "stage geometry");
    char* ringB = smem;
    float* xs = (float*)(smem + GQ_SLOTS * SB);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, S = p.S;
    const int m = (S == 1) ? 0 : (int)(blockIdx.x % (unsigned)S);
    const int wg = (S == 1) ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned)S);
    const size_t stream_bytes = (size_t)K * RW * 2;
    const int nstages = (int)((stream_bytes + SA - 1) / SA);
    const int nb_mine = (p.n_blocks - wg + p.n_wg - 1) / p.n_wg;       // row blocks wg, wg+n_wg, ...
    const int T = nb_mine * nstages;                                   // stages of this workgroup
    constexpr int NS = NW;                                             // every wave stages x
    const int kpad = nstages * KS + 320;                               // launcher guarantees kpad <= X_CH*NS*512
    const uint16_t* xrow = p.x + (size_t)m * K;

    if (wave >= NCW) {
        // ================================ helper waves ============================================
        const int hw = wave - NCW;
        u32x4 buf[R][NP];
        uint4 xv[XC], nv[XC];
        x_issue<NORM, NS>(p, xrow, wave, lane, xv, nv);
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): x (and the norm weights) have landed (see gemv_chain_kernel)
        const size_t last16 = stream_bytes - 16;
        unsigned loff[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) loff[i] = (unsigned)(((i * NH + hw) * 64 + lane) * 16);
        int ib = wg, is = 0, issued = 0;
        auto issue_next = [&](u32x4 (&dst)[NP]) {
            const size_t soff = (size_t)is * SA;
            const char* sb = (const char*)p.w + (size_t)ib * stream_bytes + soff;     // wave-uniform
            const unsigned lim = (soff + SA <= stream_bytes) ? 0xFFFFFFFFu : (unsigned)(last16 - soff);
#pragma unroll
            for (int i = 0; i < NP; i++) ld_nt_asm(dst[i], loff[i] < lim ? loff[i] : lim, sb);
            if (issued + 1 < T) { issued++; if (++is == nstages) { is = 0; ib += p.n_wg; } }
        };
        constexpr bool late = NORM;                                    // the weight stream starts behind the norm's fold (see gemv_chain_kernel)
        constexpr int EARLY = late ? (LNB_EARLY_QUAD < R ? LNB_EARLY_QUAD : R) : R;
#pragma unroll
        for (int j = 0; j < EARLY; j++) issue_next(buf[j]);
        x_store<NORM, NS>(p, xs, kpad, wave, lane, xv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        TIMED_BARRIER();                                               // B1: xs (or the squares) are in LDS
        if (NORM) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            rms_fold<rms_nf(NH)>(p, xs, ringB, hw < rms_nf(NH) ? hw : -1, lane, t_aux);      // X1, X2 inside; the first four helpers fold (four consecutive waves: four SIMDs)
            {                                                          // the walker walks now: the issue stall costs nothing here
#pragma unroll
                for (int j = EARLY; j < R; j++) issue_next(buf[j]);
            }
            TIMED_BARRIER();                                           // B2: r published
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if constexpr (NORM) x_normalize<NS>(p, xs, kpad, xs[kpad], wave, lane, xv, nv);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            TIMED_BARRIER();                                           // B3: xs normalised
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // unit of load i: q = kc*RW + r (stage-local k-chunk kc, row r); its eight products are the slots j0, j0+1 of group kc/2
        unsigned doff[NP]; int xoff[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int q = (i * NH + hw) * 64 + lane, kc = q / RW, r = q % RW;
            const int j0 = (kc & 1) * 2, sw = (r >> 1) & 3;
            doff[i] = (unsigned)((((kc >> 1) * RW + r) * 4 + (j0 ^ sw)) * 16);      // slot j0+1 sits at doff ^ 16 (j0 is even)
            xoff[i] = kc * 8;
        }
        int st = 0;                                        // stage-in-block of the stage being converted
        for (int it0 = 0; it0 <= T + 1; it0 += R) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int t = it0 + j;
                if (t <= T + 1) {
                    if (t < T) {
                        const long long tb_ = p.dbg_full ? clock64() : 0;
                        wait_ring<(R - 1) * NP, NP>(buf[j]);           // stage t landed; R-1 younger stages stay in flight
                        if (p.dbg_full) t_x += clock64() - tb_;
                        char* dst = ringB + (size_t)(t % GQ_SLOTS) * SB;
                        const float* xst = xs + (size_t)st * KS;
                        float4 xa[NP], xb[NP];
#pragma unroll
                        for (int i = 0; i < NP; i++) { xa[i] = *(const float4*)(xst + xoff[i]); xb[i] = *(const float4*)(xst + xoff[i] + 4); }
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            const u32x4 v = buf[j][i];
                            // exact products (8-bit x 8-bit significands): val1F32 * val2F32, operations_lineartransform.go:60
                            *(float4*)(dst + doff[i]) = mul4(xa[i], bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
                            *(float4*)(dst + (doff[i] ^ 16u)) = mul4(xb[i], bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
                        }
                        if (++st == nstages) st = 0;
                        __builtin_amdgcn_sched_barrier(0);             // refill AFTER the slot has been consumed (no register copies)
                        issue_next(buf[j]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // ds_writes complete before the barrier
                    TIMED_BARRIER();
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
        DBG_EXIT();
        return;
    }
    // ==================================== chain waves ==================================================
    {
        uint4 xv[XC], nv[XC];
        x_issue<NORM, NS>(p, xrow, wave, lane, xv, nv);
        LNB_STAMP(0);
        x_store<NORM, NS>(p, xs, kpad, wave, lane, xv);
        LNB_STAMP(1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        TIMED_BARRIER();                                               // B1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        LNB_STAMP(2);
        if (NORM) {
            float r;
            if (wave == 0) {
                r = rms_scale_wide<rms_nf(NH)>(p, xs, ringB, lane, t_aux);     // X1, X2 inside
                LNB_STAMP(3);
                if (lane == 0) xs[kpad] = r;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                TIMED_BARRIER();                                       // B2
                LNB_STAMP(4);
            } else {
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();     // X1, X2
                TIMED_BARRIER();                                       // B2
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                r = xs[kpad];
            }
            if constexpr (NORM) x_normalize<NS>(p, xs, kpad, r, wave, lane, xv, nv);
            LNB_STAMP(5);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            TIMED_BARRIER();                                           // B3
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            LNB_STAMP(6);
        }
    }
    if (p.dbg) { t_x = clock64() - t_begin; t_wait = 0; }              // (chain waves report the barrier wait of the main loop only)
    __builtin_amdgcn_s_setprio(3);
    constexpr int last_rows = RW - 16 * (NCW - 1);                     // rows of the last chain wave
    const int rows_here = wave == NCW - 1 ? last_rows : 16;
    const int rq = (lane >> 2) % rows_here, row = wave * 16 + rq, jq = lane & 3;
    const char* const src0 = ringB + (size_t)((row * 4 + (jq ^ ((row >> 1) & 3))) * 16);
    float acc[1] = {0.0f};
    float4 pq[4];
    int st = 0, blk = wg;
    for (int it = 0; it <= T + 1; it++) {
        if (it >= 2) {
            const char* cur = src0 + (size_t)((it - 2) % GQ_SLOTS) * SB;
            const char* nxt = src0 + (size_t)((it - 1) % GQ_SLOTS) * SB;       // complete since the last barrier (past the end: stale bytes, never added)
            if (it == 2) { pq[0] = *(const float4*)cur; pq[1] = *(const float4*)(cur + RW * 64); pq[2] = *(const float4*)(cur + 2 * RW * 64); }
#pragma unroll
            for (int g = 0; g < GS; g++) {
                pq[(g + 3) & 3] = g + 3 < GS ? *(const float4*)(cur + (g + 3) * (RW * 64)) : *(const float4*)(nxt + (g + 3 - GS) * (RW * 64));
                __builtin_amdgcn_sched_barrier(0);                     // the read is issued HERE, three groups ahead of its adds
                chain16q(acc[0], pq[g & 3]);                           // valDstF32 += p, k ascending (operations_lineartransform.go:63)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++st == nstages) {                                     // end of a row block: write it out, start the next
                gemv_epilogue<1, EPI, 4>(p, acc, m, blk * RW + row, jq == 0 && lane < rows_here * 4 && blk * RW + row < p.n_rows);
                acc[0] = 0.0f; st = 0; blk += p.n_wg;
            }
        }
        TIMED_BARRIER();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    DBG_EXIT();
}

// ------------------------------------------------------------------------------------------------
// Row-broadcast exact GEMV for THIN matrices (wo, w2: 16 output rows per CU are all there is to do).
//
// In gemv_chain_kernel a chain lane receives its products through the LDS: measured on gfx950 a ds_read_b128 occupies the
// wave for ~12.5 cycles, i.e. ~3.1 cycles per k-step on top of the 4.33-cycle dependent add, and the thin kernels are pure
// chain latency (K x ~7.6 cycles).  Here the products never leave the register file:
//   a wave owns 4 output rows, one per DPP row of 16 lanes; lane (q, j) streams the weights of row q with k = j (mod 16)
//   (16 B = eight of them, k ascending, per 128-step chunk), multiplies them by x (exact products) and every lane of the
//   row adds the 16 lanes' products in k order with  v_add_f32_dpp row_newbcast:j  -- a DPP operand costs nothing measurable
//   (8.75 vs 8.63 ticks per dependent add in tools/chainbench2.hip), so a k-step is ~4.33 cycles + ~0.8 of producer work.
// All 16 lanes of a row compute the same sum (same operands, same order); lane j == 0 stores it.
// Four waves (16 rows) per workgroup share x (f32, transposed per chunk in LDS); weights come through the same hand-counted
// register ring as gemv_chain_kernel (RING_LOAD / RING_RETIRE, tools/isa_audit.py).
// grid.x = S * n_wg, block = 256; dynamic LDS = K * 4 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int RC_R = 14;                                    // 1 KiB chunks in flight per wave (56 KiB per CU)
// 16 dependent adds in ONE asm statement: acc += pr(lane 0) ... += pr(lane 15) of this lane's DPP row.  (One statement per add
// made hipcc's hazard recognizer put an s_nop behind every one of them -- it cannot see which operand is the DPP one -- and
// doubled the step time.  Inside the block no further wait states are needed: acc is the plain operand.)
// the 128 dependent adds of one chunk in ONE asm statement: acc += pr[i](lane 0) ... += pr[i](lane 15), i = 0..7.  (Separate
// statements make hipcc's hazard recognizer put an s_nop behind each -- it cannot see which operand is the DPP one.)
// VALU write of pr -> DPP read needs 2 wait states: the leading s_nop covers whatever hipcc scheduled last.
DEVINL void chain128(float& acc, const float (&pr)[8]) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %5, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %6, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %7, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %8, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(acc) : "v"(pr[0]), "v"(pr[1]), "v"(pr[2]), "v"(pr[3]), "v"(pr[4]), "v"(pr[5]), "v"(pr[6]), "v"(pr[7]));
}
template <int EPI>
__global__ __launch_bounds__(256) void rowcast_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xT = (float*)smem;                                // x[128c + 16i + j] at xT[(c*16 + j)*8 + i]
    const long long t_begin = p.dbg ? clock64() : 0;         // LNB_GEMV_TIMING: [wg][wave] = {total, -, prologue, -}
    const long long t_wall0 = p.dbg ? (long long)wall_clock64() : 0;
    long long t_x = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K, S = p.S, nchunks = K >> 7;
    const int m = (S == 1) ? 0 : (int)(blockIdx.x % (unsigned)S);
    const int wg = (S == 1) ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned)S);
    const int nb_mine = (p.n_blocks - wg + p.n_wg - 1) / p.n_wg;       // 16-row blocks wg, wg+n_wg, ...
    const int T = nb_mine * nchunks;                                   // chunks this wave walks
    const size_t tile_bytes = (size_t)nchunks * 1024;                  // one wave tile (4 rows)
    const unsigned voff = (unsigned)lane * 16u;
    // x first, all of its loads in flight at once, and the weight stream only once x has landed: queued in front of x -- or
    // even right behind it -- the 14 MB burst of first weight loads delays the x rows that every CU reads (measured both ways)
    constexpr int RC_XU = 8;                                 // 16 B units of x per thread: K <= 16384
    const uint16_t* xrow = p.x + (size_t)m * K;
    uint4 xv[RC_XU];
#pragma unroll
    for (int i = 0; i < RC_XU; i++) {
        const int u = tid + i * 256;
        xv[i] = *(const uint4*)(xrow + (size_t)(u < (K >> 3) ? u : 0) * 8);      // unconditional (clamped) loads
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): x has landed (hipcc then needs no wait of its own below, which
                                                             // would also drain the weight loads that are issued next)
    u32x4 buf[RC_R];
    // issue cursor: a running wave-uniform pointer (1 KiB per chunk; at the end of a tile jump to this wave's tile of the
    // next block); past the last chunk it stays put and re-reads
    const char* sb = (const char*)p.w + ((size_t)wg * 4 + wave) * tile_bytes;
    const size_t blk_jump = (size_t)p.n_wg * 4 * tile_bytes - tile_bytes;                                  // from the end of a tile
    int ic = 0, issued = 0;
    auto issue_next = [&](u32x4& dst) {
        ld_nt_asm(dst, voff, sb);
        if (issued + 1 < T) { issued++; sb += 1024; if (++ic == nchunks) { ic = 0; sb += blk_jump; } }
    };
#pragma unroll
    for (int j = 0; j < RC_R; j++) issue_next(buf[j]);
#pragma unroll
    for (int i = 0; i < RC_XU; i++) {
        const int u = tid + i * 256;
        if (u < (K >> 3)) {
            const uint4 v = xv[i];
            const int k = u * 8, c = k >> 7, ii = (k & 127) >> 4, j0 = k & 15;
            float* d = xT + ((size_t)(c * 16 + j0) * 8 + ii);
            d[0] = bf_lo(v.x); d[8] = bf_hi(v.x); d[16] = bf_lo(v.y); d[24] = bf_hi(v.y);
            d[32] = bf_lo(v.z); d[40] = bf_hi(v.z); d[48] = bf_lo(v.w); d[56] = bf_hi(v.w);
        }
    }
    __syncthreads();
    if (p.dbg) t_x = clock64() - t_begin;
    LNB_STAMP(6);
    if (p.prio) __builtin_amdgcn_s_setprio(3);               // (rung (a) of the FFN ladder: what a prioritised w2 chain keeps beside a co-resident gate|up workgroup)
    const float* xl = xT + (size_t)(lane & 15) * 8;
    float acc = 0.0f;
    float pr[8];
    auto products = [&](float (&d)[8], const u32x4& v, const float4& xa, const float4& xb) {   // exact: 8-bit x 8-bit significands
        const float4 pa = mul4(xa, bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)), pb = mul4(xb, bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
        d[0] = pa.x; d[1] = pa.y; d[2] = pa.z; d[3] = pa.w; d[4] = pb.x; d[5] = pb.y; d[6] = pb.z; d[7] = pb.w;
    };
    int c = 0, blk = wg;
    // the x values of a chunk are read from the LDS one chunk ahead, in the shadow of the previous chunk's 128 adds
    float4 xa = *(const float4*)(xl), xb = *(const float4*)(xl + 4);
    for (int t0 = 0; t0 < T; t0 += RC_R) {
#pragma unroll
        for (int j = 0; j < RC_R; j++) {
            if (t0 + j < T) {
                // chunk t0+j landed; RC_R-1 younger ones stay in flight.  (The wait names buf[j] itself: handing wait_ring a cast
                // reference made hipcc copy the register BEFORE the wait -- caught by tools/isa_audit.py.)
                asm volatile("s_waitcnt vmcnt(%1) ; RING_RETIRE %0" : "+v"(buf[j]) : "n"(RC_R - 1) : "memory");
                products(pr, buf[j], xa, xb);
                // pin the products in front of the refill: otherwise hipcc sinks the unpack/multiply into the chain below, behind
                // the asm that reloads buf[j], and keeps the old value alive through a register copy made BEFORE the wait
                asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
                issue_next(buf[j]);
                {
                    const int cn = (c + 1 == nchunks) ? 0 : c + 1;     // next chunk of this row (or chunk 0 of the next block's)
                    xa = *(const float4*)(xl + (size_t)cn * 128); xb = *(const float4*)(xl + (size_t)cn * 128 + 4);
                }
                __builtin_amdgcn_sched_barrier(0);                     // the reads are issued HERE, waited for after the chain
                chain128(acc, pr);                                     // valDstF32 += p, k ascending (operations_lineartransform.go:63)
                if (++c == nchunks) {                                  // end of this wave's 4 rows
                    const int n = blk * 16 + wave * 4 + (lane >> 4);
                    if ((lane & 15) == 0 && n < p.n_rows) {
                        const size_t o = (size_t)m * p.n_rows + n;
                        if (EPI == EPI_RESID) p.out[o] = bf_trunc(bf_wide(p.res[o]) + bf_wide(bf_trunc(acc)));   // ml.Add, operations_impl.go:320-332
                        else p.out[o] = bf_trunc(acc);
                    }
                    acc = 0.0f; c = 0; blk += p.n_wg;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
    if (p.dbg && lane == 0) { long long* d_ = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4; d_[0] = clock64() - t_begin; d_[1] = 0; d_[2] = t_x; d_[3] = t_x; }
    LNB_WALL_EXIT();
}

// ------------------------------------------------------------------------------------------------
// rowcast_lds_kernel (round 4): the row-broadcast chain with its PRODUCER work moved to a second wave.
//
// tools/chainbench3.hip: a dependent v_add_f32_dpp row_newbcast costs 5.3 cycles when nothing else is issued between the adds,
// and whatever the chain wave issues besides them is paid in full (a wave issues one instruction per ~4.4 cycles): rowcast_kernel's
// chain wave also unpacks and multiplies its weights, reads x and refills its register ring -- 6.2-6.6 cycles per step.  Every
// way of handing a lane its OWN operand per step is worse (ds_read_b128 per 4 steps: 9.0 cycles per step with 64, 32 or 16 lanes
// active -- the returning data occupies the SIMD's register write port like four VALU ops; products made by the matrix pipe with a
// one-hot selector operand: 7.5), but a DPP operand serves 16 lanes, so here one ds_read_b128 feeds 64 steps: 5.56 measured.
//   waves 0..3  chain : 4 rows each, one per DPP row; per 128-step chunk two ds_read_b128 (the next chunk's products) + 128 adds;
//   waves 4..7  helper: wave 4+c streams the weights of chain wave c (same lane mapping as rowcast_kernel: lane (q, j) holds the
//                       eight weights of row q with k = j mod 16), multiplies them by x (exact products) and leaves them in a
//                       lane-private LDS ring -- wave w and w+4 share a SIMD, the helper is the younger wave there and only takes
//                       the issue slots the chain's DPP latency leaves free (NOTES.md 5.8: 5.3 cycles per step beside a younger
//                       vector-heavy wave);
// stages of RL_SC chunks in a three-slot ring, one s_barrier per stage: at iteration t the helpers write stage t while the chain
// waves add stage t-2 and prefetch the first chunk of stage t-1 (complete since the previous barrier), so no LDS round trip is
// exposed at a stage boundary.  x lives in the LDS as raw bf16 (K * 2 bytes), transposed per chunk like the weights.
// Same weight layout as rowcast_kernel (tag RW 4); requires K % (128 * RL_SC) == 0 and at least two stages (the launcher falls back otherwise).
// grid.x = S * n_wg, block = 512, dynamic LDS = rl_lds_bytes(K).
// ------------------------------------------------------------------------------------------------
constexpr int RL_SC = 4;                                    // 128-step chunks per stage
constexpr int RL_SLOTS = 3;
constexpr int RL_R = 12;                                    // 1 KiB weight chunks in flight per helper wave (a multiple of RL_SC * RL_SLOTS' unroll)
constexpr int RL_STAGE = 4 * RL_SC * 2048;                  // bytes of products per stage: 4 pairs x RL_SC chunks x 2 KiB
__host__ __device__ constexpr size_t rl_lds_bytes(int K) { return (size_t)RL_SLOTS * RL_STAGE + (size_t)K * 2; }
template <int EPI>
__global__ __launch_bounds__(512) void rowcast_lds_kernel(GemvParams p) {
    static_assert(RL_R == RL_SC * RL_SLOTS, "the helper's ring index and the product slot are static inside a three-stage unroll");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    uint16_t* xb = (uint16_t*)(smem + RL_SLOTS * RL_STAGE);  // x[128c + 16i + j] at xb[(c*16 + j)*8 + i]
    const long long t_begin = p.dbg ? clock64() : 0;         // LNB_GEMV_TIMING: [wg][wave] = {total, barrier wait, x staged, chain start (chain waves)}
    const long long t_wall0 = p.dbg ? (long long)wall_clock64() : 0;
    long long t_x = 0, t_wait = 0, t_chain = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), pair = wave & 3;
    const int K = p.K, S = p.S, nchunks = K >> 7, nst = nchunks / RL_SC;
    const int m = (S == 1) ? 0 : (int)(blockIdx.x % (unsigned)S);
    const int wg = (S == 1) ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned)S);
    const int nb_mine = (p.n_blocks - wg + p.n_wg - 1) / p.n_wg;       // 16-row blocks wg, wg+n_wg, ...
    const int NS = nb_mine * nst;                                      // stages this workgroup walks
    // x first, all of its loads in flight at once; the weight stream starts once x has landed (NOTES.md 5.1)
    constexpr int RL_XU = 4;                                 // 16 B units of x per thread: K <= 16384
    const uint16_t* xrow = p.x + (size_t)m * K;
    uint4 xv[RL_XU];
#pragma unroll
    for (int i = 0; i < RL_XU; i++) {
        const int u = tid + i * 512;
        xv[i] = *(const uint4*)(xrow + (size_t)(u < (K >> 3) ? u : 0) * 8);      // unconditional (clamped) loads
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): hipcc then needs no wait of its own below (it would drain the weight ring too)
    u32x4 buf[RL_R];
    const size_t tile_bytes = (size_t)nchunks * 1024;        // one wave tile (4 rows)
    const char* sb = (const char*)p.w + ((size_t)wg * 4 + pair) * tile_bytes;
    const size_t blk_jump = (size_t)p.n_wg * 4 * tile_bytes - tile_bytes;
    const unsigned voff = (unsigned)lane * 16u;
    const int T = NS * RL_SC;
    int ic = 0, issued = 0;
    auto stage_x = [&]() {
#pragma unroll
        for (int i = 0; i < RL_XU; i++) {
            const int u = tid + i * 512;
            if (u < (K >> 3)) {
                const uint4 v = xv[i];
                const int k = u * 8, c = k >> 7, ii = (k & 127) >> 4, j0 = k & 15;
                uint16_t* d = xb + ((size_t)(c * 16 + j0) * 8 + ii);
                d[0] = (uint16_t)v.x; d[8] = (uint16_t)(v.x >> 16); d[16] = (uint16_t)v.y; d[24] = (uint16_t)(v.y >> 16);
                d[32] = (uint16_t)v.z; d[40] = (uint16_t)(v.z >> 16); d[48] = (uint16_t)v.w; d[56] = (uint16_t)(v.w >> 16);
            }
        }
        __syncthreads();
        if (p.dbg) t_x = clock64() - t_begin;
    };
#define RL_BARRIER() do { if (p.dbg_full) { const long long tb_ = clock64(); __builtin_amdgcn_s_barrier(); t_wait += clock64() - tb_; } else __builtin_amdgcn_s_barrier(); } while (0)
    // Schedule (iterations it = 0 .. NS, one s_barrier each).  Iteration 0 fills the ring from BOTH sides: the helper makes stage 0, the
    // chain wave -- idle until there is something to add -- makes stage 1 from four weight loads of its own (otherwise it sits through
    // two helper stages, ~4 k cycles per launch measured).  From then on: iteration it, the helper writes stage it+1 (slot (it+1) % 3),
    // the chain wave adds stage it-1 (slot (it-1) % 3) and prefetches the first chunk of stage it (complete since the previous barrier),
    // so no LDS round trip is exposed at a stage boundary.
    auto products = [&](float4& pa, float4& pb, const u32x4& b, const uint4& xq) {      // exact: 8-bit x 8-bit significands (operations_lineartransform.go:60)
        pa = mul4(make_float4(bf_lo(xq.x), bf_hi(xq.x), bf_lo(xq.y), bf_hi(xq.y)), bf_lo(b.x), bf_hi(b.x), bf_lo(b.y), bf_hi(b.y));
        pb = mul4(make_float4(bf_lo(xq.z), bf_hi(xq.z), bf_lo(xq.w), bf_hi(xq.w)), bf_lo(b.z), bf_hi(b.z), bf_lo(b.w), bf_hi(b.w));
    };
    const char* xl = (const char*)xb + (size_t)(lane & 15) * 16;
    if (wave >= 4) {
        // ================================ helper: stage 0, then stages 2, 3, ... of chain wave `pair` ===================
        // (the two roles are separate straight-line paths from here on: tools/isa_audit.py follows each role's ring registers on its own path)
        const int Th = T - RL_SC;                            // loads of this wave: every chunk but the four of stage 1
        auto advance = [&]() { sb += 1024; if (++ic == nchunks) { ic = 0; sb += blk_jump; } };
        auto issue_h = [&](u32x4& dst) {
            ld_nt_asm(dst, voff, sb);
            if (issued + 1 < Th) { issued++; advance(); if (issued == RL_SC) { for (int j = 0; j < RL_SC; j++) advance(); } }   // past the last chunk: stays put, re-reads
        };
        // the first stage's loads in front of the x scatter, the rest of the ring behind it: twelve loads per wave issued into the cold
        // memory pipeline stall the issuing wave, and with it the scatter every wave waits for (prologue 2.9 k -> 4.4 k cycles measured)
#pragma unroll
        for (int j = 0; j < RL_SC; j++) issue_h(buf[j]);
        stage_x();
#pragma unroll
        for (int j = RL_SC; j < RL_R; j++) issue_h(buf[j]);
        char* const dst0 = ring + (size_t)pair * (RL_SC * 2048) + (size_t)lane * 16;
        int c = 0;
        for (int it0 = 0; it0 <= NS; it0 += RL_SLOTS) {
#pragma unroll
            for (int u = 0; u < RL_SLOTS; u++) {
                const int it = it0 + u;
                if (it <= NS) {
                    if (it == 0 || it + 1 < NS) {
                        char* dst = dst0 + (size_t)(it == 0 ? 0 : (u + 1) % RL_SLOTS) * RL_STAGE;
#pragma unroll
                        for (int cc = 0; cc < RL_SC; cc++) {
                            u32x4& b = buf[u * RL_SC + cc];
                            asm volatile("s_waitcnt vmcnt(%1) ; RING_RETIRE %0" : "+v"(b) : "n"(RL_R - 1) : "memory");
                            const uint4 xq = *(const uint4*)(xl + (size_t)c * 256);
                            float4 pa, pb;
                            products(pa, pb, b, xq);
                            // pin the products in front of the refill (else hipcc sinks the unpack behind the asm that reloads b)
                            asm volatile("" : "+v"(pa.x), "+v"(pa.y), "+v"(pa.z), "+v"(pa.w), "+v"(pb.x), "+v"(pb.y), "+v"(pb.z), "+v"(pb.w));
                            issue_h(b);
                            *(float4*)(dst + cc * 2048) = pa; *(float4*)(dst + cc * 2048 + 1024) = pb;
                            if (++c == nchunks) c = 0;
                        }
                        if (it == 0) c = (2 * RL_SC) % nchunks;            // stage 1 is the chain wave's
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // ds_writes complete before the barrier
                    RL_BARRIER();
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
    } else {
        // ================================ chain wave: 4 rows, one per DPP row =========================================
        u32x4 cb[RL_SC];
#pragma unroll
        for (int j = 0; j < RL_SC; j++) ld_nt_asm(cb[j], voff, sb + (size_t)(RL_SC + j) * 1024);      // stage 1: chunks 4 .. 7 of this wave's first tile (K >= 1024)
        stage_x();
        const char* const src0 = ring + (size_t)pair * (RL_SC * 2048) + (size_t)lane * 16;
        {   // iteration 0: stage 1 into slot 1
            char* dst = ring + (size_t)pair * (RL_SC * 2048) + (size_t)lane * 16 + (size_t)1 * RL_STAGE;
#pragma unroll
            for (int cc = 0; cc < RL_SC; cc++) {
                if (cc == 0) asm volatile("s_waitcnt vmcnt(3) ; RING_RETIRE %0" : "+v"(cb[0]) :: "memory");
                if (cc == 1) asm volatile("s_waitcnt vmcnt(2) ; RING_RETIRE %0" : "+v"(cb[1]) :: "memory");
                if (cc == 2) asm volatile("s_waitcnt vmcnt(1) ; RING_RETIRE %0" : "+v"(cb[2]) :: "memory");
                if (cc == 3) asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE %0" : "+v"(cb[3]) :: "memory");
                const uint4 xq = *(const uint4*)(xl + (size_t)(RL_SC + cc) * 256);
                float4 pa, pb;
                products(pa, pb, cb[cc], xq);
                *(float4*)(dst + cc * 2048) = pa; *(float4*)(dst + cc * 2048 + 1024) = pb;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            RL_BARRIER();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __builtin_amdgcn_s_setprio(3);
        if (p.dbg) { t_chain = clock64() - t_begin; t_wait = 0; }         // (chain waves report the barrier wait of the main loop only)
        LNB_STAMP(6);
        float acc = 0.0f;
        float4 ba[2], bb[2];                                 // products of the chunk being added / of the next one
        ba[0] = *(const float4*)src0; bb[0] = *(const float4*)(src0 + 1024);     // stage 0, chunk 0
        int sdone = 0, blk = wg;
        for (int it0 = 0; it0 <= NS; it0 += RL_SLOTS) {
#pragma unroll
            for (int u = 0; u < RL_SLOTS; u++) {
                const int it = it0 + u;
                if (it >= 1 && it <= NS) {
                    // stage it-1 lives in slot (u+2) % 3, stage it (complete since the last barrier) in slot u
                    const char* cur = src0 + (size_t)((u + 2) % RL_SLOTS) * RL_STAGE;
                    const char* nxt = src0 + (size_t)u * RL_STAGE;
#pragma unroll
                    for (int cc = 0; cc < RL_SC; cc++) {
                        const char* q = cc + 1 < RL_SC ? cur + (cc + 1) * 2048 : nxt;      // past the last stage: stale bytes, never added
                        ba[(cc + 1) & 1] = *(const float4*)q; bb[(cc + 1) & 1] = *(const float4*)(q + 1024);
                        __builtin_amdgcn_sched_barrier(0);         // the reads are issued HERE, in front of the chunk's 128 adds
                        const float4 a = ba[cc & 1], b = bb[cc & 1];
                        const float pr[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        chain128(acc, pr);                         // valDstF32 += p, k ascending (operations_lineartransform.go:63)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (++sdone == nst) {                          // end of this wave's 4 rows
                        const int n = blk * 16 + pair * 4 + (lane >> 4);
                        if ((lane & 15) == 0 && n < p.n_rows) {
                            const size_t o = (size_t)m * p.n_rows + n;
                            if (EPI == EPI_RESID) p.out[o] = bf_trunc(bf_wide(p.res[o]) + bf_wide(bf_trunc(acc)));   // ml.Add, operations_impl.go:320-332
                            else p.out[o] = bf_trunc(acc);
                        }
                        acc = 0.0f; sdone = 0; blk += p.n_wg;
                    }
                    RL_BARRIER();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
            }
        }
    }
#undef RL_BARRIER
    if (p.dbg && lane == 0) { long long* d_ = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4; d_[0] = clock64() - t_begin; d_[1] = t_wait; d_[2] = t_x; d_[3] = t_chain; }
    LNB_WALL_EXIT();
}

// ------------------------------------------------------------------------------------------------
// Prefill (S >= 16 rows per call): the same exact chains on the f32 MATRIX cores.
//
// v_mfma_f32_16x16x4_f32 is, bit for bit, the k-ordered chain  D = fma(a_k3,b_k3, fma(a_k2,b_k2, fma(a_k1,b_k1, fma(a_k0,b_k0, C))))
// (one rounding per product, no wider internal accumulation; verified against the reference's loop on MI355X by
// tools/mfma_exact.hip), and bf16 x bf16 products are exact, so fma == the reference's multiply-then-add.  One instruction
// advances 16 output rows x 16 batch rows by four k-steps in 32 cycles: ~20x the chain throughput of the S = 1 kernels,
// with identical bits.  (It does not help decode: one batch column uses 1/16 of the instruction, 10 cycles per k-step.)
//
// rmsnorm_rows_kernel: RMSNorm of S rows, one wave per row (the rows are the parallelism here; each wave walks its row's
// sequential sum of squares out of the LDS), output bf16 [S][K] with the reference's two truncations.
// gemm_mfma_kernel: workgroup = 4 waves = 64 output rows x 128 batch rows; K is walked in 128-step slabs staged through the
// LDS as f32, k-major and padded so that both operand fragments are bank-conflict free:
//   A (weights): lane (i = l&15, kk = l>>4) reads As[4g+kk][16w+i];  B (x): Bs[4g+kk][16t+j]   (row strides 80 / 144 floats)
// wave w owns n-tile w and all 8 m-tiles: 8 (16 for the two-chain w1|w3) MFMAs per 1-2 A reads + 8 B reads.
// Weights are gathered 16 B at a time from whichever tiled layout the matrix was stored in (tiled_index).
// grid = (ceil(n_rows/64), ceil(S/128)); dynamic LDS = (NCH*128*80 + 128*144) * 4 bytes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rmsnorm_rows_kernel(const uint16_t* x, const uint16_t* w, uint16_t* out, int S, int K, float eps) {
    // one wave per row: the row's squares go to the LDS with coalesced loads, every lane then walks the SAME sequential sum
    // (broadcast float4 reads, the next 16 values in flight behind 16 adds), and the lanes share the normalisation
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sq = (float*)smem;                                // K (+32 zero pad) f32
    const int m = blockIdx.x, lane = threadIdx.x;
    const uint16_t* xr = x + (size_t)m * K;
    for (int k = lane * 8; k < K + 32; k += 512) {           // Pow(x,2): exact in f32 (impl:197-217)
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) v = *(const uint4*)(xr + k);
        const float a0 = bf_lo(v.x), a1 = bf_hi(v.x), a2 = bf_lo(v.y), a3 = bf_hi(v.y), a4 = bf_lo(v.z), a5 = bf_hi(v.z), a6 = bf_lo(v.w), a7 = bf_hi(v.w);
        if (k < K + 32) { *(float4*)(sq + k) = make_float4(a0 * a0, a1 * a1, a2 * a2, a3 * a3); *(float4*)(sq + k + 4) = make_float4(a4 * a4, a5 * a5, a6 * a6, a7 * a7); }
    }
    __syncthreads();
    float sum = 0.0f;                                        // Mean's serial f32 sum, k ascending (impl:236-251); + 0.0 past K changes nothing
    float4 a0 = *(const float4*)(sq), a1 = *(const float4*)(sq + 4), a2 = *(const float4*)(sq + 8), a3 = *(const float4*)(sq + 12);
    for (int k0 = 0; k0 < K; k0 += 32) {
        const float4 b0 = *(const float4*)(sq + k0 + 16), b1 = *(const float4*)(sq + k0 + 20), b2 = *(const float4*)(sq + k0 + 24), b3 = *(const float4*)(sq + k0 + 28);
        __builtin_amdgcn_sched_barrier(0);
        touch16(a0, a1, a2, a3);
        __builtin_amdgcn_sched_barrier(0);
        sum = add4(sum, a0); sum = add4(sum, a1); sum = add4(sum, a2); sum = add4(sum, a3);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 32 < K) { a0 = *(const float4*)(sq + k0 + 32); a1 = *(const float4*)(sq + k0 + 36); a2 = *(const float4*)(sq + k0 + 40); a3 = *(const float4*)(sq + k0 + 44); }
        __builtin_amdgcn_sched_barrier(0);
        touch16(b0, b1, b2, b3);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 16 < K) { sum = add4(sum, b0); sum = add4(sum, b1); sum = add4(sum, b2); sum = add4(sum, b3); }
        __builtin_amdgcn_sched_barrier(0);
    }
    float mean = __fdiv_rn(sum, (float)K);
    mean = mean + eps;
    const float r = (float)(1.0 / sqrt((double)mean));
    uint16_t* o = out + (size_t)m * K;
    for (int k = lane * 8; k < K; k += 512) {                // trunc(trunc(x*r)*w) (llamatransformer.go:656,638)
        const uint4 v = *(const uint4*)(xr + k), g = *(const uint4*)(w + k);
        const uint32_t xs_[4] = {v.x, v.y, v.z, v.w}, ws_[4] = {g.x, g.y, g.z, g.w};
        uint32_t r4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint16_t lo = bf_trunc(bf_wide(bf_trunc(bf_lo(xs_[e]) * r)) * bf_lo(ws_[e]));
            const uint16_t hi = bf_trunc(bf_wide(bf_trunc(bf_hi(xs_[e]) * r)) * bf_hi(ws_[e]));
            r4[e] = (uint32_t)lo | ((uint32_t)hi << 16);
        }
        *(uint4*)(o + k) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GM_NB = 64, GM_KS = 128;
constexpr unsigned GM_MB64_UPTO = 0xFFFFFFFFu;              // 64-row batch tiles up to this many 64x128 tiles: always -- tools/gemmbench.hip, S = 256..2048:
                                                            // 64 x 64 tiles are 2-28 % faster than 64 x 128 everywhere (more workgroups per CU to overlap staging)
template <int EPI> DEVINL void gemm_epilogue4(const GemmParams& p, const f32x4& g, const f32x4& u, int m, int n0) {
    // one lane: batch row m, four consecutive output rows n0..n0+3 (n0 % 4 == 0)
    if (m >= p.S) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int n = n0 + r;
        if (n >= p.n_rows) continue;
        const size_t o = (size_t)m * p.n_rows + n;
        if (EPI == EPI_STORE) p.out[o] = bf_trunc(g[r]);
        else if (EPI == EPI_RESID) p.out[o] = bf_trunc(bf_wide(p.res[o]) + bf_wide(bf_trunc(g[r])));            // ml.Add, impl:320-332
        else if (EPI == EPI_SILU_MUL) {                                                                           // activations.go:36-39, :614
            const uint16_t gs = bf_trunc(p.silu[bf_trunc(g[r])]);
            const uint16_t y = bf_trunc(bf_wide(gs) * bf_wide(bf_trunc(u[r])));
            if (p.out_xt) p.out_xt[xt_group(m, p.n_rows) + xt_index(m & 15, n)] = y;       // the next product reads column groups (mfma_pair_kernel)
            else p.out[o] = y;
        } else if (EPI == EPI_QKV_ROPE) {                                                                         // llamatransformer.go:297-403
            // rows of one sequence: consecutive positions of its cache; rows of a batch (btab): row m is sequence m's one new token
            const int pos = p.btab ? p.btab->st[m]->pos : p.st->pos + m;
            const int seq_len = p.btab ? p.btab->seq_len[m] : p.seq_len;
            uint16_t* const cache_k = p.btab ? p.bkv->ck[m] : p.cache_k;
            uint16_t* const cache_v = p.btab ? p.bkv->cv[m] : p.cache_v;
            const uint16_t mine = bf_trunc(g[r]), other = bf_trunc(g[r ^ 1]);                                      // RoPE partner 2i <-> 2i+1: same lane
            if (n < p.q_dim + p.kv_dim) {
                const int d = n % p.head_dim, i = d >> 1;
                const float2 cs = *(const float2*)(p.cis + ((size_t)pos * (p.head_dim >> 1) + i) * 2);
                const double cr = (double)cs.x, ci = (double)cs.y;
                uint16_t r16;
                if ((n & 1) == 0) { const double a = (double)bf_wide(mine), bb = (double)bf_wide(other); r16 = bf_trunc((float)(a * cr - bb * ci)); }
                else              { const double a = (double)bf_wide(other), bb = (double)bf_wide(mine); r16 = bf_trunc((float)(a * ci + bb * cr)); }
                if (n < p.q_dim) p.q_out[(size_t)m * p.q_dim + n] = r16;
                else {
                    const int kc = n - p.q_dim, kh = kc / p.head_dim;
                    cache_k[(((size_t)kh * (p.head_dim >> 3) + (d >> 3)) * seq_len + pos) * 8 + (d & 7)] = r16;
                }
            } else cache_v[(size_t)pos * p.kv_dim + (n - p.q_dim - p.kv_dim)] = mine;
        }
    }
}

// WN = waves along n: 4 -> the workgroup owns 64 output rows, every wave all 8 m-tiles; 1 -> the workgroup owns 16 output rows and
// its four waves split the 8 m-tiles (4x the workgroups for thin matrices at small S, where 64-row tiles fill a quarter of the CUs)
// LDS operand tiles of gemm_mfma_kernel, as raw bf16 in per-lane STREAMS: the MFMA lane (i = l&15, kk = l>>4) of k-group g consumes
// k = 4g + kk, so everything that lane will read during a 128-step slab is laid out contiguously for it:
//   A (weights): stream (chain c, n-tile w, kk, i) = its 32 values, g ascending (64 B + 16 B pad): 4 ds_read_b128 per slab;
//   B (x):       stream (kk, i) = [g/2][m-tile t][g&1] (512 B + 16 B pad): one ds_read_b128 = four m-tiles of two k-groups.
// (An f32 k-major tile cost nine ds_reads per eight MFMAs and an LDS round trip in front of every pair; raw bf16 halves the LDS,
// two workgroups fit a CU, and a staged 16 B unit -- 8 consecutive k = two k-groups -- lands as four packed words.)
constexpr int GM_APAD = 80;                                 // stream strides in bytes: conflict-free 16 B reads across 16 lanes
// The one-chain 64-row tiles (WN 4, NCH 1) keep x in the LDS as f32 (bf16 << 16, converted once per element while staging): the B
// operand -- each value feeds ONE MFMA there -- then needs no widening op between matrix instructions (+6 % at S = 512).  The
// two-chain w1|w3 kernel uses every B value twice and is faster with the compact tile (two workgroups per CU at 64 x 128); the
// 16-row tiles (WN 1) keep raw bf16 pairs too.
__host__ __device__ constexpr bool gemm_bf32(int NCH, int WN) { return WN == 4 && NCH == 1; }
__host__ __device__ constexpr int gemm_bpad(int MB, bool f32 = false) { return (f32 ? 32 : 16) * (MB / 16) * 4 + 16; }   // stream bytes (+ pad)
__host__ __device__ constexpr size_t gemm_lds_a(int NCH, int WN) { return (size_t)NCH * WN * 64 * GM_APAD; }
__host__ __device__ constexpr size_t gemm_lds_bytes(int NCH, int WN, int MB = 128) { return gemm_lds_a(NCH, WN) + (size_t)64 * gemm_bpad(MB, gemm_bf32(NCH, WN)); }

#ifndef GM_DBG
#define GM_DBG 0                                            // tools/gemmbench.hip: 1 = stage only the first slab, 2 = no barriers, 4 = no LDS reads
#endif
// MB = batch rows per workgroup: 128, or 64 (WN = 4 only) when 128-row tiles would put at most one workgroup on a CU
template <int EPI, int NCH, int WN, int MB = 128>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmParams p) {
    static_assert(MB == 128 || (MB == 64 && WN == 4), "batch tile");
    constexpr bool BF32 = gemm_bf32(NCH, WN);                // x tile as f32: stream (kk, i) = [g][m-tile t]
    constexpr int GM_MB = MB, GM_BPAD = gemm_bpad(MB, BF32), TB = MB / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                         // streams [(c*WN + w)*4 + kk][i] of GM_APAD bytes
    char* Bs = smem + gemm_lds_a(NCH, WN);                   // streams [kk][i] of GM_BPAD bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NB = 16 * WN, MT = WN == 4 ? TB : 2;      // output rows per workgroup, m-tiles per wave
    const int n0 = blockIdx.x * NB, m0 = blockIdx.y * GM_MB;
    const int nt_off = WN == 4 ? wave * 16 : 0, mt0 = WN == 4 ? 0 : wave * 2;
    const int K = p.K;
    f32x4 acc[NCH][MT];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fk = lane >> 4;
    // staging registers: the NEXT slab's 16 B units are loaded before the current slab's MFMAs and written to the LDS after them
    // (a load-use loop paid one memory round trip per unit).  Unit q of a thread: weights = row tid % NB, unit kc = tid / NB + q*256/NB
    // of the slab; x = batch row tid % 128, unit tid / 128 + 2q.  Their addresses are a per-thread base + uniform strides (both tiled
    // layouts are affine in the slab, unit and chain index), so a full slab is staged without a compare, a select or a divergent
    // branch; only the last slab of a K that is not a multiple of 128 takes the checked path.
    constexpr int WU = NB * (GM_KS / 8) / 256, XU = GM_MB * (GM_KS / 8) / 256, WQ = 256 / NB, XQ = 256 / GM_MB;   // 4 (or 1) and 8 (4) units per thread
    uint4 wreg[NCH][WU], xreg[XU];
    const bool rw4 = p.rw == 4;                              // row-broadcast layout: a 16 B unit holds k0 + 16e + kc, e = 0..7
    const int kst = rw4 ? 16 : 1;                            // k stride inside a 16 B weight unit
    const int wrow = tid & (NB - 1), wkc = tid / NB, xrow = tid & (GM_MB - 1), xkc = tid / GM_MB;
    int wn = n0 + wrow; wn = wn < p.n_rows ? wn : p.n_rows - 1;                             // clamped rows are computed and dropped
    int xm = m0 + xrow; xm = xm < p.S ? xm : p.S - 1;
    const uint16_t* wp = p.w + tiled_index(wn, rw4 ? wkc : 8 * wkc, 0, K, p.rw, p.nch);     // slab 0, unit q = 0, chain 0
    const uint16_t* xp = p.x + (size_t)xm * K + 8 * xkc;
    const size_t w_slab = rw4 ? (size_t)512 : (size_t)16 * p.nch * p.rw * 8;                // elements per 128 k
    const size_t w_unit = rw4 ? (size_t)WQ * 8 : (size_t)WQ * p.nch * p.rw * 8;             // ... per unit step q
    const size_t w_chain = (size_t)p.rw * 8;                                                // ... per chain (chain layouts only)
    auto issue = [&](int k0) {
        if (k0 + GM_KS <= K) {                               // full slab (uniform branch)
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int q = 0; q < WU; q++) wreg[c][q] = *(const uint4*)(wp + c * w_chain + q * w_unit);
#pragma unroll
            for (int q = 0; q < XU; q++) xreg[q] = *(const uint4*)(xp + 8 * XQ * q);
            wp += w_slab; xp += GM_KS;
            return;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int q = 0; q < WU; q++) {
                const int kc = wkc + q * WQ;
                int kf = rw4 ? k0 + kc : k0 + 8 * kc;
                kf = kf + 7 * kst < K ? kf : 0;              // (zeroed past K in commit: no use here, or the load would be waited for at once)
                wreg[c][q] = *(const uint4*)(p.w + tiled_index(wn, kf, c, K, p.rw, p.nch));
            }
#pragma unroll
        for (int q = 0; q < XU; q++) {
            int kf = k0 + 8 * (xkc + XQ * q);
            kf = kf < K ? kf : 0;
            xreg[q] = *(const uint4*)(p.x + (size_t)xm * K + kf);
        }
    };
    // a unit of 8 consecutive k = k-groups 2kc (elements 0..3) and 2kc+1 (4..7): lane-stream kk gets the word (e[kk] | e[4+kk] << 16)
    auto pack4 = [](const uint4& v, uint32_t (&w)[4]) {
        w[0] = (v.x & 0xFFFFu) | (v.z << 16); w[1] = (v.x >> 16) | (v.z & 0xFFFF0000u);
        w[2] = (v.y & 0xFFFFu) | (v.w << 16); w[3] = (v.y >> 16) | (v.w & 0xFFFF0000u);
    };
    char* const wblk = As + (size_t)((wrow >> 4) * 4) * 16 * GM_APAD + (size_t)(wrow & 15) * GM_APAD;   // (+ c*WN*64*GM_APAD per chain)
    char* const xblk = Bs + (size_t)(xrow & 15) * GM_BPAD + ((BF32 ? 2 : 1) * xkc * TB + (xrow >> 4)) * 4;
    auto commit = [&](int k0) {                              // raw bf16, no conversion (beyond K: zeros)
        const bool full = k0 + GM_KS <= K;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int q = 0; q < WU; q++) {
                const int kc = wkc + q * WQ;
                uint4 v = wreg[c][q];
                if (!full && !((rw4 ? k0 + kc : k0 + 8 * kc) + 7 * kst < K)) v = make_uint4(0, 0, 0, 0);
                char* blk = wblk + (size_t)c * WN * 64 * GM_APAD;
                if (rw4) {                                   // element e -> k = 16e + kc: k-group 4e + kc/4, stream kc&3
                    char* d = blk + (size_t)(kc & 3) * 16 * GM_APAD + (kc >> 2) * 2;
                    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 8; e++) *(uint16_t*)(d + e * 8) = (uint16_t)(wd[e >> 1] >> ((e & 1) * 16));
                } else {
                    uint32_t w4[4]; pack4(v, w4);
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) *(uint32_t*)(blk + (size_t)kk * 16 * GM_APAD + kc * 4) = w4[kk];
                }
            }
#pragma unroll
        for (int q = 0; q < XU; q++) {
            uint4 v = xreg[q];
            if (!full && !(k0 + 8 * (xkc + XQ * q) < K)) v = make_uint4(0, 0, 0, 0);
            if (BF32) {                                      // element kk -> k-group 2kc, element 4+kk -> k-group 2kc+1, widened here
                const uint32_t e0[4] = {v.x << 16, v.x & 0xFFFF0000u, v.y << 16, v.y & 0xFFFF0000u};
                const uint32_t e1[4] = {v.z << 16, v.z & 0xFFFF0000u, v.w << 16, v.w & 0xFFFF0000u};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    char* d = xblk + (size_t)kk * 16 * GM_BPAD + q * 2 * XQ * TB * 4;
                    *(uint32_t*)d = e0[kk]; *(uint32_t*)(d + TB * 4) = e1[kk];
                }
            } else {
                uint32_t w4[4]; pack4(v, w4);
#pragma unroll
                for (int kk = 0; kk < 4; kk++) *(uint32_t*)(xblk + (size_t)kk * 16 * GM_BPAD + q * XQ * TB * 4) = w4[kk];
            }
        }
    };
    const char* ap = As + (size_t)(((WN == 4 ? wave : 0) * 4 + fk) * 16 + fi) * GM_APAD;       // (+ c*WN*64*GM_APAD for chain c)
    const char* bp = Bs + (size_t)(fk * 16 + fi) * GM_BPAD + mt0 * 4;
    issue(0);
    for (int k0 = 0; k0 < K; k0 += GM_KS) {
        if (!(GM_DBG & 1) || k0 == 0) commit(k0);
        if (!(GM_DBG & 2)) __syncthreads();
        if (!(GM_DBG & 1) && k0 + GM_KS < K) issue(k0 + GM_KS);   // in flight during this slab's MFMAs
        // ---- 32 k-groups of 4 as 16 pairs: acc[n-tile][m-tile t] = mfma(A, B, acc), k ascending (beyond K both operands are 0).
        // Fully unrolled and software-pipelined by hand, three stages (the sched_barriers pin them):
        //   read    pair j+2's B words (and every fourth pair the next eight groups' A words): ds_read_b128, raw bf16 pairs;
        //   widen   pair j+1's operands into their own registers (lo half: shift, hi half: mask -- 2*(MT+NCH) VALU ops);
        //   MFMA    pair j: 2 k-groups x MT x NCH matrix instructions back to back, nothing in between.
        // Measured (tools/gemmbench.hip, tools/mfma_lds_bench.hip): a widening op right in front of its MFMA -- hipcc's choice, into
        // ONE reused register -- serialises on that register (write-after-read against the MFMA in flight + the VALU->MFMA hazard
        // nops): 52 instead of 32 cycles per MFMA; ds_reads among MFMAs are free; an LDS round trip in front of each MFMA pair
        // (hipcc's other choice) costs 40 %.
        constexpr int NP = GM_KS / 8;
        uint32_t fb[2][BF32 ? 2 : 1][MT], fa[2][NCH][4];    // raw ring: pair j+1 (being widened / waiting), pair j+2 being read
        float wa[2][2][NCH], wb[2][2][BF32 ? 1 : MT];        // [pair parity][k-group parity]
        auto read_b = [&](int j) {
            const int sl = j & 1;
            if (GM_DBG & 4) { for (int h = 0; h < (BF32 ? 2 : 1); h++) for (int t = 0; t < MT; t++) fb[sl][h][t] = (uint32_t)(k0 + j + t) * 0x10001u; return; }
            if (BF32) {                                      // two k-groups x MT f32 values
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int t4 = 0; t4 < MT; t4 += 4) {
                        const uint4 v = *(const uint4*)(bp + ((2 * j + h) * TB + t4) * 4);
                        fb[sl][h % (BF32 ? 2 : 1)][t4] = v.x; fb[sl][h % (BF32 ? 2 : 1)][(t4 + 1) % MT] = v.y;
                        fb[sl][h % (BF32 ? 2 : 1)][(t4 + 2) % MT] = v.z; fb[sl][h % (BF32 ? 2 : 1)][(t4 + 3) % MT] = v.w;
                    }
            } else if (MT == 8) {                            // raw bf16 pairs: one word = (k-group 2j | k-group 2j+1 << 16) of one m-tile
                const uint4 lo = *(const uint4*)(bp + j * 32), hi = *(const uint4*)(bp + j * 32 + 16);
                fb[sl][0][0] = lo.x; fb[sl][0][1] = lo.y; fb[sl][0][2 % MT] = lo.z; fb[sl][0][3 % MT] = lo.w;
                fb[sl][0][4 % MT] = hi.x; fb[sl][0][5 % MT] = hi.y; fb[sl][0][6 % MT] = hi.z; fb[sl][0][7 % MT] = hi.w;
            } else if (MT == 4) {
                const uint4 lo = *(const uint4*)(bp + j * 16);
                fb[sl][0][0] = lo.x; fb[sl][0][1] = lo.y; fb[sl][0][2 % MT] = lo.z; fb[sl][0][3 % MT] = lo.w;
            } else {
                const uint2 v = *(const uint2*)(bp + j * (TB * 4));
                fb[sl][0][0] = v.x; fb[sl][0][1 % MT] = v.y;
            }
        };
        auto read_a = [&](int o) {                           // octet o: k-groups 8o .. 8o+7
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (GM_DBG & 4) { for (int e = 0; e < 4; e++) fa[o & 1][c][e] = (uint32_t)(k0 + o + e) * 0x10001u; continue; }
                const uint4 v = *(const uint4*)(ap + (size_t)c * WN * 64 * GM_APAD + o * 16);
                fa[o & 1][c][0] = v.x; fa[o & 1][c][1] = v.y; fa[o & 1][c][2] = v.z; fa[o & 1][c][3] = v.w;
            }
        };
        auto widen = [&](int j) {
            const int sl = j & 1, q = j & 1;
#pragma unroll
            for (int c = 0; c < NCH; c++) { const uint32_t w = fa[(j >> 2) & 1][c][j & 3]; wa[q][0][c] = bf_lo(w); wa[q][1][c] = bf_hi(w); }
            if (!BF32) {
#pragma unroll
                for (int t = 0; t < MT; t++) { wb[q][0][t % (BF32 ? 1 : MT)] = bf_lo(fb[sl][0][t]); wb[q][1][t % (BF32 ? 1 : MT)] = bf_hi(fb[sl][0][t]); }
            }
        };
        read_a(0); read_b(0);
        widen(0);
        if (!BF32) read_b(1);
#pragma unroll
        for (int j = 0; j < NP; j++) {
            if (j + 1 < NP) widen(j + 1);                    // (frees raw slot j+1 & 1 ... which pair j+2 then refills)
            if (BF32) { if (j + 1 < NP) read_b(j + 1); }     // f32 tile: pair j+1's values are read while pair j's MFMAs run, used as they are
            else if (j + 2 < NP) read_b(j + 2);
            if ((j & 3) == 1 && j + 3 < NP) read_a(j / 4 + 1);   // octet o+1 is first widened at pair 4o+3, last use of octet o-1 was pair 4o-1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int par = 0; par < 2; par++)
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    const float bval = BF32 ? __uint_as_float(fb[j & 1][par % (BF32 ? 2 : 1)][t]) : wb[j & 1][par][t % (BF32 ? 1 : MT)];
#pragma unroll
                    for (int c = 0; c < NCH; c++) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j & 1][par][c], bval, acc[c][t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(GM_DBG & 2)) __syncthreads();
    }
    // D layout: lane holds rows (lane>>4)*4 + r of the n-tile, column lane&15 of the m-tile
#pragma unroll
    for (int t = 0; t < MT; t++)
        gemm_epilogue4<EPI>(p, acc[0][t], acc[NCH - 1][t], m0 + (mt0 + t) * 16 + (lane & 15), n0 + nt_off + (lane >> 4) * 4);
}

// The softmax numerator of every possible raw score, in f64: a q.k chain is truncated to bf16 before anything else happens to it
// (llamatransformer.go:459), so  s -> trunc(s / sqrt(hd)) -> (+ 0 where unmasked) -> exp  is a function of 16 bits.  The table is
// filled by the DEVICE code path attn_exact_kernel evaluates inline (same __fdiv_rn, same exp), so a lookup is bit-identical to
// computing it (adding the mask's +0 can only turn -0 into +0, and exp(-0) == exp(+0)); the prefill attention replaces ~130
// instructions per element by one 8-byte load from 512 KB (the reference does the same for SiLU, activations.go:15-25).
__global__ void exp_table_kernel(double* tab, float divisor) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1 << 16)) return;
    const uint16_t s16 = bf_trunc(__fdiv_rn(bf_wide((uint16_t)i), divisor));        // DivToScalar :464
    tab[i] = exp((double)bf_wide(s16));                                              // Softmax impl:498
}
extern "C" hipError_t lnbk_exp_table(double* tab, float divisor, hipStream_t st) {
    hipLaunchKernelGGL(exp_table_kernel, dim3(256), dim3(256), 0, st, tab, divisor);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attn_mfma_kernel: the prefill form of the attention (S >= 16 query rows), the same arithmetic as attn_exact_kernel below with
// both matrix products on the exact f32 matrix cores (v_mfma_f32_16x16x4_f32 == the reference's k-ordered chain, see gemm_mfma_kernel).
// One WAVE owns 16 query rows of one head and walks the cached positions in tiles of 16, twice:
//   pass 1  S^T tile [16 j][16 i] = sum_d K[j][d] Q[i][d] (d ascending: 32 dependent MFMAs), then per element trunc, /sqrt(hd), mask,
//           exp in f64; the row sums Z_i must be added in j order in f64: the tile goes through a 2.3 KB LDS patch and lane i < 16
//           walks row i's sixteen values (the reference's serial sum, impl:492-499);
//   pass 2  the scores again (cheaper than keeping S x T doubles), p = trunc(f32(e / Z_i)), transposed through the LDS into the A
//           operand layout, and out[i][d] += p[i][j] v[j][d] with j ascending inside and across the MFMAs (lane n owns the HD/16
//           consecutive output dims n*HD/16.., so one 16 B load of a V row feeds its eight MFMAs).
// Masked positions (j % S > i, the modulo-broadcast [S,S] mask) contribute e = 0 -> p = +0; acc + (+-0) == acc and acc is never -0,
// so with the standard causal layout (pos0 == 0) the tiles above the diagonal are skipped like attn_exact_kernel does.
// K cache layout [kv head][d/8][position][8]: a lane (j, kk) loads its position's 16 B units and picks elements kk and 4+kk
// (k-groups 2c, 2c+1) with one v_perm_b32 each; the four kk-lanes of a position load the same unit (L1 traffic, not HBM).
// grid (H, ceil(S/64)), block 256 = 4 independent waves (no workgroup barrier anywhere).
// ------------------------------------------------------------------------------------------------
// workgroup -> (head, sub-block) for the attention grids (gridDim = (heads, sub-blocks)).  The hardware hands workgroup w to XCD w % 8,
// each XCD has its own L2, and the query heads of one GQA group read the SAME K / V rows: with the plain (blockIdx.x, blockIdx.y)
// mapping the four heads of a group sit on four XCDs and the rows are fetched from HBM four times (rocprofv3 FETCH_SIZE of the
// long-context kernels at T = 4101: 33.3 MB per launch against 8.4 MB of V).  Here consecutive VIRTUAL ids stay on one XCD and virtual
// ids are head-major, so a group's heads share an L2 (8B shape: XCD x runs heads 4x .. 4x+3 = KV head x).
// xcd_head_block: inside an XCD head-major (a head's sub-blocks follow each other: the decode kernels, equal-cost sub-blocks; measured
// 191.0 against 189.5 tokens/s at configs[2] for the other order).  xcd_head_block_bmajor: sub-block-major (all of the XCD's heads for
// sub-block 0, then sub-block 1, ...), so a caller whose sub-blocks differ in cost can hand out the longest ones first (prefill).
DEVINL void xcd_head_block(int& h, int& b) {
    const unsigned H = gridDim.x, nb = gridDim.y, lin = blockIdx.y * H + blockIdx.x;
    if ((H & 7u) == 0) {
        const unsigned v = (lin & 7u) * ((H >> 3) * nb) + (lin >> 3);     // consecutive virtual ids stay on one XCD
        h = (int)(v / nb); b = (int)(v % nb);
    } else { h = (int)blockIdx.x; b = (int)blockIdx.y; }
}
DEVINL void xcd_head_block_bmajor(int& h, int& b) {
    const unsigned H = gridDim.x, lin = blockIdx.y * H + blockIdx.x;
    if ((H & 7u) == 0) {
        const unsigned hg = H >> 3, w = lin >> 3;            // heads per XCD; index inside the XCD's share
        h = (int)((lin & 7u) * hg + w % hg); b = (int)(w / hg);
    } else { h = (int)blockIdx.x; b = (int)blockIdx.y; }
}
constexpr int ATM_ET = 144, ATM_PT = 80, ATM_WLDS = 16 * ATM_ET + 16 * ATM_PT + 128;     // per-wave LDS patch: e tile | p tile | Z row
template <int HD> __global__ __launch_bounds__(256, 2) void attn_mfma_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char sm[4 * ATM_WLDS];
    constexpr int NK = HD / 8, DPL = HD / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, by_; xcd_head_block_bmajor(h, by_);
    by_ = (int)gridDim.y - 1 - by_;                          // the last query rows see the longest context: handed out first
    const int i0 = (by_ * 4 + wave) * 16;
    const int S = p.S, H = p.H, KVH = p.KVH;
    if (i0 >= S) return;                                     // (waves are independent)
    const int pos0 = p.st->pos, T = pos0 + S;
    const int kvh = h / (H / KVH);
    const int fi = lane & 15, fk = lane >> 4;
    char* et = sm + wave * ATM_WLDS;
    char* pt = et + 16 * ATM_ET;
    double* zrow = (double*)(pt + 16 * ATM_PT);
    const uint4* kbase = (const uint4*)p.cache_k + (size_t)kvh * NK * p.seq_len;
    const uint16_t* vbase = p.cache_v + (size_t)kvh * HD + fi * DPL;
    const size_t vrow = (size_t)KVH * HD;
    const uint32_t sel = 0x0c0cu | ((uint32_t)(2 * fk) << 16) | ((uint32_t)(2 * fk + 1) << 24);   // element kk of a 4-element word pair
    const int Tmax = (S > 1 && pos0 == 0) ? (i0 + 16 < T ? i0 + 16 : T) : T;
    const int NJ = (Tmax + 15) >> 4;
    const int irow = i0 + fi;                                // this lane's query row in the S^T tile (column i)

    float qf[2 * NK];                                        // Q[irow][4g + kk], g = 0 .. HD/4-1
    {
        const uint4* q = (const uint4*)(p.q + ((size_t)(irow < S ? irow : S - 1) * H + h) * HD);
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const uint4 u = q[c];
            qf[2 * c] = __uint_as_float(__builtin_amdgcn_perm(u.y, u.x, sel));
            qf[2 * c + 1] = __uint_as_float(__builtin_amdgcn_perm(u.w, u.z, sel));
        }
    }
    auto load_k = [&](uint4 (&k)[NK], int j0) {
        int j = j0 + fi; j = j < T ? j : T - 1;
#pragma unroll
        for (int c = 0; c < NK; c++) k[c] = kbase[(size_t)c * p.seq_len + j];
    };
    // raw scores of the tile: lane holds positions j0 + 4*kk + r (r = 0..3) against query row irow
    auto qk = [&](const uint4 (&k)[NK]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NK; c++) {                       // MatMul q.k, d ascending (operations_matmul.go:37-55)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(__builtin_amdgcn_perm(k[c].y, k[c].x, sel)), qf[2 * c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(__builtin_amdgcn_perm(k[c].w, k[c].z, sel)), qf[2 * c + 1], acc, 0, 0, 0);
        }
        return acc;
    };
    // e[r] = exp(score) (0 where masked or past T)
    auto expo = [&](const f32x4& acc, int j0, double (&e)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = j0 + 4 * fk + r;
            const bool dead = j >= T || ((S > 1) && ((pos0 == 0 ? j : j % S) > irow));   // triu(-inf,1) broadcast by modulo (tensoriterators.go:47-55);
                                                                                         // pos0 == 0: T == S, no wrap (uniform branch, saves the division)
            const double ev = p.exp_tab[bf_trunc(acc[r])];                       // / sqrt(hd) :464, (+ mask 0 :469-473), exp impl:498: tabulated
            e[r] = dead ? 0.0 : ev;                                              // exp(-inf) == 0
        }
    };
#define ATM_LDS_TURN() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

    // ---- pass 1: Z_i = sum_j exp(s_ij), f64, j ascending (lane i < 16 carries row i)
    double z = 0.0;
    uint4 kt[NK];                                            // the tile's K rows; the next tile's are loaded into the same registers as
                                                             // soon as the 32 MFMAs have read them, and land during the exps
    auto zsum = [&](int j0, bool more) {
        const f32x4 sc = qk(kt);
        if (more) load_k(kt, j0 + 16);
        double e[4];
        expo(sc, j0, e);
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2* w = (d2*)(et + fi * ATM_ET + fk * 32);
        w[0] = d2{e[0], e[1]}; w[1] = d2{e[2], e[3]};
        ATM_LDS_TURN();
        if (lane < 16) {
            const d2* row = (const d2*)(et + lane * ATM_ET);
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = row[u];
#pragma unroll
            for (int u = 0; u < 8; u++) { z += v[u].x; z += v[u].y; }
        }
        ATM_LDS_TURN();
    };
    load_k(kt, 0);
    for (int jt = 0; jt < NJ; jt++) zsum(jt * 16, jt + 1 < NJ);      // (wave-uniform trip count)
    if (lane < 16) zrow[lane] = z;
    ATM_LDS_TURN();
    const double zi = zrow[fi];

    // ---- pass 2: p = trunc(f32(e / Z_i)), out = sum_j p_j v_j (j ascending)
    f32x4 o[DPL];
#pragma unroll
    for (int t = 0; t < DPL; t++) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pv = [&](int j0, bool more) {
        const f32x4 sc = qk(kt);
        if (more) load_k(kt, j0 + 16);
        uint4 vv[4];                                         // V[j0 + 4g + kk][dims fi*DPL ..]: in flight during the exps
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int j = j0 + 4 * g + fk; j = j < T ? j : T - 1;
            const uint16_t* a = vbase + (size_t)j * vrow;
            if (DPL == 8) vv[g] = *(const uint4*)a;
            else { const uint2 t2 = *(const uint2*)a; vv[g] = make_uint4(t2.x, t2.y, 0, 0); }
        }
        double e[4];
        expo(sc, j0, e);
        float4 pf;                                           // impl:506 + ToBFloat16 :493
        pf.x = bf_wide(bf_trunc((float)(e[0] / zi))); pf.y = bf_wide(bf_trunc((float)(e[1] / zi)));
        pf.z = bf_wide(bf_trunc((float)(e[2] / zi))); pf.w = bf_wide(bf_trunc((float)(e[3] / zi)));
        *(float4*)(pt + fi * ATM_PT + fk * 16) = pf;         // p[i = fi][j = 4kk .. 4kk+3]
        ATM_LDS_TURN();
        float pa[4];
#pragma unroll
        for (int g = 0; g < 4; g++) pa[g] = *(const float*)(pt + fi * ATM_PT + (4 * g + fk) * 4);     // A operand: p[i = fi][j = 4g + kk]
        ATM_LDS_TURN();
#pragma unroll
        for (int g = 0; g < 4; g++) {                        // MatMul p.v, j ascending (llamatransformer.go:504-514)
            const uint32_t wd[4] = {vv[g].x, vv[g].y, vv[g].z, vv[g].w};
#pragma unroll
            for (int t = 0; t < DPL; t++)
                o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[g], (t & 1) ? bf_hi(wd[t >> 1]) : bf_lo(wd[t >> 1]), o[t], 0, 0, 0);
        }
    };
    load_k(kt, 0);
    for (int jt = 0; jt < NJ; jt++) pv(jt * 16, jt + 1 < NJ);
#undef ATM_LDS_TURN
    // D layout: lane holds query rows 4*kk + r, output dims fi*DPL + t
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = i0 + 4 * fk + r;
        if (row >= S) continue;
        uint32_t w[DPL / 2];
#pragma unroll
        for (int t = 0; t < DPL; t += 2) w[t / 2] = (uint32_t)bf_trunc(o[t][r]) | ((uint32_t)bf_trunc(o[t + 1][r]) << 16);
        uint16_t* dst = p.out + ((size_t)row * H + h) * HD + fi * DPL;
        if (DPL == 8) *(uint4*)dst = make_uint4(w[0], w[1], w[2 % (DPL / 2)], w[3 % (DPL / 2)]);
        else *(uint2*)dst = make_uint2(w[0], w[1]);
    }
}

// ------------------------------------------------------------------------------------------------
// attn_mfma3_kernel (round 6): attn_mfma_kernel with the scores computed ONCE and the exp-table look-ups issued a tile ahead.
//
// profiles/r06_prefill_stalls.md: of the 11.3 k cycles a wave of attn_mfma_kernel spends per (query tile, position tile) pair, 7.0 k are issue stalls on the matrix pipe
// (a wave's program is chains of dependent matrix instructions, 32 for q.k -- and pass 2 runs that chain a second time only to get the same sixteen-bit values back) and
// 2.0 k are spent parked on the four table look-ups of each pass.  Here
//   pass 1  keeps what the table is indexed with: the truncated raw score (0xFF80 = -inf -> exp == 0 for masked / past-the-end elements), four 16-bit values per lane and
//           tile = one 8-byte store into score_idx[head][query tile][position tile][lane] (512 B per tile, coalesced; written and read by the same lane);
//   pass 2  reads them back: no K rows, no q.k chain -- a third of the matrix instructions gone;
//   both    issue a tile's look-ups and let the NEXT tile's matrix work run before the values are used (pass 1: tile t's exponentials are added to Z after tile t + 1's
//           chain; pass 2: tile t + 1's indices, look-ups and V rows go in flight before tile t's 4 x HD/16 matrix instructions).
// Every vector-memory load of the loops is an asm load with a hand-counted wait (hipcc drains vmcnt(0) around loop-carried loads, NOTES 5.1); tools/isa_audit.py follows the
// registers.  Counts assume what LLVM assumes for gfx9: loads and stores retire in issue order on the one counter.  Loads are unconditional (clamped tile / position), the
// tile loops of pass 2 are unrolled by two for static buffer indices; a padded tile is loaded and not consumed.
// Same arithmetic per element in the same order as attn_mfma_kernel: same bits (tests/test_gpu_round6.py runs both).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
DEVINL void atm_ld16(u32x4& d, unsigned voff, const char* sb) { asm volatile("global_load_dwordx4 %0, %1, %2 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory"); }
DEVINL void atm_ld8(u32x2& d, unsigned voff, const char* sb) { asm volatile("global_load_dwordx2 %0, %1, %2 ; RING_LOAD" : "=&v"(d) : "v"(voff), "s"(sb) : "memory"); }
DEVINL const char* atm_uniform(const char* a) {             // a pointer that IS wave-uniform, told to the compiler (an "s" operand must not be handed a VGPR pair)
    const size_t v = (size_t)a;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(((size_t)hi << 32) | lo);
}
template <int HD> __global__ __launch_bounds__(256, 2) void attn_mfma3_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char sm[4 * ATM_WLDS];
    constexpr int NK = HD / 8, DPL = HD / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, by_; xcd_head_block_bmajor(h, by_);
    by_ = (int)gridDim.y - 1 - by_;                          // the last query rows see the longest context: handed out first
    const int itile = by_ * 4 + wave, i0 = itile * 16;
    const int S = p.S, H = p.H, KVH = p.KVH;
    if (i0 >= S) return;                                     // (waves are independent)
    const int pos0 = p.st->pos, T = pos0 + S;
    const int kvh = h / (H / KVH);
    const int fi = lane & 15, fk = lane >> 4;
    char* et = sm + wave * ATM_WLDS;
    char* pt = et + 16 * ATM_ET;
    double* zrow = (double*)(pt + 16 * ATM_PT);
    const uint32_t sel = 0x0c0cu | ((uint32_t)(2 * fk) << 16) | ((uint32_t)(2 * fk + 1) << 24);   // element kk of a 4-element word pair
    const int Tmax = (S > 1 && pos0 == 0) ? (i0 + 16 < T ? i0 + 16 : T) : T;
    const int NJ = (Tmax + 15) >> 4;
    const int irow = i0 + fi;                                // this lane's query row in the S^T tile (column i)
    const char* const kb = atm_uniform((const char*)p.cache_k + (size_t)kvh * NK * p.seq_len * 16);            // K runs of the KV head: [d / 8][position][8]
    const size_t krun = (size_t)p.seq_len * 16;
    const char* const vb = atm_uniform((const char*)p.cache_v + (size_t)kvh * HD * 2);
    const unsigned vrow2 = (unsigned)(KVH * HD * 2), vlane = (unsigned)(fi * DPL * 2);
    const char* const xb = atm_uniform((const char*)(p.score_idx + ((size_t)(h * ((S + 15) >> 4) + itile) * p.sidx_jt) * 64));
    const char* const tab = atm_uniform((const char*)p.exp_tab);

    float qf[2 * NK];                                        // Q[irow][4g + kk], g = 0 .. HD/4-1
    {
        const uint4* q = (const uint4*)(p.q + ((size_t)(irow < S ? irow : S - 1) * H + h) * HD);
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const uint4 u = q[c];
            qf[2 * c] = __uint_as_float(__builtin_amdgcn_perm(u.y, u.x, sel));
            qf[2 * c + 1] = __uint_as_float(__builtin_amdgcn_perm(u.w, u.z, sel));
        }
    }
    __builtin_amdgcn_sched_barrier(0);                       // (the compiler's own loads are consumed in front of the first asm load)
#define ATM_LDS_TURN() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define ATM_RETIRE(r) asm volatile("; RING_RETIRE %0" : "+v"(r))
    typedef double d2 __attribute__((ext_vector_type(2)));
    auto as_double = [](const u32x2& v) { return __hiloint2double((int)v.y, (int)v.x); };

    // ---- pass 1: raw scores -> table indices (kept) -> Z_i = sum_j exp(s_ij), f64, j ascending (lane i < 16 carries row i)
    u32x4 kr[NK];
    u32x2 er[4];
    auto k_issue = [&](int jt) {
        int j = jt * 16 + fi; j = j < T ? j : T - 1;
        const unsigned voff = (unsigned)j * 16u;
#pragma unroll
        for (int c = 0; c < NK; c++) atm_ld16(kr[c], voff, kb + (size_t)c * krun);
    };
    double z = 0.0;
    auto zadd = [&]() {                                      // the look-ups in er have landed: the tile goes through the LDS patch, lane i walks row i's sixteen values
        d2* w = (d2*)(et + fi * ATM_ET + fk * 32);
        w[0] = d2{as_double(er[0]), as_double(er[1])}; w[1] = d2{as_double(er[2]), as_double(er[3])};
        ATM_LDS_TURN();
        if (lane < 16) {
            const d2* row = (const d2*)(et + lane * ATM_ET);
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = row[u];
#pragma unroll
            for (int u = 0; u < 8; u++) { z += v[u].x; z += v[u].y; }
        }
        ATM_LDS_TURN();
    };
    k_issue(0);
#pragma unroll
    for (int r = 0; r < 4; r++) atm_ld8(er[r], 0xFF80u * 8u, tab);     // "tile -1": four zeros (exp(-inf)), so that the loop has one shape; z + 0.0 == z
    u32x2 pk = {0u, 0u};                                     // the previous tile's packed indices: stored one tile late, right behind the next K rows -- by the time a
                                                             // counted wait has the store in front of it, the store is a whole chain old (the counts below count LOADS only)
    for (int jt = 0; jt < NJ; jt++) {                        // (wave-uniform trip count)   in flight here: K(jt) [NK], (store), look-ups(jt - 1) [4]
        const int j0 = jt * 16;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // K(jt) has landed; the four look-ups behind it may not have
#pragma unroll
        for (int c = 0; c < NK; c++) ATM_RETIRE(kr[c]);
        float ko[2 * NK];
#pragma unroll
        for (int c = 0; c < NK; c++) {
            ko[2 * c] = __uint_as_float(__builtin_amdgcn_perm(kr[c].y, kr[c].x, sel));
            ko[2 * c + 1] = __uint_as_float(__builtin_amdgcn_perm(kr[c].w, kr[c].z, sel));
        }
#pragma unroll
        for (int g = 0; g < 2 * NK; g++) asm volatile("" : "+v"(ko[g]));      // widened in front of the refill
        __builtin_amdgcn_sched_barrier(0);
        k_issue(jt + 1 < NJ ? jt + 1 : jt);                  // in flight: look-ups(jt - 1) [4], K(jt + 1) [NK], store
        if (jt > 0) asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"((unsigned)(lane * 8 + (jt - 1) * 512)), "v"(pk), "s"(xb) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2 * NK; g++) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(ko[g], qf[g], sc, 0, 0, 0);      // MatMul q.k, d ascending (operations_matmul.go:37-55)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NK) : "memory");             // the PREVIOUS tile's exponentials: their latency ran under this tile's chain
#pragma unroll
        for (int r = 0; r < 4; r++) ATM_RETIRE(er[r]);
        zadd();
        unsigned ix[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = j0 + 4 * fk + r;
            const bool dead = j >= T || ((S > 1) && ((pos0 == 0 ? j : j % S) > irow));   // triu(-inf,1) broadcast by modulo (tensoriterators.go:47-55)
            ix[r] = dead ? 0xFF80u : (unsigned)bf_trunc(sc[r]);                          // -inf: exp == 0 (the mask's -inf added to the score, :469-473)
        }
        asm volatile("" : "+v"(ix[0]), "+v"(ix[1]), "+v"(ix[2]), "+v"(ix[3]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; r++) atm_ld8(er[r], ix[r] * 8u, tab);           // / sqrt(hd) :464, exp impl:498: tabulated over the raw score (exp_table_kernel)
        pk = u32x2{ix[0] | (ix[1] << 16), ix[2] | (ix[3] << 16)};
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"((unsigned)(lane * 8 + (NJ - 1) * 512)), "v"(pk), "s"(xb) : "memory");
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
#pragma unroll
    for (int c = 0; c < NK; c++) ATM_RETIRE(kr[c]);
#pragma unroll
    for (int r = 0; r < 4; r++) ATM_RETIRE(er[r]);
    zadd();                                                  // the last tile's
    if (lane < 16) zrow[lane] = z;
    ATM_LDS_TURN();
    const double zi = zrow[fi];
    __builtin_amdgcn_sched_barrier(0);

    // ---- pass 2: p = trunc(f32(e / Z_i)), out = sum_j p_j v_j (j ascending); e from the kept indices
    f32x4 o[DPL];
#pragma unroll
    for (int t = 0; t < DPL; t++) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 ixn;                                               // the NEXT tile's packed indices
    u32x2 e2[2][4];
    using VT = std::conditional_t<DPL == 8, u32x4, u32x2>;   // HD / 16 dims of a V row per lane: 16 or 8 bytes
    VT vv[2][4];
    auto idx_issue = [&](int t) { t = t < NJ ? t : NJ - 1; atm_ld8(ixn, (unsigned)(lane * 8 + t * 512), xb); };
    auto look_issue = [&](u32x2 (&e)[4], const unsigned (&o8)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) atm_ld8(e[r], o8[r], tab);
    };
    auto v_issue = [&](VT (&v)[4], int t) {                  // V[16 t + 4g + kk][dims fi*DPL ..]
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int j = t * 16 + 4 * g + fk; j = j < T ? j : T - 1;
            const unsigned voff = (unsigned)j * vrow2 + vlane;
            if constexpr (DPL == 8) atm_ld16(v[g], voff, vb); else atm_ld8(v[g], voff, vb);
        }
    };
    auto offs_of = [&](unsigned (&o8)[4]) {                  // byte offsets into the table from the packed indices in ixn
        o8[0] = (ixn.x & 0xFFFFu) * 8u; o8[1] = (ixn.x >> 16) * 8u; o8[2] = (ixn.y & 0xFFFFu) * 8u; o8[3] = (ixn.y >> 16) * 8u;
        asm volatile("" : "+v"(o8[0]), "+v"(o8[1]), "+v"(o8[2]), "+v"(o8[3]));
    };
    {
        idx_issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ATM_RETIRE(ixn);
        unsigned o8[4]; offs_of(o8);
        __builtin_amdgcn_sched_barrier(0);
        idx_issue(1); look_issue(e2[0], o8); v_issue(vv[0], 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    auto tile = [&](auto curc, int t) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        // in flight: look-ups(t) [4], V(t) [4], indices(t + 1) [1] -- all issued a whole tile ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; r++) { ATM_RETIRE(e2[cur][r]); ATM_RETIRE(vv[cur][r]); }
        ATM_RETIRE(ixn);
        unsigned o8[4]; offs_of(o8);
        __builtin_amdgcn_sched_barrier(0);
        idx_issue(t + 2); look_issue(e2[nxt], o8); v_issue(vv[nxt], t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t < NJ) {                                        // (wave-uniform; a padded tile is loaded and not consumed)
            float4 pf;                                       // impl:506 + ToBFloat16 :493
            pf.x = bf_wide(bf_trunc((float)(as_double(e2[cur][0]) / zi))); pf.y = bf_wide(bf_trunc((float)(as_double(e2[cur][1]) / zi)));
            pf.z = bf_wide(bf_trunc((float)(as_double(e2[cur][2]) / zi))); pf.w = bf_wide(bf_trunc((float)(as_double(e2[cur][3]) / zi)));
            *(float4*)(pt + fi * ATM_PT + fk * 16) = pf;     // p[i = fi][j = 4kk .. 4kk+3]
            ATM_LDS_TURN();
            float pa[4];
#pragma unroll
            for (int g = 0; g < 4; g++) pa[g] = *(const float*)(pt + fi * ATM_PT + (4 * g + fk) * 4);     // A operand: p[i = fi][j = 4g + kk]
            ATM_LDS_TURN();
#pragma unroll
            for (int g = 0; g < 4; g++) {                    // MatMul p.v, j ascending (llamatransformer.go:504-514)
                uint32_t wd[DPL / 2];
#pragma unroll
                for (int q = 0; q < DPL / 2; q++) wd[q] = vv[cur][g][q];
#pragma unroll
                for (int t2 = 0; t2 < DPL; t2++)
                    o[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[g], (t2 & 1) ? bf_hi(wd[t2 >> 1]) : bf_lo(wd[t2 >> 1]), o[t2], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < NJ; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");
#undef ATM_RETIRE
#undef ATM_LDS_TURN
    // D layout: lane holds query rows 4*kk + r, output dims fi*DPL + t
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = i0 + 4 * fk + r;
        if (row >= S) continue;
        uint32_t w[DPL / 2];
#pragma unroll
        for (int t = 0; t < DPL; t += 2) w[t / 2] = (uint32_t)bf_trunc(o[t][r]) | ((uint32_t)bf_trunc(o[t + 1][r]) << 16);
        uint16_t* dst = p.out + ((size_t)row * H + h) * HD + fi * DPL;
        if (DPL == 8) *(uint4*)dst = make_uint4(w[0], w[1], w[2 % (DPL / 2)], w[3 % (DPL / 2)]);
        else *(uint2*)dst = make_uint2(w[0], w[1]);
    }
}

// ------------------------------------------------------------------------------------------------
// attn_mfma2_kernel (round 6): attn_mfma_kernel with TWO query tiles (32 rows) per wave against each tile of cached positions.
//
// attn_mfma_kernel was 0.517 MFMA-busy for three rounds (profiles/r05_prefill_mfma_counters.md): per tile a wave runs 32 DEPENDENT matrix
// instructions (40 cycles of latency at a 32-cycle issue rate), waits for four table look-ups, turns the LDS patch twice, walks sixteen f64 adds on
// a quarter of its lanes -- and then does it all again in pass 2.  With two query tiles A and B per wave the SAME K operand (one v_perm per k-group instead
// of one per matrix instruction) feeds two independent accumulator chains, S_A^T and S_B^T, alternately: no dependent-issue stall; the V rows of a tile
// are loaded and unpacked once for both; the serial Z adds of the 32 rows run on lanes 0..31 in the same sixteen instructions; the LDS turns, the look-up
// latency and the loop overhead are paid once per 2 x 96 matrix instructions.  Same arithmetic per element, same order (j ascending inside and across the
// matrix instructions), same bits.  Tile A's rows see the diagonal one tile earlier than B's: its masked elements are e = 0, p = +0 (wasted work, not wrong).
// grid (H, ceil(S / 128)), block 256 = 4 independent waves x 32 query rows.  NOT the default (ATM2_MIN_S below): measured slower than the 16-row form.
// (Found on the way: widening an operand with an `asm volatile` VALU instruction right in front of the matrix instruction that reads it gave WRONG scores -- the
// compiler's hazard recognizer does not cover an asm-defined register feeding v_mfma; the packed words are made opaque with an EMPTY asm instead.)
// ------------------------------------------------------------------------------------------------
constexpr int ATM2_WLDS = 32 * ATM_ET + 32 * ATM_PT + 256;   // per-wave LDS patch: e tiles A|B | p tiles A|B | Z rows
constexpr int ATM2_MIN_S = 0;                                // 0: off.  Measured (profiles/r06_prefill_attn.log, 8B shape, whole Forward): 4096 rows 536.1 -> 537.8 ms, 2048: 249.9 -> 256.8, 512: 62.2 -> 64.8 -- SLOWER:
                                                             // the dependent-issue stall it removes is not what bounds the kernel (table look-up latency, eight f64 divisions and the LDS turns per tile are), and 32-row
                                                             // waves halve the number of jobs on a triangular workload.  Kept bit-exact and selectable (LNB_ATTN_MFMA2=<min rows>) for the next attempt.
template <int HD> __global__ __launch_bounds__(256, 2) void attn_mfma2_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char sm[4 * ATM2_WLDS];
    constexpr int NK = HD / 8, DPL = HD / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, by_; xcd_head_block_bmajor(h, by_);
    by_ = (int)gridDim.y - 1 - by_;                          // the last query rows see the longest context: handed out first
    const int i0 = (by_ * 4 + wave) * 32;
    const int S = p.S, H = p.H, KVH = p.KVH;
    if (i0 >= S) return;                                     // (waves are independent)
    const int pos0 = p.st->pos, T = pos0 + S;
    const int kvh = h / (H / KVH);
    const int fi = lane & 15, fk = lane >> 4;
    char* et = sm + wave * ATM2_WLDS;
    char* pt = et + 32 * ATM_ET;
    double* zrow = (double*)(pt + 32 * ATM_PT);
    const uint4* kbase = (const uint4*)p.cache_k + (size_t)kvh * NK * p.seq_len;
    const uint16_t* vbase = p.cache_v + (size_t)kvh * HD + fi * DPL;
    const size_t vrow = (size_t)KVH * HD;
    const uint32_t sel = 0x0c0cu | ((uint32_t)(2 * fk) << 16) | ((uint32_t)(2 * fk + 1) << 24);
    const int Tmax = (S > 1 && pos0 == 0) ? (i0 + 32 < T ? i0 + 32 : T) : T;       // tile B's diagonal
    const int NJ = (Tmax + 15) >> 4;
    const int irow[2] = {i0 + fi, i0 + 16 + fi};

    // Q[irow][4g + kk] of both query tiles, PACKED: the values are bf16, so the two k-groups of an 8-chunk share a register (low half: k-group 2c, high
    // half: 2c + 1) and are widened by one shift / mask in front of their matrix instruction -- 32 instead of 64 registers, which is what lets the kernel
    // keep two workgroups per CU without spilling
    uint32_t qp[2][NK];
    const uint32_t selp = ((uint32_t)(2 * (fk & 1)) | ((uint32_t)(2 * (fk & 1) + 1) << 8)) | (((uint32_t)(4 + 2 * (fk & 1)) | ((uint32_t)(5 + 2 * (fk & 1)) << 8)) << 16);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint4* q = (const uint4*)(p.q + ((size_t)(irow[t] < S ? irow[t] : S - 1) * H + h) * HD);
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const uint4 u = q[c];
            const uint32_t lo = fk < 2 ? u.x : u.y, hi = fk < 2 ? u.z : u.w;      // element kk of the first / second four of the chunk
            qp[t][c] = __builtin_amdgcn_perm(hi, lo, selp);                       // bytes: [elem kk of (x|y)] | [elem kk of (z|w)] << 16
        }
    }
    auto load_k = [&](uint4 (&k)[NK], int j0) {
        int j = j0 + fi; j = j < T ? j : T - 1;
#pragma unroll
        for (int c = 0; c < NK; c++) k[c] = kbase[(size_t)c * p.seq_len + j];
    };
    // raw scores of the tile for both query tiles: two independent chains fed by ONE K operand per k-group, d ascending (operations_matmul.go:37-55)
    auto qk2 = [&](const uint4 (&k)[NK], f32x4& sa, f32x4& sb) {
        sa = f32x4{0.f, 0.f, 0.f, 0.f}; sb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const float k0 = __uint_as_float(__builtin_amdgcn_perm(k[c].y, k[c].x, sel)), k1 = __uint_as_float(__builtin_amdgcn_perm(k[c].w, k[c].z, sel));
            // (the empty volatile asm makes the packed word opaque in every iteration: left to itself hipcc hoists the loop-invariant widening out of the
            // tile loop and is back at 64 registers + spills)
            uint32_t qa = qp[0][c], qb = qp[1][c];
            asm volatile("" : "+v"(qa), "+v"(qb));
            const float a0 = bf_lo(qa), a1 = bf_hi(qa), b0 = bf_lo(qb), b1 = bf_hi(qb);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, a0, sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, b0, sb, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, a1, sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, b1, sb, 0, 0, 0);
        }
    };
    auto expo = [&](const f32x4& acc, int j0, int ir, double (&e)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = j0 + 4 * fk + r;
            const bool dead = j >= T || ((S > 1) && ((pos0 == 0 ? j : j % S) > ir));   // triu(-inf,1) broadcast by modulo (tensoriterators.go:47-55)
            const double ev = p.exp_tab[bf_trunc(acc[r])];                       // / sqrt(hd) :464, (+ mask 0 :469-473), exp impl:498: tabulated
            e[r] = dead ? 0.0 : ev;
        }
    };
#define ATM_LDS_TURN() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    typedef double d2 __attribute__((ext_vector_type(2)));

    // ---- pass 1: Z_i = sum_j exp(s_ij), f64, j ascending (lane i < 32 carries row i of the 32)
    double z = 0.0;
    uint4 kt[NK];
    load_k(kt, 0);
    for (int jt = 0; jt < NJ; jt++) {                        // (wave-uniform trip count)
        const int j0 = jt * 16;
        f32x4 sa, sb;
        qk2(kt, sa, sb);
        if (jt + 1 < NJ) load_k(kt, j0 + 16);
        double ea[4], eb[4];
        expo(sa, j0, irow[0], ea); expo(sb, j0, irow[1], eb);
        d2* wa = (d2*)(et + fi * ATM_ET + fk * 32);
        d2* wb = (d2*)(et + (16 + fi) * ATM_ET + fk * 32);
        wa[0] = d2{ea[0], ea[1]}; wa[1] = d2{ea[2], ea[3]};
        wb[0] = d2{eb[0], eb[1]}; wb[1] = d2{eb[2], eb[3]};
        ATM_LDS_TURN();
        if (lane < 32) {
            const d2* row = (const d2*)(et + lane * ATM_ET);
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = row[u];
#pragma unroll
            for (int u = 0; u < 8; u++) { z += v[u].x; z += v[u].y; }
        }
        ATM_LDS_TURN();
    }
    if (lane < 32) zrow[lane] = z;
    ATM_LDS_TURN();
    const double zia = zrow[fi], zib = zrow[16 + fi];

    // ---- pass 2: p = trunc(f32(e / Z_i)), out = sum_j p_j v_j (j ascending)
    f32x4 oa[DPL], ob[DPL];
#pragma unroll
    for (int t = 0; t < DPL; t++) { oa[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ob[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    load_k(kt, 0);
    for (int jt = 0; jt < NJ; jt++) {
        const int j0 = jt * 16;
        f32x4 sa, sb;
        qk2(kt, sa, sb);
        if (jt + 1 < NJ) load_k(kt, j0 + 16);
        uint4 vv[4];                                         // V[j0 + 4g + kk][dims fi*DPL ..]: once for both query tiles
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int j = j0 + 4 * g + fk; j = j < T ? j : T - 1;
            const uint16_t* a = vbase + (size_t)j * vrow;
            if (DPL == 8) vv[g] = *(const uint4*)a;
            else { const uint2 t2 = *(const uint2*)a; vv[g] = make_uint4(t2.x, t2.y, 0, 0); }
        }
        double ea[4], eb[4];
        expo(sa, j0, irow[0], ea); expo(sb, j0, irow[1], eb);
        float4 pfa, pfb;                                     // impl:506 + ToBFloat16 :493
        pfa.x = bf_wide(bf_trunc((float)(ea[0] / zia))); pfa.y = bf_wide(bf_trunc((float)(ea[1] / zia)));
        pfa.z = bf_wide(bf_trunc((float)(ea[2] / zia))); pfa.w = bf_wide(bf_trunc((float)(ea[3] / zia)));
        pfb.x = bf_wide(bf_trunc((float)(eb[0] / zib))); pfb.y = bf_wide(bf_trunc((float)(eb[1] / zib)));
        pfb.z = bf_wide(bf_trunc((float)(eb[2] / zib))); pfb.w = bf_wide(bf_trunc((float)(eb[3] / zib)));
        *(float4*)(pt + fi * ATM_PT + fk * 16) = pfa;        // p[i = fi][j = 4kk .. 4kk+3] of tile A, B
        *(float4*)(pt + (16 + fi) * ATM_PT + fk * 16) = pfb;
        ATM_LDS_TURN();
        float pa[4], pb[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {                        // A operands: p[i = fi][j = 4g + kk]
            pa[g] = *(const float*)(pt + fi * ATM_PT + (4 * g + fk) * 4);
            pb[g] = *(const float*)(pt + (16 + fi) * ATM_PT + (4 * g + fk) * 4);
        }
        ATM_LDS_TURN();
#pragma unroll
        for (int g = 0; g < 4; g++) {                        // MatMul p.v, j ascending (llamatransformer.go:504-514)
            const uint32_t wd[4] = {vv[g].x, vv[g].y, vv[g].z, vv[g].w};
#pragma unroll
            for (int t = 0; t < DPL; t++) {
                const float vf = (t & 1) ? bf_hi(wd[t >> 1]) : bf_lo(wd[t >> 1]);
                oa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[g], vf, oa[t], 0, 0, 0);
                ob[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[g], vf, ob[t], 0, 0, 0);
            }
        }
    }
#undef ATM_LDS_TURN
    // D layout: lane holds query rows 4*kk + r, output dims fi*DPL + t
#pragma unroll
    for (int tile = 0; tile < 2; tile++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = i0 + 16 * tile + 4 * fk + r;
            if (row >= S) continue;
            uint32_t w[DPL / 2];
#pragma unroll
            for (int t = 0; t < DPL; t += 2) w[t / 2] = tile ? ((uint32_t)bf_trunc(ob[t][r]) | ((uint32_t)bf_trunc(ob[t + 1][r]) << 16))
                                                             : ((uint32_t)bf_trunc(oa[t][r]) | ((uint32_t)bf_trunc(oa[t + 1][r]) << 16));
            uint16_t* dst = p.out + ((size_t)row * H + h) * HD + fi * DPL;
            if (DPL == 8) *(uint4*)dst = make_uint4(w[0], w[1], w[2 % (DPL / 2)], w[3 % (DPL / 2)]);
            else *(uint2*)dst = make_uint2(w[0], w[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Exact attention for one (head, query row): scores -> /sqrt(hd) -> mask -> f64 softmax -> PV.
// llamatransformer.go:409-514.  GQA head h reads KV head h/n_rep straight from the un-repeated cache
// (attentionRepeatKV :529-559 and the four Transposes :435-449 become index arithmetic).
// grid (H, S), block 512 = 8 waves (two per SIMD: a single wave issues at most one instruction per ~4.4 cycles):
//   scores : every thread owns one cached position j (512 per pass): the 128-long q.k chain (d ascending) over its K row, read
//            with 16 B loads that are contiguous across the wave (K cache layout [kv head][d/8][position][8]); the next pass's
//            rows are in flight during this pass's chains (register ping-pong); then exp in f64;
//   Z      : wave 0 walks sum_j exp(s_j) in f64, j ascending, 16 values per half-iteration with the other half's LDS reads in
//            flight and one counted wait per half (~9 cycles per dependent v_add_f64);
//   PV     : the role split of the GEMV -- waves 2,3,6,7 produce the EXACT products p_j*v[j,d] (bf16 x bf16) into a double-
//            buffered LDS ring [position][dim] (V rows come through a hand-counted asm prefetch ring three chunks deep);
//            waves 0,1 own one output dim per lane and only add, a whole 64-position chunk in flight per step.
// dynamic LDS: [T f64 e (+ zero pad)][T f32 p (+ zero pad)][hd f32 q][16 B][2 x 64 x hd f32 product ring]
// ------------------------------------------------------------------------------------------------
constexpr int ATT_JC = 64;                                  // cached positions per PV chunk
constexpr int ATT_RP = ATT_JC + 4;                          // product ring: [dim][position] f32, rows padded to 68 floats -- the adders read FOUR positions of their dim per
                                                            // ds_read_b128 (16 reads + 64 adds per chunk instead of 64 + 64: they were issue-bound), lanes 272 B apart: no conflict
__host__ __device__ inline size_t attn_off_pw(int seq_len) { return (((size_t)seq_len + 32) * 8 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t attn_off_q(int seq_len) { return attn_off_pw(seq_len) + ((((size_t)seq_len + ATT_JC) * 4 + 15) & ~(size_t)15); }
__host__ __device__ inline size_t attn_off_z(int seq_len, int hd) { return attn_off_q(seq_len) + (((size_t)hd * 4 + 15) & ~(size_t)15); }
__host__ __device__ inline size_t attn_off_ring(int seq_len, int hd) { return attn_off_z(seq_len, hd) + 128; }   // zb: Z | - | flag | - | 8 wave partials

constexpr int ATT_NT = 512;                                 // 8 waves: two per SIMD (one wave alone issues ~1 instruction / 4.4 cycles)
constexpr int ATT_NPROD = 4;                                // PV: waves 0,1 add; waves 2,3,6,7 produce; waves 4,5 share the adders' SIMDs and
                                                            // stay idle there (two issue-hungry waves on one SIMD halve each other)
constexpr int ATT_VU = ATT_JC / 4 / ATT_NPROD;              // 4-row units per producer per chunk

// V rows of PV chunk c for producer pwv.  One 16 B load covers 8 dims of one position: 16 lanes per row, 4 rows per instruction;
// a lane's four units are FOUR CONSECUTIVE positions c*64 + 4 (4 pwv + lane/16) + u of the same 8 dims (so that it can write its
// products as float4 runs along the position axis of the ring).  A PV step (~64 dependent adds) is several times shorter
// than a load round trip, so three chunks are kept in flight; hipcc drains vmcnt(0) around loop-carried register prefetch,
// hence the same hand-counted asm ring as the GEMV (RING_LOAD / RING_RETIRE, checked by tools/isa_audit.py).
// Always issued (rows clamped to T-1, whose p is +0 beyond the row) so that the count is the same on every path.
DEVINL void attn_load_v(u32x4 (&v)[ATT_VU], const uint16_t* vbase, uint32_t row_bytes, int c, int pwv, int lane, int T) {
    const uint32_t dq = (uint32_t)(lane & 15) * 16u;                      // byte offset of this lane's 8 dims
#pragma unroll
    for (int u = 0; u < ATT_VU; u++) {
        int j = c * ATT_JC + (pwv * 4 + (lane >> 4)) * 4 + u;
        j = j < T ? j : T - 1;
        const char* a = (const char*)vbase + ((size_t)(uint32_t)j * row_bytes + dq);
        asm volatile("global_load_dwordx4 %0, %1, off ; RING_LOAD" : "=&v"(v[u]) : "v"(a) : "memory");
    }
}
template <int N> DEVINL void attn_retire_v(u32x4 (&v)[ATT_VU]) {      // v's loads are done once at most N younger ones are outstanding
    static_assert(ATT_VU == 4, "operand list below");
    asm volatile("s_waitcnt vmcnt(%4) ; RING_RETIRE %0 %1 %2 %3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(N) : "memory");
}

// K rows of one scores pass: lane owns position j; chunk c of its row sits at kbase[c*seq_len + j] (position-contiguous)
template <int NK> DEVINL void attn_load_k(uint4 (&k)[NK], const uint4* kbase, int seq_len, int j) {
    const uint4* kr = kbase + j;
#pragma unroll
    for (int c = 0; c < NK; c++) { k[c] = *kr; kr += seq_len; }
}

// score + exp of one cached position (llamatransformer.go:456-473, Softmax impl:498)
template <int NK> DEVINL void attn_score(const uint4 (&k)[NK], const float* qf, int j, int T, int S, int i, float divisor, double* e, const double* exp_tab = nullptr) {
    if (j >= T) return;
    const bool masked = (S > 1) && ((j % S) > i);            // triu(-inf,1) broadcast by modulo (tensoriterators.go:47-55)
    double ev = 0.0;                                         // exp(-inf) == 0
    if (!masked) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < NK; c++) {                       // MatMul q.k, d ascending (operations_matmul.go:37-55)
            const float4 xa = *(const float4*)(qf + c * 8), xb = *(const float4*)(qf + c * 8 + 4);
            acc = mac8(acc, xa, xb, k[c]);
        }
        uint16_t s = bf_trunc(acc);
        if (exp_tab) ev = exp_tab[s];                        // the model's table of the three lines below over all 65536 raw scores (exp_table_kernel): the batched grids, where
                                                             // the vector unit is the bound (one stream alone: the ~130 inline instructions are faster than the load's round trip, measured)
        else {
            s = bf_trunc(__fdiv_rn(bf_wide(s), divisor));   // DivToScalar :464
            if (S > 1) s = bf_trunc(bf_wide(s) + 0.0f);      // Add(scores, mask) with mask==0 :469-473
            ev = exp((double)bf_wide(s));
        }
    }
    e[j] = ev;
}

// PV pipeline, one barrier per step on every wave: at step c the producers turn chunk c (rows already in vc) into exact products
// in ring slot c&1 ([position][dim] f32) after putting chunk c+3's rows in flight (vn); the adders add chunk c-1.
// The roles are split ONCE (three loops with the same barrier count) so that the ring's load/retire pairing is a property of
// straight-line code that tools/isa_audit.py can check.
template <int HD> DEVINL int attn_row_dim(int r) { return (r % (HD / 8)) * 8 + r / (HD / 8); }      // ring row r holds dim (r % (hd/8)) * 8 + r / (hd/8)
template <int HD, int D = 3> DEVINL void attn_pv_produce(int c, int nchunks, int pwv, int lane, int T, const uint16_t* vbase, uint32_t row_bytes,
                                              const float* pw, char* ring, u32x4 (&vc)[ATT_VU], u32x4 (&vn)[ATT_VU]) {
    constexpr int SLOT = ATT_RP * HD * 4;
    attn_load_v(vn, vbase, row_bytes, c + D, pwv, lane, T);
    attn_retire_v<D * ATT_VU>(vc);                           // chunks c+1 .. c+D stay in flight
    if (c < nchunks && (lane & 15) * 8 < HD) {
        const int pg = pwv * 4 + (lane >> 4);                // positions 4 pg .. 4 pg + 3 of the chunk
        // ring row of dim d: (d % 8) * (HD / 8) + d / 8 -- the 16 lanes of a write (same k, dim groups 0..15) hit 16 consecutive rows, 272 B
        // apart: 16 different 16-byte bank groups (row-major by dim: 8 rows apart, all in two bank groups -- measured 3x the whole kernel)
        float* dst = (float*)(ring + (c & 1) * SLOT) + (lane & 15) * ATT_RP + 4 * pg;
        const float4 pj = *(const float4*)(pw + c * ATT_JC + 4 * pg);
        const u32x4 a = vc[0], b4 = vc[1], c4 = vc[2], d4 = vc[3];    // exact products: 8-bit x 8-bit significands
        constexpr int KR = (HD / 8) * ATT_RP;                // row distance between k and k + 1
        *(float4*)(dst + 0 * KR) = make_float4(pj.x * bf_lo(a.x), pj.y * bf_lo(b4.x), pj.z * bf_lo(c4.x), pj.w * bf_lo(d4.x));
        *(float4*)(dst + 1 * KR) = make_float4(pj.x * bf_hi(a.x), pj.y * bf_hi(b4.x), pj.z * bf_hi(c4.x), pj.w * bf_hi(d4.x));
        *(float4*)(dst + 2 * KR) = make_float4(pj.x * bf_lo(a.y), pj.y * bf_lo(b4.y), pj.z * bf_lo(c4.y), pj.w * bf_lo(d4.y));
        *(float4*)(dst + 3 * KR) = make_float4(pj.x * bf_hi(a.y), pj.y * bf_hi(b4.y), pj.z * bf_hi(c4.y), pj.w * bf_hi(d4.y));
        *(float4*)(dst + 4 * KR) = make_float4(pj.x * bf_lo(a.z), pj.y * bf_lo(b4.z), pj.z * bf_lo(c4.z), pj.w * bf_lo(d4.z));
        *(float4*)(dst + 5 * KR) = make_float4(pj.x * bf_hi(a.z), pj.y * bf_hi(b4.z), pj.z * bf_hi(c4.z), pj.w * bf_hi(d4.z));
        *(float4*)(dst + 6 * KR) = make_float4(pj.x * bf_lo(a.w), pj.y * bf_lo(b4.w), pj.z * bf_lo(c4.w), pj.w * bf_lo(d4.w));
        *(float4*)(dst + 7 * KR) = make_float4(pj.x * bf_hi(a.w), pj.y * bf_hi(b4.w), pj.z * bf_hi(c4.w), pj.w * bf_hi(d4.w));
    }
    __syncthreads();
}
template <int HD> DEVINL void attn_pv_add(int c, int nchunks, int r, const char* ring, float& acc) {      // r: the lane's ring ROW (its dim: attn_row_dim)
    constexpr int SLOT = ATT_RP * HD * 4;
    if (c > 0 && c <= nchunks && r < HD) {
        // positions past the end of the row carry p == +0: their products are +-0 and acc is never -0, so whole chunks are added
        const float* src = (const float*)(ring + ((c - 1) & 1) * SLOT) + r * ATT_RP;
        float4 a[ATT_JC / 4];
#pragma unroll
        for (int j = 0; j < ATT_JC / 4; j++) a[j] = *(const float4*)(src + 4 * j);     // whole chunk in flight at once
#pragma unroll
        for (int j = 0; j < ATT_JC / 4; j++) acc = add4(acc, a[j]);
    }
    __syncthreads();
}

// p_j = trunc_bf16(f32(e / Z)) evaluated with an ESTIMATE Zt of the reference's serial f64 sum Z, and certified: any order of adding T
// non-negative doubles is within (T-1) 2^-53 of the exact sum, so Z lies in Zt (1 +- eps), eps = (4T + 8) 2^-53 -- and p is a step
// function of Z.  The quotient is formed as e * (1 / Zt) (within 2 ulps of e / Zt); it must not lie within delta32 = 4T + 12 ulps of the
// one point per bf16 cell where the result changes (low 45 mantissa bits = 2^45 - 2^28); f32-denormal quotients are certified by
// evaluating both ends of the interval.  Returns p; sets bad when the element cannot be certified (the caller then walks the serial sum).
// (Header of attn_long_pv_kernel / NOTES.md 5.9.)
struct CertZ { double rzt, zlo, zhi; unsigned delta32; };
DEVINL CertZ cert_z(double zt, int T) {
    const double epsr = (double)(4 * T + 8) * 1.1102230246251565e-16;
    CertZ c; c.rzt = 1.0 / zt; c.zlo = zt * (1.0 - epsr); c.zhi = zt * (1.0 + epsr); c.delta32 = 4u * (unsigned)T + 12u;
    return c;
}
DEVINL float cert_p(double e, const CertZ& c, int& bad) {
    const double q = e * c.rzt;
    const float pj = bf_wide(bf_trunc((float)q));
    const unsigned qh = (unsigned)__double2hiint(q), ql = (unsigned)__double2loint(q);
    const unsigned eq = (qh >> 20) & 0x7FFu;
    if (eq - 897u <= 126u) {                                 // f32-normal quotient (2^-126 <= q < 2): distance from the step point of its bf16 cell
        if ((qh & 0x1FFFu) == 0x1FFFu && ql - (0xF0000000u - c.delta32) <= 2u * c.delta32) bad = 1;
    } else if (q != 0.0) {                                   // f32 denormals, and anything that is not a finite in-range quotient (inf, NaN)
        const float plo = bf_wide(bf_trunc((float)(e / c.zlo))), phi = bf_wide(bf_trunc((float)(e / c.zhi)));
        if (__float_as_uint(plo) != __float_as_uint(phi) || __float_as_uint(pj) != __float_as_uint(plo)) bad = 1;
    }
    return pj;
}
// the reference's serial softmax denominator, walked by ONE wave: rowExpSum += exp(...), j ascending, f64 (impl:492-499).  16 values per half,
// the other half's LDS reads in flight, one wait per half; branch-free (e is zero-padded to a multiple of 32, and readable 16 values further)
DEVINL double attn_zseq_wave(const double* e, int T) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* e2 = (const d2*)e;
    double z = 0.0;
    d2 a[8], b[8];
#define ATT_TOUCH8(r) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]))
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = e2[u];
    ATT_TOUCH8(a);
    for (int j = 0; j < T; j += 32) {
#pragma unroll
        for (int u = 0; u < 8; u++) b[u] = e2[((j + 16) >> 1) + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) { z += a[u].x; z += a[u].y; }
        __builtin_amdgcn_sched_barrier(0);
        ATT_TOUCH8(b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = e2[((j + 32) >> 1) + u];      // may run past the row's padding: unused then
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) { z += b[u].x; z += b[u].y; }
        __builtin_amdgcn_sched_barrier(0);
        ATT_TOUCH8(a);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef ATT_TOUCH8
    return z;
}
// DENSE = the batched decode's heads x sequences grids (thousands of workgroups): registers for TWO workgroups per CU (4 waves per SIMD, 128
// VGPRs) instead of one -- the K rows of the next 512 positions are then NOT kept in flight during a pass's score chains (the ping-pong
// buffer is 64 of the 178 VGPRs of the single stream's form, which launches 32 workgroups on 256 CUs and wants every register)
template <int HD, bool DENSE = false> __global__ __launch_bounds__(ATT_NT, DENSE ? 4 : 2) void attn_exact_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = HD / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, i;
    // batched decode (DENSE): sequence-major inside an XCD -- the (H / 8) heads an XCD owns are the query heads of ONE KV head (H = 32, KVH = 8),
    // so the workgroups of one sequence's four heads are dispatched back to back on the same XCD and the K / V rows the first one pulls are
    // L2 hits for the other three (head-major order: 128 sequences x 92 KB between two readers of the same rows -- every read went to HBM)
    if (DENSE && !p.head_major) xcd_head_block_bmajor(h, i); else xcd_head_block(h, i);
    // batched decode: query row i is the one new token of SEQUENCE i of the batch -- its own position, caches and cache length
    const BatchTab* const bt = p.btab;
    const int S = bt ? 1 : p.S, KVH = p.KVH;
    const int seq_len = bt ? bt->seq_len[i] : p.seq_len;
    const int pos0 = bt ? bt->st[i]->pos : p.st->pos, T = pos0 + S;
    const int kvh = h / (p.H / KVH);
    double* e = (double*)smem;
    float* pw = (float*)(smem + attn_off_pw(p.lds_T));
    float* qf = (float*)(smem + attn_off_q(p.lds_T));
    double* zb = (double*)(smem + attn_off_z(p.lds_T, HD));
    char* ring = smem + attn_off_ring(p.lds_T, HD);
    const uint32_t row_bytes = (uint32_t)KVH * HD * 2;
    // K cache is stored [kv head][d/8][position][8] so that "one position per lane" reads are contiguous across the wave
    const uint4* kbase = (const uint4*)(bt ? p.bkv->ck[i] : p.cache_k) + (size_t)kvh * NK * seq_len;
    const uint16_t* vbase = (bt ? p.bkv->cv[i] : p.cache_v) + (size_t)kvh * HD;
    const int Tend = (S > 1 && pos0 == 0) ? (i + 1) : T;    // see PV below
    const int nchunks = (Tend + ATT_JC - 1) / ATT_JC;
#define ATT_STAMP(n) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) p.dbg[wave * 16 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    ATT_STAMP(0);

    // q first (everything waits on it), then what does not depend on q: this thread's K row and chunk 0 of V (producers)
    const uint16_t* q = p.q + ((size_t)i * p.H + h) * HD;
    const uint16_t q16 = q[tid < HD ? tid : 0];             // unconditional: a predicated load would be waited on at once
    uint4 ka[NK], kb[NK];
    attn_load_k<NK>(ka, kbase, seq_len, tid < seq_len ? tid : seq_len - 1);      // (clamped to the ARRAY end: the first pass's K rows do not wait for the position word; a row past T is never scored)
    if (tid < HD) qf[tid] = bf_wide(q16);
    __syncthreads();
    ATT_STAMP(1);

    // scores: ATT_NT positions per pass, the next pass's rows in flight during this pass's chains (registers ping-pong)
    if constexpr (DENSE) {
        for (int j0 = 0; j0 < T; j0 += ATT_NT) {
            const int j = j0 + tid;
            if (j0 > 0) attn_load_k<NK>(ka, kbase, seq_len, j < T ? j : T - 1);
            attn_score<NK>(ka, qf, j, T, S, i, p.divisor, e, DENSE ? p.exp_tab : nullptr);
        }
    } else
    for (int j0 = 0; j0 < T; j0 += 2 * ATT_NT) {
        const int j = j0 + tid;
        const bool more = j0 + ATT_NT < T, more2 = j0 + 2 * ATT_NT < T;       // block-uniform
        if (more) attn_load_k<NK>(kb, kbase, seq_len, j + ATT_NT < T ? j + ATT_NT : T - 1);
        attn_score<NK>(ka, qf, j, T, S, i, p.divisor, e, DENSE ? p.exp_tab : nullptr);
        if (more) {
            if (more2) attn_load_k<NK>(ka, kbase, seq_len, j + 2 * ATT_NT < T ? j + 2 * ATT_NT : T - 1);
            attn_score<NK>(kb, qf, j + ATT_NT, T, S, i, p.divisor, e);
        }
    }
    // zero padding so that the Z chain can run whole groups (x + 0.0 == x for x >= +0)
    for (int j = T + tid; j < ((T + 31) & ~31); j += ATT_NT) e[j] = 0.0;
    // hipcc's waitcnt pass is path-insensitive: on a (statically possible, dynamically impossible) path a K prefetch of the scores
    // loop is still pending, and it then drains vmcnt(0) in front of every PV step when a V register is reused.  An explicit
    // vmcnt(0) here (all K loads are consumed anyway) clears its scoreboard.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // producers: the first three PV chunks of V go in flight now and land during the Z chain
    const bool producer = (wave & 3) >= 2;
    const int pwv = (wave & 3) - 2 + ((wave >> 2) << 1);                   // waves 2,3,6,7 -> producers 0..3
    // (single stream: SEVEN chunks -- 448 positions, the whole row at the bench's contexts -- go in flight here, behind the scores: their K
    //  registers are free now, and with three chunks the producers waited for memory in every PV step: 7.4 k cycles of PV at T = 272,
    //  of which the adders' chains are 2.4 k.  The batched DENSE form has 128 registers: three chunks.)
    constexpr int VD = DENSE ? 3 : 7;
    u32x4 v0[ATT_VU], v1[ATT_VU], v2[ATT_VU], v3[ATT_VU], v4[ATT_VU], v5[ATT_VU], v6[ATT_VU], v7[ATT_VU];
    if (producer) {
        attn_load_v(v0, vbase, row_bytes, 0, pwv, lane, T);
        attn_load_v(v1, vbase, row_bytes, 1, pwv, lane, T);
        attn_load_v(v2, vbase, row_bytes, 2, pwv, lane, T);
        if constexpr (VD == 7) {
            attn_load_v(v3, vbase, row_bytes, 3, pwv, lane, T); attn_load_v(v4, vbase, row_bytes, 4, pwv, lane, T);
            attn_load_v(v5, vbase, row_bytes, 5, pwv, lane, T); attn_load_v(v6, vbase, row_bytes, 6, pwv, lane, T);
        }
    }
    ATT_STAMP(2);
    __syncthreads();
    ATT_STAMP(3);
    // ---- softmax denominator.  The reference adds the T exponentials one by one in f64 (impl:492-499): ~9 cycles per position on one
    // wave (1.1 us at T = 272, 2 us at 512).  As in the long-context kernels it is replaced by a tree estimate + CERTIFIED p_j
    // (cert_p); only a row that cannot be certified (or force_zseq) walks the serial sum.
    int* const zflag = (int*)(zb + 2);
    {
        double part = 0.0;
        for (int j = tid; j < T; j += ATT_NT) part += e[j];
        part = wave_sum_f64(part);
        if (lane == 0) zb[4 + wave] = part;
        if (tid == 0) *zflag = 0;
    }
    __syncthreads();
    const int Tp = (T + ATT_JC - 1) / ATT_JC * ATT_JC;
    {
        const double zt = ((zb[4] + zb[5]) + (zb[6] + zb[7])) + ((zb[8] + zb[9]) + (zb[10] + zb[11]));
        const CertZ cz = cert_z(zt, T);
        int bad = p.force_zseq;
        for (int j = tid; j < Tp; j += ATT_NT)               // impl:506 + ToBFloat16 :493 ; +0 padding up to the chunk
            pw[j] = j < T ? cert_p(e[j], cz, bad) : 0.0f;
        if (bad) *zflag = 1;
    }
    ATT_STAMP(4);
    __syncthreads();
    if (*zflag) {
        if (wave == 0) {                                     // rowExpSum += exp(...), j ascending, f64 (impl:492-499)
            const double z = attn_zseq_wave(e, T);
            if (lane == 0) { zb[0] = z; if (p.zseq_count && tid == 0) atomicAdd(p.zseq_count, 1); }
        }
        __syncthreads();
        const double z = zb[0];
        for (int j = tid; j < T; j += ATT_NT) pw[j] = bf_wide(bf_trunc((float)(e[j] / z)));
    }
    __syncthreads();
    ATT_STAMP(5);

    // PV: out[d] = trunc(sum_{j ascending} p_j * v[j,d]).  Masked columns have p == +0 and acc is never -0, so with the
    // standard causal layout (pos0 == 0) the chain can stop at the chunk that holds the diagonal (Tend).
    float acc = 0.0f;
    if (producer) {
        if constexpr (VD == 7) {
            for (int c = 0; c <= nchunks; c += 8) {          // eight steps per trip: the V register sets rotate without copies
                attn_pv_produce<HD, 7>(c, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v0, v7);
                attn_pv_produce<HD, 7>(c + 1, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v1, v0);
                attn_pv_produce<HD, 7>(c + 2, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v2, v1);
                attn_pv_produce<HD, 7>(c + 3, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v3, v2);
                attn_pv_produce<HD, 7>(c + 4, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v4, v3);
                attn_pv_produce<HD, 7>(c + 5, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v5, v4);
                attn_pv_produce<HD, 7>(c + 6, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v6, v5);
                attn_pv_produce<HD, 7>(c + 7, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v7, v6);
            }
        } else
        for (int c = 0; c <= nchunks; c += 4) {              // four steps per trip: the V register sets rotate without copies
            attn_pv_produce<HD>(c, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v0, v3);
            attn_pv_produce<HD>(c + 1, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v1, v0);
            attn_pv_produce<HD>(c + 2, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v2, v1);
            attn_pv_produce<HD>(c + 3, nchunks, pwv, lane, T, vbase, row_bytes, pw, ring, v3, v2);
        }
    } else if (wave < 2) {
        const int d = wave * 64 + lane;
        for (int c = 0; c <= nchunks; c += (VD + 1)) {       // (the same number of barriers as the producers' trips)
            attn_pv_add<HD>(c, nchunks, d, ring, acc); attn_pv_add<HD>(c + 1, nchunks, d, ring, acc);
            attn_pv_add<HD>(c + 2, nchunks, d, ring, acc); attn_pv_add<HD>(c + 3, nchunks, d, ring, acc);
            if constexpr (VD == 7) {
                attn_pv_add<HD>(c + 4, nchunks, d, ring, acc); attn_pv_add<HD>(c + 5, nchunks, d, ring, acc);
                attn_pv_add<HD>(c + 6, nchunks, d, ring, acc); attn_pv_add<HD>(c + 7, nchunks, d, ring, acc);
            }
        }
    } else {
        for (int c = 0; c <= nchunks; c += (VD + 1)) {
            __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
            if constexpr (VD == 7) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");   // prefetches past the last chunk
    ATT_STAMP(6);
    if (wave < 2) {
        const int r = wave * 64 + lane;
        if (r < HD) {
            const int d = attn_row_dim<HD>(r);
            if (bt && p.out_xt) p.out_xt[xt_group(i, p.H * HD) + xt_index(i & 15, h * HD + d)] = bf_trunc(acc);   // column batches: straight into the B-operand layout of the wo product
            else p.out[((size_t)i * p.H + h) * HD + d] = bf_trunc(acc);        // [S, H*hd] (:508-514)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attn_gqa_kernel<HD, G>: the batched decode's attention, one workgroup per (KV head, sequence) serving the G = H / KVH = 4 query heads
// that share the KV head (round 4).  attn_exact_kernel's grid is (H, sequences): at 128 sequences 4096 workgroups of three dependent
// memory round trips each (q -> K rows -> V rows), two resident per CU -- 63 us per layer -- and the vector unit does everything one
// lane-operation at a time (a wave64 instruction occupies its SIMD for four cycles: with two workgroups per CU the kernel is bound by
// the NUMBER of vector instructions, not by memory: profiles/r04_gqa_stamps.log).  Here:
//   scores : a thread owns (cached position, head PAIR): its K row feeds both heads' chains through ONE v_pk_fma_f32 per dimension
//            (q staged in the LDS as (head 2a, head 2a+1) pairs; the same fused multiply-add per lane as mac8); 256 positions per pass;
//   exp    : one 8-byte load from the model's table of exp(trunc(s / sqrt(hd))) over all 65536 bf16 scores (exp_table_kernel: filled by
//            the very instruction sequence attn_exact_kernel evaluates inline -- ~130 f64 instructions per score);
//   Z      : tree estimate + certified p_j per head (cert_p); a head that cannot be certified walks the reference's serial sum on its wave;
//   PV     : V rows are staged ONCE per workgroup (waves 4..7: raw bf16, 32 positions per chunk, four chunks in flight in registers and a
//            ring of four LDS buffers, one barrier per chunk); wave g < 4 adds head g, a lane owns two neighbouring dims: one ds_read_b32,
//            two unpack operations, v_pk_mul_f32 and v_pk_add_f32 per position (multiply, then add: bf16 x bf16 is exact in f32 and the
//            reference rounds only the sum; j ascending) -- the adders do nothing else, the stagers nothing but copy.
// Same bits as attn_exact_kernel per head.  dynamic LDS: [G][Tp] f64 e | [G][Tp] f32 p | q pairs | zb | [4][32][HD] bf16 V.
// grid (KVH, sequences), block 512, two workgroups per CU.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int gqa_tp(int lds_T) { return ((lds_T + 63) / 64) * 64 + 64; }
__host__ __device__ inline size_t gqa_lds_bytes(int lds_T, int hd, int G) { return (size_t)G * gqa_tp(lds_T) * 12 + (size_t)G * hd * 4 + 512 + 4 * 32 * (size_t)hd * 2; }
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// adder side of attn_gqa_kernel: a quarter of a V chunk (8 positions of the lane's bf16 pair, ring buffer (q / 4) & 3) and its 8 p values
template <int HD> DEVINL void gqa_read_piece(uint32_t (&w)[8], float4 (&pq)[2], const char* vl, const float* pg, int q) {
    const char* vb = vl + ((q >> 2) & 3) * (32 * HD * 2) + (q & 3) * (8 * HD * 2);
    pq[0] = *(const float4*)(pg + q * 8); pq[1] = *(const float4*)(pg + q * 8 + 4);           // (wave-uniform addresses: broadcast reads)
#pragma unroll
    for (int u = 0; u < 8; u++) w[u] = *(const uint32_t*)(vb + u * (HD * 2));
}
DEVINL void gqa_add_piece(f32x2_t& a, const uint32_t (&w)[8], const float4 (&pq)[2]) {      // acc += p_j * v[j][d], j ascending: multiply, then add (:508-514)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        a += f32x2_t{pq[k].x, pq[k].x} * f32x2_t{bf_lo(w[4 * k]), bf_hi(w[4 * k])};
        a += f32x2_t{pq[k].y, pq[k].y} * f32x2_t{bf_lo(w[4 * k + 1]), bf_hi(w[4 * k + 1])};
        a += f32x2_t{pq[k].z, pq[k].z} * f32x2_t{bf_lo(w[4 * k + 2]), bf_hi(w[4 * k + 2])};
        a += f32x2_t{pq[k].w, pq[k].w} * f32x2_t{bf_lo(w[4 * k + 3]), bf_hi(w[4 * k + 3])};
    }
}
template <int HD, int G> __global__ __launch_bounds__(512, 4) void attn_gqa_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = HD / 8, NT = 512, NPOS = 256, VC = 32;      // a scores pass: 256 positions x 2 head pairs; V chunks of 32 positions
    static_assert(G == 4 && HD == 128, "two head pairs; one V unit per thread and chunk; two adder waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kvh = (int)blockIdx.x, i = (int)blockIdx.y;      // (dispatch order: the eight KV heads of a sequence go to the eight XCDs)
    const BatchTab* const bt = p.btab;
    const int seq_len = bt->seq_len[i], T = bt->st[i]->pos + 1, Tp = gqa_tp(p.lds_T);
    double* e = (double*)smem;                                  // [G][Tp]
    float* pw = (float*)(smem + (size_t)G * Tp * 8);            // [G][Tp]
    float* qi = (float*)(smem + (size_t)G * Tp * 12);           // [G / 2][HD][2]: (head 2a, head 2a+1) pairs
    double* zb = (double*)(smem + (size_t)G * Tp * 12 + (size_t)G * HD * 4);     // per head: 4 wave partials | - | Z | flag
    uint16_t* vs = (uint16_t*)(smem + (size_t)G * Tp * 12 + (size_t)G * HD * 4 + 512);   // ring [4][VC][HD] bf16
    const uint32_t row_bytes = (uint32_t)p.KVH * HD * 2;
    const uint4* kbase = (const uint4*)p.bkv->ck[i] + (size_t)kvh * NK * seq_len;
    const uint16_t* vbase = p.bkv->cv[i] + (size_t)kvh * HD;
    const int jl = tid & (NPOS - 1), hp = wave >> 2, pwv = wave & 3;       // position inside a pass, head pair, wave inside the pair
    const int nch = (T + VC - 1) / VC, Tc = nch * VC;
#define GQA_STAMP(n) do { if (p.dbg && blockIdx.x == 3 && blockIdx.y == gridDim.y / 2 && lane == 0) p.dbg[wave * 16 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    GQA_STAMP(0);

    // q of the G heads (contiguous in the row) as head pairs, and this thread's K row of the first pass
    const uint16_t q16 = p.q[((size_t)i * p.H + (size_t)kvh * G) * HD + tid];
    uint4 ka[NK];
    if (pwv * 64 < T) attn_load_k<NK>(ka, kbase, seq_len, jl < T ? jl : T - 1);
    { const int g = tid >> 7, d = tid & (HD - 1); qi[((g >> 1) * HD + d) * 2 + (g & 1)] = bf_wide(q16); }
    __syncthreads();
    GQA_STAMP(1);
    // ---- scores: one cached position per thread and pass, both heads of the pair over the same K row (llamatransformer.go:456-473)
    const float* qh = qi + hp * HD * 2;
    for (int j0 = 0; j0 < T; j0 += NPOS) {
        const int j = j0 + jl;
        if (j0 + pwv * 64 >= T) continue;                       // (wave-uniform: a wave past the end of the row leaves the issue slots to the others)
        if (j0 > 0) attn_load_k<NK>(ka, kbase, seq_len, j < T ? j : T - 1);
        f32x2_t acc = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < NK; c++) {                          // MatMul q.k, d ascending (operations_matmul.go:37-55): acc = fma(q_d, k_d, acc) per head
            const float4 qa = *(const float4*)(qh + c * 16), qb = *(const float4*)(qh + c * 16 + 4), qc = *(const float4*)(qh + c * 16 + 8), qd = *(const float4*)(qh + c * 16 + 12);
            const uint4 k = ka[c];
            float kk;
            kk = bf_lo(k.x); acc = __builtin_elementwise_fma(f32x2_t{qa.x, qa.y}, f32x2_t{kk, kk}, acc);
            kk = bf_hi(k.x); acc = __builtin_elementwise_fma(f32x2_t{qa.z, qa.w}, f32x2_t{kk, kk}, acc);
            kk = bf_lo(k.y); acc = __builtin_elementwise_fma(f32x2_t{qb.x, qb.y}, f32x2_t{kk, kk}, acc);
            kk = bf_hi(k.y); acc = __builtin_elementwise_fma(f32x2_t{qb.z, qb.w}, f32x2_t{kk, kk}, acc);
            kk = bf_lo(k.z); acc = __builtin_elementwise_fma(f32x2_t{qc.x, qc.y}, f32x2_t{kk, kk}, acc);
            kk = bf_hi(k.z); acc = __builtin_elementwise_fma(f32x2_t{qc.z, qc.w}, f32x2_t{kk, kk}, acc);
            kk = bf_lo(k.w); acc = __builtin_elementwise_fma(f32x2_t{qd.x, qd.y}, f32x2_t{kk, kk}, acc);
            kk = bf_hi(k.w); acc = __builtin_elementwise_fma(f32x2_t{qd.z, qd.w}, f32x2_t{kk, kk}, acc);
        }
        if (j < T) {                                            // / sqrt(hd) :464, exp impl:498: tabulated over the raw bf16 score
            const double e0 = p.exp_tab[bf_trunc(acc.x)], e1 = p.exp_tab[bf_trunc(acc.y)];
            e[(hp * 2) * Tp + j] = e0; e[(hp * 2 + 1) * Tp + j] = e1;
        }
    }
    GQA_STAMP(2);
    // V chunks (32 positions x 256 B, raw bf16): staged by waves 4..7, two 16-byte units per thread and chunk, rows clamped to T-1 (p == +0
    // there).  Four chunks in flight in registers + a ring of four LDS buffers: the adders (waves 0..3) never wait for memory.  Hand-counted
    // asm ring (hipcc drains vmcnt(0) around loop-carried register prefetch): RING_LOAD / RING_RETIRE, checked by tools/isa_audit.py.
    const bool stager = wave >= 4;
    const int st_t = tid - 256;
    u32x4 s0[2], s1[2], s2[2], s3[2];
#define GQA_VLOAD(c, R_) do { int ja_ = (c) * VC + (st_t >> 4), jb_ = ja_ + 16; ja_ = ja_ < T ? ja_ : T - 1; jb_ = jb_ < T ? jb_ : T - 1; \
        const char* aa_ = (const char*)vbase + ((size_t)(uint32_t)ja_ * row_bytes + (uint32_t)(st_t & 15) * 16u); \
        const char* ab_ = (const char*)vbase + ((size_t)(uint32_t)jb_ * row_bytes + (uint32_t)(st_t & 15) * 16u); \
        asm volatile("global_load_dwordx4 %0, %1, off ; RING_LOAD" : "=&v"(R_[0]) : "v"(aa_) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, off ; RING_LOAD" : "=&v"(R_[1]) : "v"(ab_) : "memory"); } while (0)
#define GQA_VSTORE(c, R_, N_) do { asm volatile("s_waitcnt vmcnt(%2) ; RING_RETIRE %0 %1" : "+v"(R_[0]), "+v"(R_[1]) : "n"(N_) : "memory"); \
        char* dst_ = (char*)vs + ((c) & 3) * (VC * HD * 2) + (st_t >> 4) * (HD * 2) + (st_t & 15) * 16; \
        *(u32x4*)dst_ = R_[0]; *(u32x4*)(dst_ + 16 * HD * 2) = R_[1]; } while (0)
    if (stager) { GQA_VLOAD(0, s0); GQA_VLOAD(1, s1); GQA_VLOAD(2, s2); GQA_VLOAD(3, s3); }     // they land during the denominators
    for (int j = T + tid; j < ((T + 31) & ~31); j += NT) {      // zero padding: the serial walk runs whole groups of 32
#pragma unroll
        for (int g = 0; g < G; g++) e[g * Tp + j] = 0.0;
    }
    __syncthreads();
    GQA_STAMP(3);
    // ---- softmax denominators: tree estimate + certified p_j per head (header of attn_long_pv_kernel)
    {
        double part[2] = {0.0, 0.0};
        for (int j = jl; j < T; j += NPOS) { part[0] += e[(hp * 2) * Tp + j]; part[1] += e[(hp * 2 + 1) * Tp + j]; }
#pragma unroll
        for (int g = 0; g < 2; g++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part[g] += __shfl_xor(part[g], o);
            if (lane == 0) zb[(hp * 2 + g) * 10 + pwv] = part[g];
        }
        if (tid < G) *(int*)(zb + tid * 10 + 9) = 0;
    }
    __syncthreads();
    {
#pragma unroll 1
        for (int g = hp * 2; g < hp * 2 + 2; g++) {             // (not unrolled: inlined copies of cert_p cost registers)
            const double* z4 = zb + g * 10;
            const double zt = (z4[0] + z4[1]) + (z4[2] + z4[3]);
            const CertZ cz = cert_z(zt, T);
            int bad = p.force_zseq;
            for (int j = jl; j < Tc; j += NPOS) pw[g * Tp + j] = j < T ? cert_p(e[g * Tp + j], cz, bad) : 0.0f;   // impl:506 + ToBFloat16 :493; +0 up to the chunk
            if (bad) *(int*)(zb + g * 10 + 9) = 1;
        }
    }
    GQA_STAMP(4);
    if (stager) {                                               // chunks 0..2 into the ring, chunks 4..6 in flight
        GQA_VSTORE(0, s0, 6); GQA_VLOAD(4, s0);
        GQA_VSTORE(1, s1, 6); GQA_VLOAD(5, s1);
        GQA_VSTORE(2, s2, 6); GQA_VLOAD(6, s2);
    }
    __syncthreads();
    GQA_STAMP(5);
    {
        int flags = 0;
#pragma unroll
        for (int g = 0; g < G; g++) flags |= *(const int*)(zb + g * 10 + 9) << g;
        if (flags) {                                            // (block-uniform) a head that could not be certified: the reference's serial sum, on wave g
            // (the serial walk wants 64 registers: the stagers' V rows must have LANDED before anything may be spilled around it)
            if (stager) asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1[0]), "+v"(s1[1]), "+v"(s2[0]), "+v"(s2[1]), "+v"(s3[0]), "+v"(s3[1]) :: "memory");
            if (wave < G && ((flags >> wave) & 1)) {
                const double z = attn_zseq_wave(e + wave * Tp, T);
                if (lane == 0) { zb[wave * 10 + 8] = z; if (p.zseq_count) atomicAdd(p.zseq_count, 1); }
            }
            __syncthreads();
#pragma unroll 1
            for (int g = 0; g < G; g++) {
                if (!((flags >> g) & 1)) continue;
                const double z = zb[g * 10 + 8];
                for (int j = tid; j < T; j += NT) pw[g * Tp + j] = bf_wide(bf_trunc((float)(e[g * Tp + j] / z)));
            }
            __syncthreads();
        }
    }
    // ---- PV: out[g][d] = trunc(sum_{j ascending} p[g][j] * v[j][d]) (:508-514).  Wave g < 4 adds head g, lane l the dims 2l, 2l+1: one
    // ds_read_b32 (a bf16 pair), two unpack operations, v_pk_mul_f32, v_pk_add_f32 per position.  One barrier per chunk: at step c the adders
    // read ring buffer c & 3 while the stagers write chunk c+3 (into the buffer read at step c-1) and put chunk c+7 in flight.
    f32x2_t a0 = {0.0f, 0.0f};
    if (stager) {
#define GQA_SSTEP(c, R_) do { if ((c) < nch) { GQA_VSTORE((c) + 3, R_, 6); GQA_VLOAD((c) + 7, R_); __syncthreads(); } } while (0)
        for (int c = 0; c < nch; c += 4) { GQA_SSTEP(c, s3); GQA_SSTEP(c + 1, s0); GQA_SSTEP(c + 2, s1); GQA_SSTEP(c + 3, s2); }
#undef GQA_SSTEP
        asm volatile("s_waitcnt vmcnt(0) ; RING_RETIRE_ALL" ::: "memory");    // prefetches past the last chunk
        GQA_STAMP(6);
        return;                                                 // (the roles never join again: the ring registers are dead on the adders' path)
    }
    {
        // (a quarter chunk -- 8 positions: 8 V words + 2 x 4 p -- is read from the LDS while the quarter before it is added: left to hipcc, every
        //  ds_read sat right in front of its use, ~100 cycles of LDS latency per two positions.  The first quarter of chunk c+1 is read during
        //  step c: chunks up to c+2 are complete then.)
        const float* pg = pw + wave * Tp;
        const char* vl = (const char*)vs + lane * 4;
        uint32_t wA[8], wB[8]; float4 pA[2], pB[2];
        gqa_read_piece<HD>(wA, pA, vl, pg, 0);
        for (int q = 0; q < 4 * nch; q += 4) {
            gqa_read_piece<HD>(wB, pB, vl, pg, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            gqa_add_piece(a0, wA, pA);
            __builtin_amdgcn_sched_barrier(0);
            gqa_read_piece<HD>(wA, pA, vl, pg, q + 2);
            __builtin_amdgcn_sched_barrier(0);
            gqa_add_piece(a0, wB, pB);
            __builtin_amdgcn_sched_barrier(0);
            gqa_read_piece<HD>(wB, pB, vl, pg, q + 3);
            __builtin_amdgcn_sched_barrier(0);
            gqa_add_piece(a0, wA, pA);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 4 < 4 * nch) gqa_read_piece<HD>(wA, pA, vl, pg, q + 4);
            __builtin_amdgcn_sched_barrier(0);
            gqa_add_piece(a0, wB, pB);
            __syncthreads();
        }
    }
    GQA_STAMP(6);
#undef GQA_STAMP
#undef GQA_VLOAD
#undef GQA_VSTORE
    {
        const int h = kvh * G + wave, d = 2 * lane;
        if (p.out_xt) { uint16_t* o = p.out_xt + xt_group(i, p.H * HD); o[xt_index(i & 15, h * HD + d)] = bf_trunc(a0.x); o[xt_index(i & 15, h * HD + d + 1)] = bf_trunc(a0.y); }     // column batches: the B-operand layout of the wo product
        else *(uint32_t*)(p.out + ((size_t)i * p.H + h) * HD + d) = (uint32_t)bf_trunc(a0.x) | ((uint32_t)bf_trunc(a0.y) << 16);       // [S, H*hd]
    }
}

// ------------------------------------------------------------------------------------------------
// Long-context decode attention (S == 1, thousands of cached positions): the same arithmetic as attn_exact_kernel, spread over the
// chip.  attn_exact_kernel is one workgroup per head: at T = 4100 its scores take 12 us, the serial f64 Z chain 16 us and PV 31 us
// (one CU pulls a whole head's V rows).  Here:
//   attn_long_scores_kernel  grid (H, ceil(seq_len / 256)): one cached position per thread, the 128-long q.k chain, /sqrt(hd), exp
//       in f64 -> e_buf[h][j]; plus a tree sum of the block's 256 values -> z_part (an ESTIMATE of the row sum, see below);
//   attn_long_pv_kernel      grid (H, hd / 16): p_j for the whole row, then out[d] = sum_j p_j v[j][d] for 16 output dims with the
//       role split of the short kernel (four producer waves: V rows -> exact products in an LDS ring, one adder wave: one chain per
//       lane, j ascending).  256 workgroups instead of 32, each pulls 32 B per position instead of 256 B.
// Z WITHOUT the serial chain, still bit-exact:  the reference adds the T exponentials one by one in f64 (operations_impl.go:492-499).
// Every way of summing T non-negative doubles is within (T-1) u of the exact sum (u = 2^-53), so the tree estimate Zt and the
// reference's Zs differ by less than eps = 4 T u relatively -- and p_j = trunc_bf16(f32(e_j / Z)) is a MONOTONE step function of Z
// whose steps are 2^-8 wide relative to p.  So p_j is evaluated with Zt and CERTIFIED: the f64 quotient q must not lie within
// 4 T + 4 f64 ulps of the one point per bf16 cell where the result changes (the f32 rounding boundary just below a bf16 grid
// point: low 45 mantissa bits == 2^45 - 2^28); quotients in the f32 denormal range are certified by evaluating the step function at
// both ends of [Zt (1 - eps), Zt (1 + eps)].  If every p_j of the row is certified, they equal the reference's bits whatever Zs is
// (monotonicity); otherwise -- about once in 10^9 elements -- the workgroup walks the reference's serial sum itself (16 us, exact)
// and re-evaluates.  force_zseq runs that path always (tests/test_gpu_configs.py compares both with the oracle).
// ------------------------------------------------------------------------------------------------
constexpr int ALS_NT = 256;                                  // positions per scores workgroup
template <int NK> DEVINL double attn_score_value(const uint4 (&k)[NK], const float* qf, float divisor) {   // attn_score for S == 1 (no mask)
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < NK; c++) {                           // MatMul q.k, d ascending (operations_matmul.go:37-55)
        const float4 xa = *(const float4*)(qf + c * 8), xb = *(const float4*)(qf + c * 8 + 4);
        acc = mac8(acc, xa, xb, k[c]);
    }
    uint16_t s = bf_trunc(acc);
    s = bf_trunc(__fdiv_rn(bf_wide(s), divisor));           // DivToScalar :464
    return exp((double)bf_wide(s));                          // Softmax impl:498
}
template <int HD> __global__ __launch_bounds__(ALS_NT) void attn_long_scores_kernel(AttnParams p) {
    constexpr int NK = HD / 8;
    __shared__ __attribute__((aligned(16))) float qf[HD];
    __shared__ double wsum[ALS_NT / 64];
    const int tid = threadIdx.x;
    int h, blk; xcd_head_block(h, blk);
    const int T = p.st->pos + 1, j0 = blk * ALS_NT;
    if (j0 >= T) return;                                     // (uniform: the grid is sized for seq_len, the captured graph serves every T)
    const int kvh = h / (p.H / p.KVH);
    const uint4* kbase = (const uint4*)p.cache_k + (size_t)kvh * NK * p.seq_len;
    const uint16_t* q = p.q + (size_t)h * HD;
    const uint16_t q16 = q[tid < HD ? tid : 0];
    const int j = j0 + tid;
    uint4 k[NK];
    attn_load_k<NK>(k, kbase, p.seq_len, j < T ? j : T - 1);
    // round 6 (AttnParams.touch): the query heads of a GQA group touch the V rows of their block for the PV launch that follows -- these workgroups sit on the
    // XCD whose L2 the PV workgroups of the group will read (xcd_head_block), so the lines land where they are needed (a touch from another XCD lands in the wrong L2, and the
    // memory-side cache is barely faster than HBM: profiles/r06_kv_touch_ab.log).  Issued BEHIND the K rows.
    // configs[2] step: attention 25.5 -> 23.7 us per layer (profiles/r06_attn_touch_ab.log)
    // One plain load per thread, straight-line (a `volatile` access compiles to a system-scope load with a full vmcnt(0) wait behind it, a conditional one makes the
    // destination a phi the compiler may copy in flight): the G heads of the group share the block's 256 * HD / 64 lines, the surplus heads repeat the first ones' (L2 hits);
    // the destination register is held until the end of the kernel by the empty asm below, the data is never looked at.
    unsigned vt;
    {
        constexpr int LPR = HD / 64 > 0 ? HD / 64 : 1;       // 128-byte lines per (row, kv head)
        const int l = ((h % (p.H / p.KVH)) % LPR) * ALS_NT + tid;
        int jl = j0 + l / LPR; jl = jl < T ? jl : T - 1;
        const char* va = (const char*)p.cache_v + ((size_t)jl * p.KVH + kvh) * HD * 2 + (l % LPR) * 128;
        va = p.touch ? va : (const char*)kbase;              // (off: a line this workgroup reads anyway)
        asm volatile("global_load_dword %0, %1, off ; RING_LOAD (never retired: tools/isa_audit.py flags any later use of the register)" : "=v"(vt) : "v"(va));
    }
    if (tid < HD) qf[tid] = bf_wide(q16);
    __syncthreads();
    double ev = 0.0;
    if (j < T) { ev = attn_score_value<NK>(k, qf, p.divisor); p.e_buf[(size_t)h * p.seq_len + j] = ev; }
    // tree sum of the block (fixed shape: deterministic); only ever used as an estimate with a rigorous error bound
    ev = wave_sum_f64(ev);
    if ((tid & 63) == 0) wsum[tid >> 6] = ev;
    __syncthreads();
    if (tid == 0) p.z_part[(size_t)h * ((p.seq_len + ALS_NT - 1) / ALS_NT) + blk] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    asm volatile("" :: "v"(vt));                              // the touch's destination stays allocated up to here
}

constexpr int ALP_DS = 16;                                   // output dims per workgroup
constexpr int ALP_BATCH = 512;                               // cached positions per PV batch (one barrier per batch) = four 128-position chunks
constexpr int ALP_NT = 512;                                  // waves 0..3: adders (4 dims each, one per DPP row); waves 4..7: producers
constexpr int ALP_SLOT = ALP_DS * ALP_BATCH;                 // floats per ring slot: [dim][chunk][half][position % 16][4]
constexpr int ALP_EU = 12;                                   // e_j per thread and round
__host__ __device__ inline size_t alp_lds_bytes(int seq_len) { return (size_t)((seq_len + ALP_BATCH - 1) / ALP_BATCH + 1) * ALP_BATCH * 4 + 2 * (size_t)ALP_SLOT * 4 + 64; }
DEVINL float alp_p(double e, double z) { return bf_wide(bf_trunc((float)(e / z))); }     // impl:506 + ToBFloat16 :493
template <int HD> DEVINL void alp_eager_body(const AttnParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, ds; xcd_head_block(h, ds);
    const int T = p.st->pos + 1, nblk = (T + ALS_NT - 1) / ALS_NT, nbatch = (T + ALP_BATCH - 1) / ALP_BATCH;
    const int Tpad = (nbatch + 1) * ALP_BATCH;               // (+ one batch of zeros: the producers run ahead)
    float* pw = (float*)smem;                                // [Tpad] p_j (+0 beyond T)
    float* ring = (float*)(smem + (size_t)((p.seq_len + ALP_BATCH - 1) / ALP_BATCH + 1) * ALP_BATCH * 4);   // [2][ALP_SLOT] products
    double* zsh = (double*)(ring + 2 * ALP_SLOT);
    const double* E = p.e_buf + (size_t)h * p.seq_len;
    const int kvh = h / (p.H / p.KVH);
#define ALP_STAMP(n) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0) p.dbg[(wave >> 2) * 16 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    ALP_STAMP(0);
    // ---- everything that depends only on addresses goes in flight first, oldest-needed first (waits retire in issue order): the first
    // round of e_j, the per-block partial sums, then the producers' first three batches of V rows (HBM: the slowest, needed last)
    // (clamped to the ARRAY ends, not to T: the addresses must not wait for the position word to arrive; what lies past T is not used)
    double ev[ALP_EU];
#pragma unroll
    for (int u = 0; u < ALP_EU; u++) { const int j = u * ALP_NT + tid; ev[u] = E[j < p.seq_len ? j : p.seq_len - 1]; }
    const int nblk_max = (p.seq_len + ALS_NT - 1) / ALS_NT;
    const double* zp = p.z_part + (size_t)h * nblk_max;
    double zmine = zp[lane < nblk_max ? lane : nblk_max - 1];
    const uint16_t* vbase = p.cache_v + (size_t)kvh * HD + (size_t)ds * ALP_DS;
    const size_t vrow = (size_t)p.KVH * HD;
    const int pl = tid & 255, half8 = pl & 1, prow = pl >> 1;                // producers: 128 positions x two 8-dim halves per round, 4 rounds per batch
    auto load = [&](uint4 (&v)[4], int b) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int j = b * ALP_BATCH + r * 128 + prow; j = j < T ? j : T - 1;
            v[r] = *(const uint4*)(vbase + (size_t)j * vrow + half8 * 8);
        }
    };
    uint4 v0[4], v1[4], v2[4], v3[4];
    if (wave >= 4) { load(v0, 0); load(v1, 1); load(v2, 2); }
    // ---- Z estimate: the per-block tree sums, added in block order (same value in every thread): one partial per lane, a register walk
    double zt = 0.0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        if (b0) zmine = zp[b0 + lane < nblk ? b0 + lane : nblk - 1];
        const int nb = nblk - b0 < 64 ? nblk - b0 : 64;
        const long long zbits = __double_as_longlong(zmine);
        for (int b = 0; b < nb; b++) {                       // v_readlane (a shuffle is an LDS round trip per step: 4 k cycles for 17 blocks)
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)zbits, b), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(zbits >> 32), b);
            zt += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
    }
    // ---- p_j with the estimate, certified (header comment).  The quotient is formed as e * (1 / Zt): within 2 ulps of the correctly
    // rounded e / Zt, which the certification band absorbs (its half-width 4 T + 12 ulps instead of 4 T + 4)
    const unsigned delta32 = 4u * (unsigned)T + 12u;         // (T < 2^20: far below the 2^28 between a cell's step point and its end)
    const double epsr = (double)(4 * T + 8) * 1.1102230246251565e-16;          // relative half-width of the interval that holds the reference's Z
    const double zlo = zt * (1.0 - epsr), zhi = zt * (1.0 + epsr), rzt = 1.0 / zt;
    int bad = p.force_zseq;
    int* const flag = (int*)(zsh + 1);                       // (own flag instead of __syncthreads_or: its library reduction takes static LDS)
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int j0 = 0; j0 < Tpad; j0 += ALP_EU * ALP_NT) {
        if (j0 != 0) {
#pragma unroll
            for (int u = 0; u < ALP_EU; u++) { const int j = j0 + u * ALP_NT + tid; ev[u] = E[j < T ? j : T - 1]; }
        }
#pragma unroll
        for (int u = 0; u < ALP_EU; u++) {
            const int j = j0 + u * ALP_NT + tid;
            if (j >= Tpad) continue;
            float pj = 0.0f;
            if (j < T) {
                const double e = ev[u];
                const double q = e * rzt;
                pj = bf_wide(bf_trunc((float)q));
                // the band test on the two 32-bit halves of q (64-bit integer code made this loop the longest phase of the kernel): the low 45
                // mantissa bits lie within delta of 2^45 - 2^28  <=>  bits 32..44 are all ones and the low word is within delta of 0xF0000000
                const unsigned qh = (unsigned)__double2hiint(q), ql = (unsigned)__double2loint(q);
                const unsigned eq = (qh >> 20) & 0x7FFu;
                if (eq - 897u <= 126u) {                     // f32-normal quotient (2^-126 <= q < 2): distance from the step point of its bf16 cell
                    if ((qh & 0x1FFFu) == 0x1FFFu && ql - (0xF0000000u - delta32) <= 2u * delta32) bad = 1;
                } else if (q != 0.0 && (alp_p(e, zlo) != alp_p(e, zhi) || pj != alp_p(e, zlo))) bad = 1;   // f32 denormals: both ends of the interval
            }
            pw[j] = pj;
        }
    }
    ALP_STAMP(1);
    if (bad) *flag = 1;
    __syncthreads();
    ALP_STAMP(2);
    if (*flag) {
        // the reference's serial sum, j ascending, f64 (operations_impl.go:492-499): one wave, 16 values in flight ahead of the adds
        if (wave == 0) {
            double z = 0.0;
            for (int j0 = 0; j0 < T; j0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = E[j0 + u < T ? j0 + u : T - 1];
#pragma unroll
                for (int u = 0; u < 16; u++) z += (j0 + u < T) ? v[u] : 0.0;
            }
            if (lane == 0) { zsh[0] = z; if (p.zseq_count && ds == 0) atomicAdd(p.zseq_count, 1); }
        }
        __syncthreads();
        const double z = zsh[0];
        for (int j0 = 0; j0 < T; j0 += ALP_EU * ALP_NT) {
#pragma unroll
            for (int u = 0; u < ALP_EU; u++) { const int j = j0 + u * ALP_NT + tid; ev[u] = E[j < T ? j : T - 1]; }
#pragma unroll
            for (int u = 0; u < ALP_EU; u++) { const int j = j0 + u * ALP_NT + tid; if (j < T) pw[j] = alp_p(ev[u], z); }
        }
        __syncthreads();
    }
    // ---- PV: out[d] = trunc(sum_{j ascending} p_j * v[j][d]) for d = ds*16 .. +15 (llamatransformer.go:504-514), the row-broadcast chain
    // of rowcast_kernel: an adder wave owns 4 output dims, one per DPP row of 16 lanes; lane (q, jj) receives the exact products of dim q at
    // the positions j = jj (mod 16) -- eight per 128-position chunk, two 16 B LDS reads -- and every lane of the row adds the 16 lanes'
    // products in position order with v_add_f32_dpp row_newbcast (chain128): 6 cycles per position instead of 8.7 for an LDS-fed chain.
    // Iteration it: the producers turn batch it (rows in registers, loaded three batches ahead) into products in ring[it & 1]
    // ([dim][chunk][half][position % 16][4]: both the scattered 4 B writes and the 16 B reads are conflict-free); the adders walk batch
    // it - 1.  Positions past T carry p == +0 (products +-0, acc is never -0).
    float acc = 0.0f;
    if (wave < 4) {
        const int d = 4 * wave + (lane >> 4), jj = lane & 15;
        ALP_STAMP(3);
        for (int it = 0; it <= nbatch; it++) {
            if (it == 2) ALP_STAMP(4);
            if (it > 0) {
                const float* src = ring + (size_t)((it - 1) & 1) * ALP_SLOT + (size_t)d * ALP_BATCH + jj * 4;
                float4 a0 = *(const float4*)(src), a1 = *(const float4*)(src + 64);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    if (c < 3) { a0 = *(const float4*)(src + (c + 1) * 128); a1 = *(const float4*)(src + (c + 1) * 128 + 64); }   // in flight behind the 128 adds
                    asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
                    __builtin_amdgcn_sched_barrier(0);
                    chain128(acc, pr);
                }
            }
            __syncthreads();
        }
        ALP_STAMP(5);
        if (jj == 0) p.out[(size_t)h * HD + ds * ALP_DS + d] = bf_trunc(acc);
        ALP_STAMP(6);
    } else {
        auto produce = [&](const uint4 (&v)[4], int b) {
            float* dst = ring + (size_t)(b & 1) * ALP_SLOT + (size_t)(half8 * 8) * ALP_BATCH + ((prow >> 6) & 1) * 64 + (prow & 15) * 4 + ((prow >> 4) & 3);
#pragma unroll
            for (int r = 0; r < 4; r++) {                    // round r = chunk r of the batch; position prow of it: e = prow >> 4, jj = prow & 15
                const float pj = pw[b * ALP_BATCH + r * 128 + prow];
                const uint4 w = v[r];                        // exact products: 8-bit x 8-bit significands
                float* q = dst + r * 128;
                q[0 * ALP_BATCH] = pj * bf_lo(w.x); q[1 * ALP_BATCH] = pj * bf_hi(w.x); q[2 * ALP_BATCH] = pj * bf_lo(w.y); q[3 * ALP_BATCH] = pj * bf_hi(w.y);
                q[4 * ALP_BATCH] = pj * bf_lo(w.z); q[5 * ALP_BATCH] = pj * bf_hi(w.z); q[6 * ALP_BATCH] = pj * bf_lo(w.w); q[7 * ALP_BATCH] = pj * bf_hi(w.w);
            }
        };
        auto step = [&](int it, const uint4 (&cur)[4], uint4 (&nxt)[4]) {
            if (it <= nbatch) {                              // (uniform; one barrier per iteration, like the adders)
                if (it < nbatch) { load(nxt, it + 3); produce(cur, it); }
                __syncthreads();
            }
        };
        ALP_STAMP(3);
        for (int it = 0; it <= nbatch; it += 4) { step(it, v0, v3); if (it == 0) ALP_STAMP(4); step(it + 1, v1, v0); step(it + 2, v2, v1); step(it + 3, v3, v2); }
        ALP_STAMP(5); ALP_STAMP(6);
    }
}

template <int HD> __global__ __launch_bounds__(ALP_NT) void attn_long_pv_kernel(AttnParams p) { alp_eager_body<HD>(p); }

// ------------------------------------------------------------------------------------------------
// attn_long_pv2_kernel (round 6): attn_long_pv_kernel with the p_j evaluated LAZILY, off the critical path.
//
// Stamps of attn_long_pv_kernel at T = 4101 (profiles/r06_att_timing.log): 10.1 k cycles pass before the first product is made -- every
// thread fetches eight e_j, forms eight certified p_j -- then 25.8 k of PV.  Only the first batch's p_j are needed to start the chain: here a
// PRODUCER thread evaluates the p_j of a position right where it multiplies that position's V row (its e_j travels with the V rows, three batches
// ahead), so what stands in front of the chain is the Z estimate and one batch.  The certification is the same (cert_p); an element that
// cannot be certified is found while the chain already runs, so the kernel finishes the optimistic pass, and if ANY element failed (about
// once in 1e9) or force_zseq is set it walks the reference's serial sum and runs the whole PV again with the exact Z: a certified pass
// equals the reference's bits whatever the serial sum is (monotone step function, attn_long_pv_kernel's header); an uncertified one is discarded.
// Same grid, block and LDS as attn_long_pv_kernel (the p array is unused).  The kernel symbol attn_long_pv2_kernel picks between the two bodies (below).
// ------------------------------------------------------------------------------------------------
template <int HD> DEVINL void alp_lazy_body(const AttnParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, ds; xcd_head_block(h, ds);
    const int T = p.st->pos + 1, nblk = (T + ALS_NT - 1) / ALS_NT, nbatch = (T + ALP_BATCH - 1) / ALP_BATCH;
    float* ring = (float*)(smem + (size_t)((p.seq_len + ALP_BATCH - 1) / ALP_BATCH + 1) * ALP_BATCH * 4);   // [2][ALP_SLOT] products
    double* zsh = (double*)(ring + 2 * ALP_SLOT);
    int* const flag = (int*)(zsh + 1);
    const double* E = p.e_buf + (size_t)h * p.seq_len;
    const int kvh = h / (p.H / p.KVH);
#define ALQ_STAMP(n) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0) p.dbg[(wave >> 2) * 16 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    ALQ_STAMP(0);
    // address-only loads first, oldest-needed first: the per-block partial sums, then the producers' first three batches of e_j and V rows
    const int nblk_max = (p.seq_len + ALS_NT - 1) / ALS_NT;
    const double* zp = p.z_part + (size_t)h * nblk_max;
    double zmine = zp[lane < nblk_max ? lane : nblk_max - 1];
    const uint16_t* vbase = p.cache_v + (size_t)kvh * HD + (size_t)ds * ALP_DS;
    const size_t vrow = (size_t)p.KVH * HD;
    const int pl = tid & 255, half8 = pl & 1, prow = pl >> 1;                // producers: 128 positions x two 8-dim halves per round, 4 rounds per batch
    struct Rows { uint4 v[4]; double e[4]; };
    auto load = [&](Rows& r_, int b) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int j = b * ALP_BATCH + r * 128 + prow; j = j < p.seq_len ? j : p.seq_len - 1;       // (clamped to the ARRAY end: no wait for the position word)
            r_.e[r] = E[j];
            r_.v[r] = *(const uint4*)(vbase + (size_t)j * vrow + half8 * 8);
        }
    };
    Rows v0, v1, v2, v3;
    if (wave >= 4) { load(v0, 0); load(v1, 1); load(v2, 2); }
    if (tid == 0) *flag = 0;
    // ---- Z estimate: the per-block tree sums in block order (same value in every thread)
    double zt = 0.0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        if (b0) zmine = zp[b0 + lane < nblk ? b0 + lane : nblk - 1];
        const int nb = nblk - b0 < 64 ? nblk - b0 : 64;
        const long long zbits = __double_as_longlong(zmine);
        for (int b = 0; b < nb; b++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)zbits, b), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(zbits >> 32), b);
            zt += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
    }
    const CertZ cz = cert_z(zt, T);
    __syncthreads();
    ALQ_STAMP(1);
    float acc = 0.0f;
    int bad = 0;
    // one pass of PV over all batches; exact_z == 0: p_j = cert_p(e_j) (sets bad), else p_j = trunc(f32(e_j / z)) with the walked serial sum z
    auto run_pv = [&](bool exact, double z) {
        acc = 0.0f;
        if (wave < 4) {
            const int d = 4 * wave + (lane >> 4), jj = lane & 15;
            for (int it = 0; it <= nbatch; it++) {
                if (it > 0) {
                    const float* src = ring + (size_t)((it - 1) & 1) * ALP_SLOT + (size_t)d * ALP_BATCH + jj * 4;
                    float4 a0 = *(const float4*)(src), a1 = *(const float4*)(src + 64);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        float pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        if (c < 3) { a0 = *(const float4*)(src + (c + 1) * 128); a1 = *(const float4*)(src + (c + 1) * 128 + 64); }   // in flight behind the 128 adds
                        asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
                        __builtin_amdgcn_sched_barrier(0);
                        chain128(acc, pr);
                    }
                }
                __syncthreads();
            }
        } else {
            auto produce = [&](const Rows& r_, int b) {
                float* dst = ring + (size_t)(b & 1) * ALP_SLOT + (size_t)(half8 * 8) * ALP_BATCH + ((prow >> 6) & 1) * 64 + (prow & 15) * 4 + ((prow >> 4) & 3);
#pragma unroll
                for (int r = 0; r < 4; r++) {                    // round r = chunk r of the batch; position prow of it
                    const int j = b * ALP_BATCH + r * 128 + prow;
                    float pj = 0.0f;                             // +0 past the row: products +-0, acc is never -0
                    if (j < T) pj = exact ? alp_p(r_.e[r], z) : cert_p(r_.e[r], cz, bad);       // impl:506 + ToBFloat16 :493
                    const uint4 w = r_.v[r];                     // exact products: 8-bit x 8-bit significands
                    float* q = dst + r * 128;
                    q[0 * ALP_BATCH] = pj * bf_lo(w.x); q[1 * ALP_BATCH] = pj * bf_hi(w.x); q[2 * ALP_BATCH] = pj * bf_lo(w.y); q[3 * ALP_BATCH] = pj * bf_hi(w.y);
                    q[4 * ALP_BATCH] = pj * bf_lo(w.z); q[5 * ALP_BATCH] = pj * bf_hi(w.z); q[6 * ALP_BATCH] = pj * bf_lo(w.w); q[7 * ALP_BATCH] = pj * bf_hi(w.w);
                }
            };
            auto step = [&](int it, const Rows& cur, Rows& nxt) {
                if (it <= nbatch) {                              // (uniform; one barrier per iteration, like the adders)
                    if (it < nbatch) { load(nxt, it + 3); produce(cur, it); }
                    __syncthreads();
                }
            };
            for (int it = 0; it <= nbatch; it += 4) { step(it, v0, v3); step(it + 1, v1, v0); step(it + 2, v2, v1); step(it + 3, v3, v2); }
        }
    };
    if (!p.force_zseq) {
        run_pv(false, 0.0);
        ALQ_STAMP(2);
        if (bad) *flag = 1;
        __syncthreads();
    }
    if (p.force_zseq || *flag) {
        // the reference's serial sum, j ascending, f64 (operations_impl.go:492-499): one wave, 16 values in flight ahead of the adds
        if (wave == 0) {
            double z = 0.0;
            for (int j0 = 0; j0 < T; j0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = E[j0 + u < T ? j0 + u : T - 1];
#pragma unroll
                for (int u = 0; u < 16; u++) z += (j0 + u < T) ? v[u] : 0.0;
            }
            if (lane == 0) { zsh[0] = z; if (p.zseq_count && ds == 0) atomicAdd(p.zseq_count, 1); }
        }
        if (wave >= 4) { load(v0, 0); load(v1, 1); load(v2, 2); }       // the producers' pipeline starts over
        __syncthreads();
        run_pv(true, zsh[0]);
    }
    ALQ_STAMP(5);
    if (wave < 4 && (lane & 15) == 0) p.out[(size_t)h * HD + ds * ALP_DS + 4 * wave + (lane >> 4)] = bf_trunc(acc);
    ALQ_STAMP(6);
#undef ALQ_STAMP
}

// up to two batches (T <= 1024) the eager form is the faster one (all 512 threads share the first batch's p_j: 11.2 against 11.5 us at T = 768,
// 9.6 against 10.1 at 272); from three batches on the lazy one (T = 2048: 15.1 -> 14.5 us, 4101: 23.0 -> 21.8).  T lives on the device (one captured
// graph serves every position), so the choice is made here, per launch, uniformly over the grid.
template <int HD> __global__ __launch_bounds__(ALP_NT) void attn_long_pv2_kernel(AttnParams p) {
    if (p.st->pos + 1 <= 2 * ALP_BATCH) alp_eager_body<HD>(p); else alp_lazy_body<HD>(p);
}

// ------------------------------------------------------------------------------------------------
// attn_one_kernel (round 6): the long-context decode attention in ONE launch.
//
// The two-launch form above pays, per layer at T = 4100: the scores launch (6.3 us), a kernel boundary, and ~8 k cycles at the head of
// the PV launch in which every workgroup waits for e_buf / z_part to come back through the fabric (stamps: profiles/r06_att_timing.log) --
// 23.1 us against a PV chain of 10.4 us.  Here the (head, 16-dim slice) workgroups of attn_long_pv_kernel compute the scores themselves:
//   T <= 512  every slice workgroup of a head scores all positions (one per thread; the K rows are L2 hits for seven of the eight) and keeps
//             the exponentials in its LDS: no exchange at all;
//   T >  512  position block b (512 positions) belongs to slice b % slices: the workgroup scores its blocks, publishes e_j (f64) and the block's
//             tree sum with WRITE-THROUGH stores (global_store ... sc1), drains them (s_waitcnt vmcnt(0)), arrives on the head's counter
//             (one device-scope atomic; the generation of the launch is old / slices, so the counter never has to be reset and a replayed
//             graph needs no per-launch argument), polls it with sc1 loads, and then reads the head's row with sc1 loads (L2 / fabric, never a
//             stale L1 line) -- the flag form MI355X_MICROARCH.md lists as valid for any placement of the workgroups (no agent fence:
//             round 3's single-launch attempt paid an L2 write-back + invalidate per fence, 34.5 us against 26.8).
// A workgroup NEVER depends on another one being resident: the poll is bounded (ATT1_TIMEOUT ticks of the 100 MHz wall clock); when it
// runs out -- another context's launch holds the CUs its peers need -- the workgroup scores the missing blocks itself (the values it stores
// are the ones the owner would store: same instructions, same bits) and goes on.  Two such launches of two contexts can therefore never
// wait for each other; the price of contention is time, not a hang (longctx == 3 forces that path for the tests).
// Everything after the scores is attn_long_pv_kernel: tree estimate of Z + certified p_j (or the serial walk), the row-broadcast PV chain.
// grid (H, hd / 16) through xcd_head_block (a head's slices and its GQA group share an XCD), block 512, dynamic LDS att1_lds_bytes().
// ------------------------------------------------------------------------------------------------
constexpr long long ATT1_TIMEOUT = 4000;                     // 40 us of wall clock: ~2x the whole kernel at 4 K positions
__host__ __device__ inline size_t att1_lds_bytes(int seq_len, int hd) { return alp_lds_bytes(seq_len) + 64 + (size_t)hd * 4 + 64 + (size_t)ALP_BATCH * 8; }
DEVINL void st_sc1_f64(double* p, double v) { __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL double ld_sc1_f64(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
template <int HD> __global__ __launch_bounds__(ALP_NT) void attn_one_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = HD / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int h, ds; xcd_head_block(h, ds);
    const int NSL = (int)gridDim.y;
    const int T = p.st->pos + 1, nbatch = (T + ALP_BATCH - 1) / ALP_BATCH;      // 512-position blocks: the unit of the scores AND of the PV batches
    const int Tpad = (nbatch + 1) * ALP_BATCH;
    float* pw = (float*)smem;
    float* ring = (float*)(smem + (size_t)((p.seq_len + ALP_BATCH - 1) / ALP_BATCH + 1) * ALP_BATCH * 4);
    double* zsh = (double*)(ring + 2 * ALP_SLOT);            // [0] serial Z | [1] flags (two ints) | [2..3] -
    int* const flag = (int*)(zsh + 1);
    double* wsum = zsh + 8;                                  // [8] wave partials of a block's tree sum
    float* qf = (float*)(wsum + 8);                          // [HD]
    double* e_s = (double*)(qf + HD);                        // [512] (T <= 512: the row's exponentials never leave the workgroup)
    double* E = p.e_buf + (size_t)h * p.seq_len;
    double* zp = p.z_part + (size_t)h * (((size_t)p.seq_len + 63) / 64 + 8);      // one partial per 64 positions (a wave of a block)
    const int kvh = h / (p.H / p.KVH);
    const bool local = nbatch <= 1;                          // (uniform over the grid: T comes from the device-side position)
#define AT1_STAMP(n) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0) p.dbg[(wave >> 2) * 16 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    AT1_STAMP(0);
    // ---- q and this workgroup's first TWO blocks of K rows go in flight first (a block = 512 positions, one per thread; block b belongs to slice
    // b % NSL; at T = 4101 slice 0 owns two: scored one after the other with the second one's rows fetched behind the first one's chain they cost
    // 11.7 k cycles, prefetched 6 k), then the producers' first three batches of V rows
    const uint4* kbase = (const uint4*)p.cache_k + (size_t)kvh * NK * p.seq_len;
    const uint16_t q16 = p.q[(size_t)h * HD + (tid < HD ? tid : 0)];
    const int b_first = local ? 0 : ds;
    uint4 ka[NK], kb[NK];
    auto load_k = [&](uint4 (&k)[NK], int b) { const int j = b * ALP_BATCH + tid; attn_load_k<NK>(k, kbase, p.seq_len, j < T ? j : T - 1); };
    load_k(ka, b_first);
    if (!local && b_first + NSL < nbatch) load_k(kb, b_first + NSL);
    const uint16_t* vbase = p.cache_v + (size_t)kvh * HD + (size_t)ds * ALP_DS;
    const size_t vrow = (size_t)p.KVH * HD;
    const int pl = tid & 255, half8 = pl & 1, prow = pl >> 1;
    auto load = [&](uint4 (&v)[4], int b) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int j = b * ALP_BATCH + r * 128 + prow; j = j < T ? j : T - 1;
            v[r] = *(const uint4*)(vbase + (size_t)j * vrow + half8 * 8);
        }
    };
    uint4 v0[4], v1[4], v2[4], v3[4];
    if (wave >= 4) { load(v0, 0); load(v1, 1); load(v2, 2); }
    if (tid < HD) qf[tid] = bf_wide(q16);
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    AT1_STAMP(1);
    // ---- scores of one 512-position block (llamatransformer.go:456-473, Softmax impl:498) + every WAVE's tree sum of its 64 exponentials (only ever
    // an ESTIMATE of Z): no workgroup barrier in here
    auto score_block = [&](const uint4 (&k)[NK], int b) {
        const int j = b * ALP_BATCH + tid;
        double ev = 0.0;
        if (j < T) ev = attn_score_value<NK>(k, qf, p.divisor);
        if (local) e_s[tid] = ev; else if (j < T) st_sc1_f64(E + j, ev);
        double sm = ev;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        if (lane == 0) { if (local) wsum[wave] = sm; else st_sc1_f64(zp + b * 8 + wave, sm); }
    };
    auto score_mine = [&](int ofs) {                         // blocks ofs, ofs + NSL, ... of this workgroup, the next one's K rows always in flight
        for (int b = ofs; b < nbatch; b += 2 * NSL) {
            if (b != ofs) { if (b + NSL < nbatch) load_k(kb, b + NSL); }
            score_block(ka, b);
            if (b + NSL < nbatch) {
                if (b + 2 * NSL < nbatch) load_k(ka, b + 2 * NSL);
                score_block(kb, b + NSL);
            }
        }
    };
    if (b_first < nbatch) score_mine(b_first);
    if (!local) {
        AT1_STAMP(2);
        // ---- publish: the write-through stores have left this CU (vmcnt counts stores on gfx9), one lane arrives and polls
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* cnt = (unsigned*)p.cnt + h;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (old / (unsigned)NSL + 1u) * (unsigned)NSL;
            int ok = 0;
            if (p.longctx != 3) {
                const long long t0 = (long long)wall_clock64();
                for (;;) {
                    if ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
                    if ((long long)wall_clock64() - t0 > ATT1_TIMEOUT) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) atomicAdd((unsigned*)p.cnt + p.H, 1u);      // (diagnostics: polls that ran out)
            }
            flag[1] = ok;
        }
        __syncthreads();
        if (!flag[1]) {                                      // peers not resident (or the forced test path): score their blocks here -- same bits
            for (int o = 1; o < NSL; o++) {
                const int ofs = (ds + o) % NSL;
                if (ofs < nbatch) { load_k(ka, ofs); if (ofs + NSL < nbatch) load_k(kb, ofs + NSL); score_mine(ofs); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else __syncthreads();                                  // (e_s and wsum are complete)
    AT1_STAMP(3);
    // ---- Z estimate: the waves' tree sums (64 positions each), 64 of them per round, reduced by a fixed-shape butterfly: every thread of every
    // workgroup of the head gets the same value (any order of adding T non-negative doubles is inside the certified band)
    double zt = 0.0;
    if (local) zt = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]));
    else {
        const int nw = (T + 63) >> 6;                        // waves that stored a partial
        for (int w0 = 0; w0 < nw; w0 += 64) {
            double z = w0 + lane < nw ? ld_sc1_f64(zp + w0 + lane) : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
            zt += z;
        }
    }
    // ---- p_j with the estimate, certified (attn_long_pv_kernel's header); e_j from the LDS (local) or with sc1 loads
    const unsigned delta32 = 4u * (unsigned)T + 12u;
    const double epsr = (double)(4 * T + 8) * 1.1102230246251565e-16;
    const double zlo = zt * (1.0 - epsr), zhi = zt * (1.0 + epsr), rzt = 1.0 / zt;
    int bad = p.force_zseq;
    double ev[ALP_EU];
    for (int j0 = 0; j0 < Tpad; j0 += ALP_EU * ALP_NT) {
#pragma unroll
        for (int u = 0; u < ALP_EU; u++) {
            const int j = j0 + u * ALP_NT + tid;
            ev[u] = local ? (j < ALP_BATCH ? e_s[j] : 0.0) : ld_sc1_f64(E + (j < T ? j : T - 1));
        }
#pragma unroll
        for (int u = 0; u < ALP_EU; u++) {
            const int j = j0 + u * ALP_NT + tid;
            if (j >= Tpad) continue;
            float pj = 0.0f;
            if (j < T) {
                const double e = ev[u];
                const double q = e * rzt;
                pj = bf_wide(bf_trunc((float)q));
                const unsigned qh = (unsigned)__double2hiint(q), ql = (unsigned)__double2loint(q);
                const unsigned eq = (qh >> 20) & 0x7FFu;
                if (eq - 897u <= 126u) {
                    if ((qh & 0x1FFFu) == 0x1FFFu && ql - (0xF0000000u - delta32) <= 2u * delta32) bad = 1;
                } else if (q != 0.0 && (alp_p(e, zlo) != alp_p(e, zhi) || pj != alp_p(e, zlo))) bad = 1;
            }
            pw[j] = pj;
        }
        if (local) break;                                    // (Tpad = 1024: one round covers it)
    }
    AT1_STAMP(4);
    if (bad) flag[0] = 1;
    __syncthreads();
    if (flag[0]) {
        // the reference's serial sum, j ascending, f64 (operations_impl.go:492-499)
        if (wave == 0) {
            double z = 0.0;
            for (int j0 = 0; j0 < T; j0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) { const int j = j0 + u < T ? j0 + u : T - 1; v[u] = local ? e_s[j] : ld_sc1_f64(E + j); }
#pragma unroll
                for (int u = 0; u < 16; u++) z += (j0 + u < T) ? v[u] : 0.0;
            }
            if (lane == 0) { zsh[0] = z; if (p.zseq_count && ds == 0) atomicAdd(p.zseq_count, 1); }
        }
        __syncthreads();
        const double z = zsh[0];
        for (int j = tid; j < T; j += ALP_NT) pw[j] = alp_p(local ? e_s[j] : ld_sc1_f64(E + j), z);
        __syncthreads();
    }
    // ---- PV: attn_long_pv_kernel's row-broadcast chain (llamatransformer.go:504-514)
    float acc = 0.0f;
    if (wave < 4) {
        const int d = 4 * wave + (lane >> 4), jj = lane & 15;
        AT1_STAMP(5);
        for (int it = 0; it <= nbatch; it++) {
            if (it > 0) {
                const float* src = ring + (size_t)((it - 1) & 1) * ALP_SLOT + (size_t)d * ALP_BATCH + jj * 4;
                float4 a0 = *(const float4*)(src), a1 = *(const float4*)(src + 64);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    if (c < 3) { a0 = *(const float4*)(src + (c + 1) * 128); a1 = *(const float4*)(src + (c + 1) * 128 + 64); }
                    asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
                    __builtin_amdgcn_sched_barrier(0);
                    chain128(acc, pr);
                }
            }
            __syncthreads();
        }
        AT1_STAMP(6);
        if (jj == 0) p.out[(size_t)h * HD + ds * ALP_DS + d] = bf_trunc(acc);
    } else {
        auto produce = [&](const uint4 (&v)[4], int b) {
            float* dst = ring + (size_t)(b & 1) * ALP_SLOT + (size_t)(half8 * 8) * ALP_BATCH + ((prow >> 6) & 1) * 64 + (prow & 15) * 4 + ((prow >> 4) & 3);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pj = pw[b * ALP_BATCH + r * 128 + prow];
                const uint4 w = v[r];
                float* q = dst + r * 128;
                q[0 * ALP_BATCH] = pj * bf_lo(w.x); q[1 * ALP_BATCH] = pj * bf_hi(w.x); q[2 * ALP_BATCH] = pj * bf_lo(w.y); q[3 * ALP_BATCH] = pj * bf_hi(w.y);
                q[4 * ALP_BATCH] = pj * bf_lo(w.z); q[5 * ALP_BATCH] = pj * bf_hi(w.z); q[6 * ALP_BATCH] = pj * bf_lo(w.w); q[7 * ALP_BATCH] = pj * bf_hi(w.w);
            }
        };
        auto step = [&](int it, const uint4 (&cur)[4], uint4 (&nxt)[4]) {
            if (it <= nbatch) {
                if (it < nbatch) { load(nxt, it + 3); produce(cur, it); }
                __syncthreads();
            }
        };
        AT1_STAMP(5);
        for (int it = 0; it <= nbatch; it += 4) { step(it, v0, v3); step(it + 1, v1, v0); step(it + 2, v2, v1); step(it + 3, v3, v2); }
        AT1_STAMP(6);
    }
#undef AT1_STAMP
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// Fwd_Get_Rows (operations_impl.go:142-173): byte copy of embedding rows
__global__ void embed_kernel(const uint16_t* emb, const int32_t* tokens, uint16_t* x, int dim, int vocab, int* err) {
    const int m = blockIdx.x;
    const int t = tokens[m];
    if (t < 0 || t >= vocab) { if (threadIdx.x == 0) atomicExch(err, 1 + m); return; }
    const uint4* src = (const uint4*)(emb + (size_t)t * dim);
    uint4* dst = (uint4*)(x + (size_t)m * dim);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) dst[i] = src[i];
}

// ml.Argmax (operations_impl.go:513-548): strict '<' scan from -MaxFloat32 => first maximum wins, NaN and
// -inf are never selected (index -1 if nothing qualifies).  Parallel form: max value, lowest index.
// Also advances the device-resident greedy loop (inference.go:211-226): next token -> tokens[0], pos += 1.
DEVINL int argmax_block(const uint16_t* logits, int V, float* sv, int* si) {      // result valid in thread 0
    const int tid = threadIdx.x;
    float best = -3.40282346638528859811704183484516925440e+38f; int bi = -1;
    // 16 B loads, four in flight per thread (the scalar one-load-per-iteration form was pure latency: 42 us for V=128256);
    // each thread still visits its own elements in ascending index order, so '<' keeps the first maximum
    const int nv = ((((uintptr_t)logits) & 15) == 0) ? V >> 3 : 0;
    const uint4* l4 = (const uint4*)logits;
    for (int c0 = tid; c0 < nv; c0 += 4096) {
        uint4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = c0 + u * 1024; w[u] = l4[c < nv ? c : c0]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c0 + u * 1024;
            if (c < nv) {
                const uint32_t q[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float a = bf_lo(q[t]), b = bf_hi(q[t]);
                    if (best < a) { best = a; bi = c * 8 + 2 * t; }
                    if (best < b) { best = b; bi = c * 8 + 2 * t + 1; }
                }
            }
        }
    }
    for (int j = nv * 8 + tid; j < V; j += 1024) { float v = bf_wide(logits[j]); if (best < v) { best = v; bi = j; } }
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            float v2 = sv[tid + s]; int i2 = si[tid + s];
            float v1 = sv[tid]; int i1 = si[tid];
            bool take = (i2 >= 0) && (i1 < 0 || v1 < v2 || (v1 == v2 && i2 < i1));
            if (take) { sv[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    return si[0];
}
__global__ __launch_bounds__(1024) void argmax_kernel(const uint16_t* logits, int V, int32_t* next_token, StepState* st,
                                                      int32_t* out_tokens, int out_cap, int advance) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int tok = argmax_block(logits, V, sv, si);
    if (threadIdx.x == 0) {
        if (!advance) next_token[0] = tok;
        else if (!st->finished) {                            // (a finished generation: nothing moves any more, see StepState)
            next_token[0] = tok;
            int n = st->n_out;
            if (n < out_cap) out_tokens[n] = tok;
            st->n_out = n + 1;
            st->pos = st->pos + 1;
            if (st->honour_stop) for (int k = 0; k < st->n_stop; k++) if (tok == st->stop[k]) st->finished = 1;     // the stop token itself is emitted (GSFinishedByReachingEOS)
        }
    }
}
// the same for the batch: block s = sequence s's logits row; its context's token word, token log and position advance (inference.go:211-226)
__global__ __launch_bounds__(1024) void batch_argmax_kernel(const uint16_t* logits, int V, const BatchTab* tab, int32_t* ring) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int s = blockIdx.x;
    const int tok = argmax_block(logits + (size_t)s * V, V, sv, si);
    if (threadIdx.x == 0) {
        StepState* st = tab->st[s];
        if (!st->finished) {                                 // a finished sequence keeps its token word, log and position (its column goes on computing the same step)
            *tab->dtok[s] = tok;
            if (ring) ring[s] = tok;                         // (pipeline: the contiguous words the last stage sends to the first)
            const int n = st->n_out;
            if (n < tab->dout_cap[s]) tab->dout[s][n] = tok;
            st->n_out = n + 1;
            st->pos = st->pos + 1;
            if (st->honour_stop) for (int k = 0; k < st->n_stop; k++) if (tok == st->stop[k]) st->finished = 1;
        }
    }
}

__global__ void set_state_kernel(StepState* st, int pos, int n_out, int honour_stop) { st->pos = pos; st->n_out = n_out; st->finished = 0; st->honour_stop = honour_stop; }
struct StopIds { int32_t n; int32_t id[LNB_MAX_STOP_IDS]; };
__global__ void set_stop_kernel(StepState* st, StopIds s) { st->n_stop = s.n; for (int k = 0; k < LNB_MAX_STOP_IDS; k++) st->stop[k] = k < s.n ? s.id[k] : -1; st->finished = 0; }
__global__ void advance_state_kernel(StepState* st, int rows) { st->pos = st->pos + rows; }   // end of a captured pipeline-stage step

// ---- weight re-tiling (load time, once) ------------------------------------------------------------
__global__ void tile_scatter_kernel(const uint16_t* src, uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 8-element chunk
    size_t total = (size_t)rows * (K >> 3);
    if (idx >= total) return;
    int r = (int)(idx / (K >> 3)), kc = (int)(idx % (K >> 3));
    uint4 v = *(const uint4*)(src + (size_t)r * K + (size_t)kc * 8);
    if (RW == 4) {                                           // row-broadcast layout: the 8 k's are not adjacent
        const uint16_t* e = (const uint16_t*)&v;
        for (int i = 0; i < 8; i++) dst[tiled_index(row_off + r, kc * 8 + i, chain, K, RW, NCH)] = e[i];
        return;
    }
    *(uint4*)(dst + tiled_index(row_off + r, kc * 8, chain, K, RW, NCH)) = v;
}
__global__ void tile_gather_kernel(const uint16_t* src, uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)rows * (K >> 3);
    if (idx >= total) return;
    int r = (int)(idx / (K >> 3)), kc = (int)(idx % (K >> 3));
    if (RW == 4) {
        for (int i = 0; i < 8; i++) dst[(size_t)r * K + (size_t)kc * 8 + i] = src[tiled_index(row_off + r, kc * 8 + i, chain, K, RW, NCH)];
        return;
    }
    uint4 v = *(const uint4*)(src + tiled_index(row_off + r, kc * 8, chain, K, RW, NCH));
    *(uint4*)(dst + (size_t)r * K + (size_t)kc * 8) = v;
}
// synthetic weights straight into their final layout (RW == 0: linear row-major)
__global__ void synth_fill_kernel(uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH,
                                  uint64_t base, int kind, float sigma) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)rows * K;
    if (idx >= total) return;
    int r = (int)(idx / K), k = (int)(idx % K);
    float g = (float)lnb_synth_isum(base, idx) * (1.0f / 53510.0f);
    float v = kind == 1 ? fmaf(0.1f, g, 1.0f) : sigma * g;
    size_t o = RW ? tiled_index(row_off + r, k, chain, K, RW, NCH) : idx;
    dst[o] = bf_trunc(v);
}

#include "lnb_batch_kernels.h"

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from lnb_api.cpp)
// ------------------------------------------------------------------------------------------------
// x staging: a whole number of stages (steps per stage = stage_bytes / (nch*rw*2)) + 64 floats of slack
static size_t xs_bytes(int K, int steps_per_stage) { return ((size_t)((K + steps_per_stage - 1) / steps_per_stage) * steps_per_stage + 320) * 4 + 16; }

template <int RW, int NCH, int SA, int NH, int R, int EPI, bool NORM, int XC>
static hipError_t launch_chain_x(const GemvParams* p, hipStream_t st) {
    auto kfn = gemv_chain_kernel<RW, NCH, SA, NH, R, EPI, NORM, XC>;
    if (!p)   // prepare: raise the dynamic-LDS limit once, outside any stream capture
        return hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t lds = 4 * (size_t)SA + xs_bytes(p->K, SA / (NCH * RW * 2));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (xs_bytes(p->K, SA / (NCH * RW * 2)) / 4 > (size_t)XC * (1 + NH) * 512) return hipErrorInvalidValue;   // x staging registers
    if (NORM && (rms_scratch_bytes(rms_nf(NH)) > 4 * (size_t)SA || seq_leaf_size(p->K, rms_nf(NH) * 64) > 256)) return hipErrorInvalidValue;   // records live in the idle ring; leaf over-read stays inside the x padding
    hipLaunchKernelGGL(kfn, dim3((unsigned)(p->S * p->n_wg)), dim3((1 + NH) * 64), lds, st, *p);
    return hipGetLastError();
}
// x staging chunks per stager wave: the norm-fused kernels are instantiated for 2, 3 and 6 chunks of 512 elements per wave (K up to ~4 K, ~6 K / 12 K with
// the 70B-like shape's 8192 in the middle one, 24 K) and the smallest that holds the padded row is launched; the plain kernels keep their 12
template <int RW, int NCH, int SA, int NH, int R, int EPI, bool NORM>
static hipError_t launch_chain_t(const GemvParams* p, hipStream_t st) {
    if constexpr (NORM) {
        if (!p) {
            hipError_t e = launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, 2>(p, st);
            if (e == hipSuccess) e = launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, 3>(p, st);
            return e != hipSuccess ? e : launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, XCh<NORM>::value>(p, st);
        }
        const size_t need = xs_bytes(p->K, SA / (NCH * RW * 2)) / 4, per = (size_t)(1 + NH) * 512;
        if (need <= 2 * per) return launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, 2>(p, st);
        if (need <= 3 * per) return launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, 3>(p, st);
    }
    return launch_chain_x<RW, NCH, SA, NH, R, EPI, NORM, XCh<NORM>::value>(p, st);
}

template <int RW, int KS, int NH, int R, int EPI, bool NORM, int XC>
static hipError_t launch_quad_x(const GemvParams* p, hipStream_t st) {
    auto kfn = gemv_quad_kernel<RW, KS, NH, R, EPI, NORM, XC>;
    if (!p) return hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    constexpr int NW = gq_ncw(RW) + NH;
    constexpr size_t SB = (size_t)RW * KS * 4;
    if (p->K % KS) return hipErrorInvalidValue;                                  // whole stages only (auto_rw in lnb_api.cpp picks this layout accordingly)
    const size_t lds = GQ_SLOTS * SB + xs_bytes(p->K, KS);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (xs_bytes(p->K, KS) / 4 > (size_t)XC * NW * 512) return hipErrorInvalidValue;     // x staging registers
    if (NORM && (rms_scratch_bytes(rms_nf(NH)) > GQ_SLOTS * SB || seq_leaf_size(p->K, rms_nf(NH) * 64) > 256)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(p->S * p->n_wg)), dim3(NW * 64), lds, st, *p);
    return hipGetLastError();
}
template <int RW, int KS, int NH, int R, int EPI, bool NORM>
static hipError_t launch_quad_t(const GemvParams* p, hipStream_t st) {
    if constexpr (NORM) {
        if (!p) {
            hipError_t e = launch_quad_x<RW, KS, NH, R, EPI, NORM, 2>(p, st);
            return e != hipSuccess ? e : launch_quad_x<RW, KS, NH, R, EPI, NORM, XCh<NORM>::value>(p, st);
        }
        if (xs_bytes(p->K, KS) / 4 <= (size_t)2 * (gq_ncw(RW) + NH) * 512) return launch_quad_x<RW, KS, NH, R, EPI, NORM, 2>(p, st);
    }
    return launch_quad_x<RW, KS, NH, R, EPI, NORM, XCh<NORM>::value>(p, st);
}

static bool env_tp_w13() { static const int on = [] { const char* e = getenv("LNB_TP_W13"); return e && *e ? atoi(e) : 1; }(); return on != 0; }   // (A/B switch of the throughput gate|up form)
template <int EPI, bool NORM, int NCH>
static hipError_t launch_gemv_rw(const GemvParams* p, int rw, hipStream_t st) {
    // RW 24 (one chain; the 8B wq|wk|wv: 6144 rows = 256 blocks of 24, one per CU): quad-DPP chain waves fed through the LDS, 256-step
    // stages, six helpers x 2 loads x 5 stages = 60 KiB in flight per CU
    // Throughput schedule (GemvParams.sched, lnb_ctx_set_schedule): the same kernel with 128-step stages -- 3 x 12 KiB of products + x = 55 KB of LDS
    // instead of 91 KB, so that a workgroup fits a CU beside the gate|up workgroup (75 KB) of another context's step; twice the stage barriers
    // (slower alone, which is why one stream keeps the 256-step form), ten stages of one load per helper in flight = the same 60 KiB per CU.
    if constexpr (NCH == 1) if (rw == 24) return (p && p->sched) ? launch_quad_t<24, 128, 6, 10, EPI, NORM>(p, st) : launch_quad_t<24, 256, 6, LNB_QUAD_R, EPI, NORM>(p, st);
    // stage geometry (one workgroup per CU).  SA = bf16 bytes per stage, R = stages in flight per helper.
    //  RW 16/32 plain (thin; chain bound): 8 KiB stages, 2 helpers x 4 loads x 7 stages = 56 KiB in flight per CU.
    //  RW 32 with the fused RMSNorm (wq|wk|wv): six helpers so that the exact parallel norm sum has 384 folding lanes, and
    //  12 KiB = 192-step stages (a stage boundary costs one barrier + one exposed LDS round trip: 96-step stages ran the
    //  chain at 10.2 cycles per step, 192-step ones at 8.7); 6 x 2 loads x 5 stages = 60 KiB in flight per CU.
    //  RW 64 (fat; HBM bound): 12 KiB stages, six helpers paired on three SIMDs (the chain wave owns the fourth),
    //  6 x 2 loads x 8 stages = 96 KiB in flight per CU (5 stages: 5.8 TB/s on the LM head, 8: 6.2, 14: worse again).
    // very long rows (70B-like w2: K = 28672): x needs more staging registers than three stager waves hold -> four helpers
    if (p && !NORM && NCH == 1 && (rw == 16 || rw == 32) && xs_bytes(p->K, 8192 / (rw * 2)) / 4 > (size_t)XCh<false>::value * 3 * 512)
        return rw == 16 ? launch_chain_t<16, 1, 8192, 4, 7, EPI, false>(p, st) : launch_chain_t<32, 1, 8192, 4, 7, EPI, false>(p, st);
    // rung (a') of the FFN ladder (profiles/r06_ffn_stream.md; LNB_RW_W2=16 LNB_W2_QUAD=1, measurement only): the down projection's 16 rows per CU on ONE
    // quad_perm chain wave + four helpers -- the only w2 form that would leave a co-resident gate|up producer three SIMDs
    if constexpr (NCH == 1 && !NORM && (EPI == EPI_RESID || EPI == EPI_STORE)) {
        const char* eq = getenv("LNB_W2_QUAD");              // (read per launch: a test switches it inside one process)
        const int quad16 = (eq && *eq) ? atoi(eq) : 0;
        if (rw == 16 && quad16 && (!p || p->K % 128 == 0)) return quad16 == 2 ? launch_quad_t<16, 256, 4, 7, EPI, false>(p, st) : launch_quad_t<16, 128, 4, 14, EPI, false>(p, st);
    }
    if (rw == 16) return launch_chain_t<16, NCH, 8192, 2, 7, EPI, NORM>(p, st);
    if (rw == 32) return (NORM && NCH == 1) ? launch_chain_t<32, 1, 12288, 6, 5, EPI, NORM>(p, st) : launch_chain_t<32, NCH, 8192, 2, 7, EPI, NORM>(p, st);
    if (rw == 64) return launch_chain_t<64, NCH, 12288, 6, 8, EPI, NORM>(p, st);
    // RW 56 (two chains only): 28672 gate/up rows = 256 blocks of 56 x 2 -- one block per CU on ALL 256 CUs instead of 224 of them;
    // seven helpers (448 lanes = one (k-chunk, row) pair each), 64-step stages
    // (throughput schedule: six ring stages instead of eight -- 133 instead of 149 VGPRs, so that two of its waves and two of the 128-step
    // wq|wk|wv kernel's fit one SIMD's 512 registers: a gate|up workgroup of one context beside a wq|wk|wv workgroup of another)
    if constexpr (NCH == 2) if (rw == 56) return (p && p->sched && env_tp_w13()) ? launch_chain_t<56, 2, 14336, 7, 6, EPI, NORM>(p, st) : launch_chain_t<56, 2, 14336, 7, 8, EPI, NORM>(p, st);
    // RW 28 (two chains, LNB_RW_W13=28): the same stream cut into 512 half-height blocks, two per workgroup -- rows [0, F/2) are complete
    // after the first block of every workgroup: the row-band order a w1|w3 -> w2 pipeline needs (measurement, NOTES.md 6.1)
    if constexpr (NCH == 2) if (rw == 28) return launch_chain_t<28, 2, 14336, 7, 8, EPI, NORM>(p, st);
    return hipErrorInvalidValue;
}

template <int EPI> static hipError_t launch_rowcast(const GemvParams* p, hipStream_t st) {
    auto kfn = rowcast_kernel<EPI>;
    auto kl = rowcast_lds_kernel<EPI>;
    if (!p) {
        hipError_t e = hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e != hipSuccess ? e : hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if ((p->K & 127) || p->K > 16384) return hipErrorInvalidValue;              // x staging: 8 x 16 B per thread, K*4 bytes of LDS
    // helper-fed chain waves (rowcast_lds_kernel) whenever K is a whole number of its 512-step stages; LNB_ROWCAST_LDS=0: the self-feeding kernel
    static const int use_lds = [] { const char* s = getenv("LNB_ROWCAST_LDS"); return s && *s ? atoi(s) : 1; }();
    // (throughput schedule: the self-feeding kernel -- four waves and K * 4 bytes of LDS instead of eight waves and 96 KB + K * 2: co-resident with
    // another context's gate|up workgroup, NOTES R5)
    if (use_lds && !p->sched && p->K % (128 * RL_SC) == 0 && p->K >= 2 * 128 * RL_SC && rl_lds_bytes(p->K) + (size_t)(p->lds_pad > 0 ? p->lds_pad : 0) <= 160 * 1024) {
        hipLaunchKernelGGL(kl, dim3((unsigned)(p->S * p->n_wg)), dim3(512), rl_lds_bytes(p->K) + (size_t)(p->lds_pad > 0 ? p->lds_pad : 0), st, *p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)(p->S * p->n_wg)), dim3(256), (size_t)p->K * 4 + (size_t)(p->lds_pad > 0 ? p->lds_pad : 0), st, *p);
    return hipGetLastError();
}

extern "C" hipError_t lnbk_gemv(const GemvParams* p, int rw, int nch, int epi, int norm, hipStream_t st) {
    if (rw == 4) {                                           // row-broadcast layout (thin matrices)
        if (nch != 1 || norm) return hipErrorInvalidValue;
        if (epi == EPI_STORE) return launch_rowcast<EPI_STORE>(p, st);
        if (epi == EPI_RESID) return launch_rowcast<EPI_RESID>(p, st);
        return hipErrorInvalidValue;
    }
    if (nch == 2) {
        if (epi == EPI_SILU_MUL && norm) return launch_gemv_rw<EPI_SILU_MUL, true, 2>(p, rw, st);
        return hipErrorInvalidValue;
    }
    switch (epi) {
        case EPI_STORE: return norm ? launch_gemv_rw<EPI_STORE, true, 1>(p, rw, st) : launch_gemv_rw<EPI_STORE, false, 1>(p, rw, st);
        case EPI_QKV_ROPE: return norm ? launch_gemv_rw<EPI_QKV_ROPE, true, 1>(p, rw, st) : hipErrorInvalidValue;
        case EPI_RESID: return norm ? hipErrorInvalidValue : launch_gemv_rw<EPI_RESID, false, 1>(p, rw, st);
        default: return hipErrorInvalidValue;
    }
}

template <int EPI, int NCH> static hipError_t launch_gemm(const GemmParams* p, hipStream_t st) {
    auto k4 = gemm_mfma_kernel<EPI, NCH, 4, 128>;
    auto k2 = gemm_mfma_kernel<EPI, NCH, 4, 64>;
    auto k1 = gemm_mfma_kernel<EPI, NCH, 1, 128>;
    const size_t lds4 = gemm_lds_bytes(NCH, 4), lds2 = gemm_lds_bytes(NCH, 4, 64), lds1 = gemm_lds_bytes(NCH, 1);   // 53 / 73 KB (37 / 57, 39 / 44): >= two workgroups per CU
    if (!p) {
        hipError_t e = hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        return e != hipSuccess ? e : hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    }
    const unsigned mb = (unsigned)((p->S + 127) / 128), nb4 = (unsigned)((p->n_rows + GM_NB - 1) / GM_NB);
    static const int force = [] { const char* e = getenv("LNB_GEMM_TILE"); return e && *e ? atoi(e) : 0; }();       // 1 / 64 / 128: measurement aid
    // the two-chain w1|w3 kernel already runs 2x the MFMAs per staged byte: with more than four 64 x 128 tiles per CU it keeps them
    // (S = 2048: 3.90 ms against 4.11 ms with 64 x 64 tiles; the one-chain kernels gain 5 % from the smaller tile there;
    // S = 512, 896 tiles: the small tile still wins, 74 against 77 ms per Forward)
    const int tile = force ? force : (nb4 * mb <= 160 ? 1 : (NCH == 2 && nb4 * mb > 1024) ? 128 : nb4 * mb <= GM_MB64_UPTO ? 64 : 128);
    if (tile == 1)             // 64 x 128 tiles would leave most CUs idle: 16-row tiles, the four waves split the batch rows
                               // (tools/gemmbench.hip: 64 / 96 / 128 tiles: 1.4-2x faster; 256 tiles and up: 64-row tiles win)
        hipLaunchKernelGGL(k1, dim3((unsigned)((p->n_rows + 15) / 16), mb), dim3(256), lds1, st, *p);
    else if (tile == 64)       // about one 64 x 128 tile per CU: 64 x 64 tiles, two to three workgroups per CU overlap staging and MFMAs
        hipLaunchKernelGGL(k2, dim3(nb4, (unsigned)((p->S + 63) / 64)), dim3(256), lds2, st, *p);
    else
        hipLaunchKernelGGL(k4, dim3(nb4, mb), dim3(256), lds4, st, *p);
    return hipGetLastError();
}

extern "C" hipError_t lnbk_gemm(const GemmParams* p, int epi, hipStream_t st) {
    const int nch = p ? p->nch : 0;
    switch (epi) {
        case EPI_STORE: return launch_gemm<EPI_STORE, 1>(p, st);
        case EPI_RESID: return launch_gemm<EPI_RESID, 1>(p, st);
        case EPI_QKV_ROPE: return launch_gemm<EPI_QKV_ROPE, 1>(p, st);
        case EPI_SILU_MUL: return (p && nch != 2) ? hipErrorInvalidValue : launch_gemm<EPI_SILU_MUL, 2>(p, st);
        default: return hipErrorInvalidValue;
    }
}
extern "C" hipError_t lnbk_rmsnorm_rows(const uint16_t* x, const uint16_t* w, uint16_t* out, int S, int K, float eps, hipStream_t st) {
    static const int wide = getenv("LNB_NORM_ROWS_WIDE") ? atoi(getenv("LNB_NORM_ROWS_WIDE")) : 1;
    const size_t lds_w = bn_scratch() + ((size_t)bn_kpad(K) + 8) * 4;                  // one workgroup of seven waves per row: the exact parallel evaluation of the serial sum
    if (wide && !(K & 127) && lds_w <= 64 * 1024 && (size_t)bn_kpad(K) <= (size_t)XCh<true>::value * (1 + BN_NH) * 512 && seq_leaf_size(K, rms_nf(BN_NH) * 64) <= 256) {
        hipLaunchKernelGGL(batch_rmsnorm_xt_kernel<false>, dim3((unsigned)S), dim3((1 + BN_NH) * 64), lds_w, st, x, w, eps, out, K);
        return hipGetLastError();
    }
    if (((size_t)K + 64) * 4 > 150 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3((unsigned)S), dim3(64), ((size_t)K + 64) * 4, st, x, w, out, S, K, eps);
    return hipGetLastError();
}

static size_t attn_lds_bytes(int seq_len, int hd) { return attn_off_ring(seq_len, hd) + 2 * (size_t)ATT_RP * hd * 4; }
extern "C" size_t lnbk_attn_short_lds(int seq_len, int hd) { return attn_lds_bytes(seq_len, hd); }
// the longest context the one-workgroup-per-head kernel can stage in 160 KB of LDS (e f64 + p f32 per position + q + the product ring)
extern "C" int lnbk_attn_short_max_T(int hd) {
    int lo = 64, hi = 1 << 20;
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (attn_lds_bytes(mid, hd) <= 160 * 1024) lo = mid; else hi = mid - 1; }
    return lo;
}
static long long* g_gqa_dbg = nullptr;                       // LNB_ATTN_GQA_DBG: phase stamps of one attn_gqa_kernel workgroup
extern "C" hipError_t lnbk_init(void) {
    static bool done = false;
    if (done) return hipSuccess;
    { hipError_t eg; for (int ep = EPI_STORE; ep <= EPI_SILU_MUL; ep++) if ((eg = lnbk_gemm(nullptr, ep, nullptr)) != hipSuccess) return eg;
      if ((eg = hipFuncSetAttribute((const void*)rmsnorm_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return eg; }
    { hipError_t e4; if ((e4 = lnbk_gemv(nullptr, 4, 1, EPI_STORE, 0, nullptr)) != hipSuccess) return e4; if ((e4 = lnbk_gemv(nullptr, 4, 1, EPI_RESID, 0, nullptr)) != hipSuccess) return e4; }
    { hipError_t e56; if ((e56 = lnbk_gemv(nullptr, 56, 2, EPI_SILU_MUL, 1, nullptr)) != hipSuccess) return e56;
      if ((e56 = lnbk_gemv(nullptr, 28, 2, EPI_SILU_MUL, 1, nullptr)) != hipSuccess) return e56; }
    { hipError_t el;
      if ((el = launch_chain_t<16, 1, 8192, 4, 7, EPI_STORE, false>(nullptr, nullptr)) != hipSuccess) return el;
      if ((el = launch_chain_t<32, 1, 8192, 4, 7, EPI_STORE, false>(nullptr, nullptr)) != hipSuccess) return el;
      if ((el = launch_chain_t<16, 1, 8192, 4, 7, EPI_RESID, false>(nullptr, nullptr)) != hipSuccess) return el;
      if ((el = launch_chain_t<32, 1, 8192, 4, 7, EPI_RESID, false>(nullptr, nullptr)) != hipSuccess) return el; }
    const int rws[3] = {16, 32, 64};
    for (int i = 0; i < 3; i++) {
        hipError_t e;
        if ((e = lnbk_gemv(nullptr, rws[i], 1, EPI_STORE, 1, nullptr)) != hipSuccess) return e;
        if ((e = lnbk_gemv(nullptr, rws[i], 1, EPI_STORE, 0, nullptr)) != hipSuccess) return e;
        if ((e = lnbk_gemv(nullptr, rws[i], 1, EPI_QKV_ROPE, 1, nullptr)) != hipSuccess) return e;
        if ((e = lnbk_gemv(nullptr, rws[i], 1, EPI_RESID, 0, nullptr)) != hipSuccess) return e;
        if ((e = lnbk_gemv(nullptr, rws[i], 2, EPI_SILU_MUL, 1, nullptr)) != hipSuccess) return e;
    }
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)attn_exact_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_exact_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_gqa_kernel<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_exact_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_exact_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv2_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_long_pv2_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_one_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_one_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_one_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if (getenv("LNB_ATTN_GQA_DBG") && !g_gqa_dbg) { if (hipMalloc(&g_gqa_dbg, 8 * 16 * 8) != hipSuccess) return hipErrorOutOfMemory; (void)hipMemset(g_gqa_dbg, 0, 8 * 16 * 8); }
    done = true;
    return hipSuccess;
}

static hipError_t launch_attn_long(const AttnParams* p, hipStream_t st) {
    const size_t lds = alp_lds_bytes(p->seq_len);
    if (lds > 160 * 1024 || p->hd % ALP_DS || !p->e_buf || !p->z_part) return hipErrorInvalidValue;
    const dim3 gs(p->H, (p->seq_len + ALS_NT - 1) / ALS_NT), gp(p->H, p->hd / ALP_DS);
    const bool lazy = [] { const char* e = getenv("LNB_ATTN_LAZY"); return !(e && *e && atoi(e) == 0); }();      // (read per launch: a test switches it inside one process)
    AttnParams ps = *p;
    { const char* e = getenv("LNB_ATTN_TOUCH"); ps.touch = (e && *e) ? (atoi(e) != 0) : 1; }      // (A/B switch; default on)
    switch (p->hd) {
    // PV: the lazily certified form (attn_long_pv2_kernel, round 6) unless LNB_ATTN_LAZY=0
#define LNB_PV(HD_) do { if (lazy) hipLaunchKernelGGL(attn_long_pv2_kernel<HD_>, gp, dim3(ALP_NT), lds, st, ps); else hipLaunchKernelGGL(attn_long_pv_kernel<HD_>, gp, dim3(ALP_NT), lds, st, ps); } while (0)
    case 128: hipLaunchKernelGGL(attn_long_scores_kernel<128>, gs, dim3(ALS_NT), 0, st, ps); LNB_PV(128); break;
    case 64: hipLaunchKernelGGL(attn_long_scores_kernel<64>, gs, dim3(ALS_NT), 0, st, ps); LNB_PV(64); break;
    case 32: hipLaunchKernelGGL(attn_long_scores_kernel<32>, gs, dim3(ALS_NT), 0, st, ps); LNB_PV(32); break;
#undef LNB_PV
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// the one-launch form (attn_one_kernel): scores, softmax and PV per (head, 16-dim slice) workgroup with a bounded in-launch exchange
static hipError_t launch_attn_one(const AttnParams* p, hipStream_t st) {
    const size_t lds = att1_lds_bytes(p->seq_len, p->hd);
    if (lds > 160 * 1024 || p->hd % ALP_DS || !p->e_buf || !p->z_part || !p->cnt) return hipErrorInvalidValue;
    const dim3 g(p->H, p->hd / ALP_DS);
    switch (p->hd) {
    case 128: hipLaunchKernelGGL(attn_one_kernel<128>, g, dim3(ALP_NT), lds, st, *p); break;
    case 64: hipLaunchKernelGGL(attn_one_kernel<64>, g, dim3(ALP_NT), lds, st, *p); break;
    case 32: hipLaunchKernelGGL(attn_one_kernel<32>, g, dim3(ALP_NT), lds, st, *p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
extern "C" size_t lnbk_attn_one_lds(int seq_len, int hd) { return att1_lds_bytes(seq_len, hd); }
extern "C" size_t lnbk_attn_long_lds(int seq_len) { return alp_lds_bytes(seq_len); }
extern "C" size_t lnbk_attn_short_lds(int seq_len, int hd);

extern "C" void lnbk_attn_gqa_dbg_dump(void) {
    if (!g_gqa_dbg) return;
    (void)hipDeviceSynchronize();
    long long h[128];
    if (hipMemcpy(h, g_gqa_dbg, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int w = 0; w < 8; w++) {
        fprintf(stderr, "[gqa] wave %d: q+K %lld scores %lld pad+barrier %lld Z+cert %lld V0 store+barrier %lld PV %lld  (start %+lld)\n", w,
                h[w * 16 + 1] - h[w * 16], h[w * 16 + 2] - h[w * 16 + 1], h[w * 16 + 3] - h[w * 16 + 2], h[w * 16 + 4] - h[w * 16 + 3], h[w * 16 + 5] - h[w * 16 + 4], h[w * 16 + 6] - h[w * 16 + 5], h[w * 16] - h[0]);
    }
}
static bool attn_batch_dense() { const char* e = getenv("LNB_ATTN_BATCH_DENSE"); return !(e && *e && atoi(e) == 0); }   // (read per launch: a test switches it inside one process)
extern "C" hipError_t lnbk_attn(const AttnParams* p, hipStream_t st) {
    if (p->longctx >= 2 && p->S == 1) return launch_attn_one(p, st);
    if (p->longctx && p->S == 1) return launch_attn_long(p, st);
    if (p->mfma && p->S >= 16 && (p->hd == 128 || p->hd == 64)) {       // prefill: 16 (attn_mfma_kernel) or 32 (attn_mfma2_kernel, round 6) query rows per wave on the matrix cores
        const char* e2 = getenv("LNB_ATTN_MFMA2");                        // 0: never the two-tile form; N > 0: from N rows on (default ATM2_MIN_S)
        const int min2 = (e2 && *e2) ? atoi(e2) : ATM2_MIN_S;              // (default 0 = off: measured slower, see attn_mfma2_kernel)
        if (min2 > 0 && p->S >= min2) {
            if (p->hd == 128) hipLaunchKernelGGL(attn_mfma2_kernel<128>, dim3(p->H, (p->S + 127) / 128), dim3(256), 0, st, *p);
            else hipLaunchKernelGGL(attn_mfma2_kernel<64>, dim3(p->H, (p->S + 127) / 128), dim3(256), 0, st, *p);
            return hipGetLastError();
        }
        if (p->score_idx && p->host_T > 0 && p->sidx_jt * 16 >= p->host_T) {              // round 6: scores once, indices kept (the caller sized the scratch for this call's context)
            if (p->hd == 128) hipLaunchKernelGGL(attn_mfma3_kernel<128>, dim3(p->H, (p->S + 63) / 64), dim3(256), 0, st, *p);
            else hipLaunchKernelGGL(attn_mfma3_kernel<64>, dim3(p->H, (p->S + 63) / 64), dim3(256), 0, st, *p);
            return hipGetLastError();
        }
        if (p->hd == 128) hipLaunchKernelGGL(attn_mfma_kernel<128>, dim3(p->H, (p->S + 63) / 64), dim3(256), 0, st, *p);
        else hipLaunchKernelGGL(attn_mfma_kernel<64>, dim3(p->H, (p->S + 63) / 64), dim3(256), 0, st, *p);
        return hipGetLastError();
    }
    size_t lds = attn_lds_bytes(p->lds_T, p->hd);
    if (lds > 160 * 1024 || p->lds_T <= 0 || p->lds_T > p->seq_len) return hipErrorInvalidValue;
    if (p->host_T > p->lds_T) return hipErrorInvalidValue;      // e[] / pw[] are sized for lds_T positions: never launch past them (lnb_api.cpp check_call refuses first)
    if (p->btab && p->hd == 128 && p->H == 4 * p->KVH) {        // batched decode: one workgroup per (KV head, sequence) when that fills the chip
        const char* e = getenv("LNB_ATTN_GQA");                 // 0: never, 1: always, default: from LNB_ATTN_GQA_MIN_WG workgroups on
        const int mode = (e && *e) ? atoi(e) : -1;
        const size_t gl = gqa_lds_bytes(p->lds_T, 128, 4);
        if (mode != 0 && p->exp_tab && gl <= 160 * 1024 && (mode == 1 || (long)p->KVH * p->S >= 256)) {
            AttnParams q = *p;
            { const char* f = getenv("LNB_ATTN_GQA_FORCE_ZSEQ"); if (f && *f && atoi(f) != 0) q.force_zseq = 1; }      // (tests: every head walks the serial denominator)
            if (getenv("LNB_ATTN_GQA_DBG")) {                   // phase stamps of one workgroup (lnbk_attn_gqa_dbg_dump prints them)
                q.dbg = g_gqa_dbg;                              // (allocated by lnbk_init: no allocation inside a stream capture)
            }
            hipLaunchKernelGGL((attn_gqa_kernel<128, 4>), dim3(p->KVH, p->S), dim3(512), gl, st, q);
            return hipGetLastError();
        }
    }
    switch (p->hd) {
    case 128:
        if (p->btab && (long)p->H * p->S > 256 && attn_batch_dense()) { // more workgroups than CUs
            AttnParams q = *p;
            { const char* e = getenv("LNB_ATTN_BATCH_HEADMAJOR"); q.head_major = (e && *e && atoi(e) != 0) ? 1 : 0; }     // (A/B of the dispatch order)
            hipLaunchKernelGGL((attn_exact_kernel<128, true>), dim3(p->H, p->S), dim3(ATT_NT), lds, st, q);
        }
        else hipLaunchKernelGGL(attn_exact_kernel<128>, dim3(p->H, p->S), dim3(ATT_NT), lds, st, *p);
        break;
    case 64: hipLaunchKernelGGL(attn_exact_kernel<64>, dim3(p->H, p->S), dim3(ATT_NT), lds, st, *p); break;
    case 32: hipLaunchKernelGGL(attn_exact_kernel<32>, dim3(p->H, p->S), dim3(ATT_NT), lds, st, *p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t lnbk_embed(const uint16_t* emb, const int32_t* tokens, uint16_t* x, int S, int dim, int vocab, int* err, hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3(S), dim3(256), 0, st, emb, tokens, x, dim, vocab, err);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_argmax(const uint16_t* logits, int V, int32_t* next_token, StepState* state, int32_t* out_tokens,
                                  int out_cap, int advance, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, logits, V, next_token, state, out_tokens, out_cap, advance);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_set_state(StepState* state, int pos, int n_out, int honour_stop, hipStream_t st) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, state, pos, n_out, honour_stop);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_set_stop(StepState* state, const int32_t* ids, int n, hipStream_t st) {
    StopIds s{}; s.n = n; for (int k = 0; k < n && k < LNB_MAX_STOP_IDS; k++) s.id[k] = ids[k];
    hipLaunchKernelGGL(set_stop_kernel, dim3(1), dim3(1), 0, st, state, s);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_advance_state(StepState* state, int rows, hipStream_t st) {
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, st, state, rows);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_tile(const uint16_t* src, uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH, int gather, hipStream_t st) {
    size_t total = (size_t)rows * (K >> 3);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (gather) hipLaunchKernelGGL(tile_gather_kernel, dim3(blocks), dim3(256), 0, st, src, dst, rows, K, row_off, chain, RW, NCH);
    else hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, st, src, dst, rows, K, row_off, chain, RW, NCH);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_synth_fill(uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH,
                                      uint64_t seed, uint32_t tensor_id, int kind, float sigma, hipStream_t st) {
    uint64_t base = lnb_splitmix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(tensor_id + 1u));
    size_t total = (size_t)rows * K;
    unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(synth_fill_kernel, dim3(blocks), dim3(256), 0, st, dst, rows, K, row_off, chain, RW, NCH, base, kind, sigma);
    return hipGetLastError();
}

// ---- batched decode launchers -----------------------------------------------------------------------
extern "C" hipError_t lnbk_m16_from_tiled(const uint16_t* src, uint16_t* dst, int rows, int K, int RW, int NCH, hipStream_t st) {
    const size_t total = (size_t)rows * NCH * K;
    hipLaunchKernelGGL(m16_from_tiled_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, rows, K, RW, NCH);
    return hipGetLastError();
}
// acc2: two tile-chains per wave (fat matrices and the gate|up pairs); else one (thin matrices: every tile on its own SIMD)
extern "C" hipError_t lnbk_stream(const StreamParams* p, int epi, int acc2, int num_cus, hipStream_t st) {
    const int G = p->n_groups > 1 ? p->n_groups : 1;
    if ((p->K & 127) || p->nseq < 1 || p->nseq > G * LNB_STREAM_COLS || p->n_chains < 1 || (G > 1 && acc2)) return hipErrorInvalidValue;
    StreamParams q = *p;
    const int ACC = acc2 ? 2 : 1;
    q.n_jobs = (p->n_chains + ACC - 1) / ACC;
    unsigned grid = (unsigned)((q.n_jobs + 3) / 4); if (grid > (unsigned)num_cus) grid = (unsigned)num_cus;
    {   // thin matrices: chain wave + helper wave per tile (mfma_pair_kernel); LNB_STREAM_PAIR=0: the one-wave form
        const char* e = getenv("LNB_STREAM_PAIR");
        if (!acc2 && (G > 1 || !(e && *e && atoi(e) == 0)) && (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_QKV_ROPE)) {
            const size_t lds = (size_t)2 * 2 * MP_BUF;
            unsigned gp = (unsigned)((q.n_jobs * G + 1) / 2); if (gp > (unsigned)num_cus) gp = (unsigned)num_cus;
            if (epi == EPI_STORE) hipLaunchKernelGGL((mfma_pair_kernel<EPI_STORE>), dim3(gp), dim3(256), lds, st, q);
            else if (epi == EPI_RESID) hipLaunchKernelGGL((mfma_pair_kernel<EPI_RESID>), dim3(gp), dim3(256), lds, st, q);
            else hipLaunchKernelGGL((mfma_pair_kernel<EPI_QKV_ROPE>), dim3(gp), dim3(256), lds, st, q);
            return hipGetLastError();
        }
    }
#define LNB_STREAM(A, E) hipLaunchKernelGGL((mfma_stream_kernel<A, E>), dim3(grid), dim3(256), 0, st, q)
    switch (epi) {
    case EPI_STORE: if (acc2) LNB_STREAM(2, EPI_STORE); else LNB_STREAM(1, EPI_STORE); break;
    case EPI_RESID: if (acc2) LNB_STREAM(2, EPI_RESID); else LNB_STREAM(1, EPI_RESID); break;
    case EPI_QKV_ROPE: if (acc2) LNB_STREAM(2, EPI_QKV_ROPE); else LNB_STREAM(1, EPI_QKV_ROPE); break;
    case EPI_SILU_MUL: if (!acc2 || p->nch != 2) return hipErrorInvalidValue; LNB_STREAM(2, EPI_SILU_MUL); break;
    default: return hipErrorInvalidValue;
    }
#undef LNB_STREAM
    return hipGetLastError();
}
extern "C" hipError_t lnbk_gemm_stream(const GemmParams* p, int epi, int num_cus, hipStream_t st);
// a one-wave kernel that occupies its stream for `us` microseconds of the constant-rate wall clock (measurement aid: lnb_runtime_info's count of
// streams that really run concurrently; the delayed second launch of tools/ffn_overlap.py)
__global__ void spin_kernel(long long ticks) { const long long t0 = (long long)wall_clock64(); while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
extern "C" hipError_t lnbk_spin(int us, hipStream_t st) {
    int khz = 100000, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (long long)us * khz / 1000);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_prepare(void) {             // raise the dynamic-LDS limits once, outside any stream capture
    static bool done = false;
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)batch_rmsnorm_xt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mfma_pair_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mfma_pair_kernel<EPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mfma_pair_kernel<EPI_QKV_ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int ep = EPI_STORE; ep <= EPI_SILU_MUL && e == hipSuccess; ep++) e = lnbk_gemm_stream(nullptr, ep, 0, nullptr);
    done = e == hipSuccess;
    return e;
}
extern "C" hipError_t lnbk_batch_rmsnorm(const uint16_t* x, const uint16_t* norm_w, float eps, uint16_t* xt, int K, int nseq, hipStream_t st) {
    const size_t lds = bn_scratch() + ((size_t)bn_kpad(K) + 8) * 4;
    if ((K & 127) || lds > 160 * 1024 || (size_t)bn_kpad(K) > (size_t)XCh<true>::value * (1 + BN_NH) * 512 || seq_leaf_size(K, rms_nf(BN_NH) * 64) > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(batch_rmsnorm_xt_kernel<true>, dim3((unsigned)nseq), dim3((1 + BN_NH) * 64), lds, st, x, norm_w, eps, xt, K);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_embed(const uint16_t* emb, const BatchTab* tab, uint16_t* x, int nseq, int dim, int vocab, int* err, hipStream_t st) {
    hipLaunchKernelGGL(batch_embed_kernel, dim3((unsigned)nseq), dim3(256), 0, st, emb, tab, x, dim, vocab, err);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_argmax(const uint16_t* logits, int V, const BatchTab* tab, int nseq, int32_t* ring, hipStream_t st) {
    hipLaunchKernelGGL(batch_argmax_kernel, dim3((unsigned)nseq), dim3(1024), 0, st, logits, V, tab, ring);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_set_state(const BatchTab* tab, const int32_t* tokens, const int32_t* pos, int32_t* ring, int honour_stop, hipStream_t st) {
    hipLaunchKernelGGL(batch_set_state_kernel, dim3(1), dim3(LNB_BATCH_MAX), 0, st, tab, tokens, pos, ring, honour_stop);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_scatter_ring(const BatchTab* tab, const int32_t* ring, hipStream_t st) {
    hipLaunchKernelGGL(batch_scatter_ring_kernel, dim3(1), dim3(LNB_BATCH_MAX), 0, st, tab, ring);
    return hipGetLastError();
}
extern "C" hipError_t lnbk_batch_advance(const BatchTab* tab, hipStream_t st) {
    hipLaunchKernelGGL(batch_advance_kernel, dim3(1), dim3(LNB_BATCH_MAX), 0, st, tab);
    return hipGetLastError();
}

// prefill product on the streaming matrix-core feed (gemm_stream_kernel): one weight tile per wave and NTW batch tiles of 16 rows.
// (batch tiles per wave and dispatch order: lnb_gemm_stream_ntw / lnb_gemm_stream_rows_fastest, lnb_device.h -- host logic, tests/test_layouts.py)
template <int EPI, int NCH> static hipError_t launch_gemm_stream(const GemmParams* p, int num_cus, hipStream_t st) {
    if (!p) {
        hipError_t e = hipSuccess;
        // every instantiation the launch below can pick (the resident-layout sources SRC 1 / 2 request up to 66,560 B at four batch
        // tiles per wave -- above the 64 KiB a kernel gets without this attribute)
#define LNB_GS_PREP(N, SRC_) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, NCH, N, SRC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        LNB_GS_PREP(1, 0); LNB_GS_PREP(2, 0); LNB_GS_PREP(4, 0);
        LNB_GS_PREP(1, 2); LNB_GS_PREP(2, 2); LNB_GS_PREP(4, 2);
        if constexpr (NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) { LNB_GS_PREP(1, 1); LNB_GS_PREP(2, 1); LNB_GS_PREP(4, 1); }
        if constexpr (NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) {        // two weight tiles per wave (TT)
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, NCH, 4, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, NCH, 4, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
#undef LNB_GS_PREP
        return e;
    }
    // A operand: the M16 copy when the model carries one, else the RESIDENT layout (round 5: no second copy needed for a prompt) -- the row-broadcast
    // layout's units are M16 units in another order (src 1), the chain layouts' units hold eight consecutive k and are transposed inside the quad (src 2)
    const int src = p->w16 ? 0 : (p->w && p->rw == 4 && NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) ? 1 : (p->w && p->rw >= 16 && p->nch == NCH) ? 2 : -1;
    if (src < 0 || (p->K & 127) || p->S < 1) return hipErrorInvalidValue;
    const int n_tiles = (p->n_rows + 15) / 16;
    const int ct = (p->S + 15) / 16;                         // batch tiles of 16 rows
    int ntw = lnb_gemm_stream_ntw(n_tiles, ct, NCH, num_cus);
    const int force = getenv("LNB_GS_NTW") ? atoi(getenv("LNB_GS_NTW")) : 0;    // (tools / experiments / tests; read per launch)
    if (force == 1 || force == 2 || force == 4) ntw = force;
    static const int force_ct = getenv("LNB_GS_NTW_CHAIN") ? atoi(getenv("LNB_GS_NTW_CHAIN")) : 0;    // (experiments: the chain layouts only)
    if (src == 2 && (force_ct == 1 || force_ct == 2 || force_ct == 4)) ntw = force_ct;
    GemmParams q = *p;
    // round 6: few batch rows on the chain layouts -- the one-k matrix instruction with FOUR WEIGHT blocks and the B operand broadcast by BLGP (gemm_blgp_kernel):
    // no transpose, no second copy.  Bit-exact and NOT the default: measured slower than the row-swap transpose it was meant to replace (profiles/r06_blgp.log: wq|wk|wv
    // of a 128-token prompt 125-130 us against 91, gate|up 346 against 301; whole Forward 16 / 128 rows 13.2 / 20.7 ms against 9.2 / 19.4) -- a wave carries FOUR weight
    // tiles as ONE dependent chain of 4 K one-k instructions with an unpack op in front of each (67 cycles per instruction instead of 32), and a quarter as many waves
    // to spread over the SIMDs (96 for wq|wk|wv at 16 rows).  LNB_GEMM_BLGP=1: where gemm_stream_kernel would run one or two batch tiles per wave; =2: every chain-layout product
    {
        const char* eb = getenv("LNB_GEMM_BLGP");            // (read per launch: a test switches it inside one process)
        const int blgp = (eb && *eb) ? atoi(eb) : 0;
        if (src == 2 && blgp && (ntw <= 2 || blgp == 2) && p->rw % 8 == 0) {
            const int tpg = 4 / NCH, n_tg = (n_tiles + tpg - 1) / tpg;
            const dim3 g((unsigned)((n_tg + 3) / 4), (unsigned)ct);
            const size_t l = (size_t)2 * 16 * GS_PITCH * 4;
            hipLaunchKernelGGL((gemm_blgp_kernel<EPI, NCH>), g, dim3(256), l, st, q);
            return hipGetLastError();
        }
    }
    const int rows_wg = 16 * ntw;
    // round 6: wo / w2 with many rows -- two neighbouring weight tiles per wave (gemm_stream_kernel's TT form: the gate|up product's shape), as long as the tile PAIRS still
    // give every CU its two workgroups.  LNB_GS_TT=0: never, 1: whenever the form exists
    bool tt = false;
    if constexpr (NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) {
        const char* et = getenv("LNB_GS_TT");                // (read per launch: a test switches it inside one process)
        const int ttm = (et && *et) ? atoi(et) : -1;
        const long wgs = (long)(((n_tiles + 1) / 2 + 3) / 4) * ((p->S + rows_wg - 1) / rows_wg);
        tt = ntw == 4 && (src == 0 || src == 1) && ttm != 0 && (ttm == 1 || wgs >= 2L * num_cus);
    }
    unsigned gx = (unsigned)(((tt ? (n_tiles + 1) / 2 : n_tiles) + 3) / 4); if (gx > (unsigned)num_cus) gx = (unsigned)num_cus;
    const dim3 grid(gx, (unsigned)((p->S + rows_wg - 1) / rows_wg));
    // Dispatch order (workgroup id % 8 = XCD, each with its own L2).  Weight-tile groups fastest: an XCD owns 1/8 of the tile groups for every
    // row group -- right while all row groups are in flight together (<= 8 of them: short prompts, batches), each weight byte is then fetched
    // once; with many row groups the same tile group comes back once per row group (4096 rows: 15.6 GB of fabric reads per gate|up launch
    // for 235 MB of weights, rocprofv3 FETCH_SIZE).  Row groups fastest: the workgroups in flight cover ~8 tile groups x all row groups, a
    // weight tile is fetched once per XCD and shared through its L2 (4.8 GB; 1-4 % faster at 4096 rows, 2-3 % slower at 512).
    static const int order = getenv("LNB_GS_ORDER") ? atoi(getenv("LNB_GS_ORDER")) : -1;
    q.rows_fastest = order >= 0 ? order : lnb_gemm_stream_rows_fastest((int)grid.y);
    const size_t lds = (size_t)2 * rows_wg * GS_PITCH * 4;
#define LNB_GS_LAUNCH(SRC_) switch (ntw) { \
    case 1: hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, 1, SRC_>), grid, dim3(256), lds, st, q); break; \
    case 2: hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, 2, SRC_>), grid, dim3(256), lds, st, q); break; \
    default: hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, 4, SRC_>), grid, dim3(256), lds, st, q); break; }
    if constexpr (NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) {
        if (tt) {
            if (src == 0) hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, 4, 0, 1>), grid, dim3(256), lds, st, q);
            else hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, 4, 1, 1>), grid, dim3(256), lds, st, q);
            return hipGetLastError();
        }
    }
    if (src == 0) { LNB_GS_LAUNCH(0) }
    else if (src == 2) { LNB_GS_LAUNCH(2) }
    else if constexpr (NCH == 1 && (EPI == EPI_STORE || EPI == EPI_RESID)) { LNB_GS_LAUNCH(1) }
#undef LNB_GS_LAUNCH
    return hipGetLastError();
}
extern "C" hipError_t lnbk_gemm_stream(const GemmParams* p, int epi, int num_cus, hipStream_t st) {
    const int nch = p ? p->nch : 0;
    switch (epi) {
        case EPI_STORE: return launch_gemm_stream<EPI_STORE, 1>(p, num_cus, st);
        case EPI_RESID: return launch_gemm_stream<EPI_RESID, 1>(p, num_cus, st);
        case EPI_QKV_ROPE: return launch_gemm_stream<EPI_QKV_ROPE, 1>(p, num_cus, st);
        case EPI_SILU_MUL: return (p && nch != 2) ? hipErrorInvalidValue : launch_gemm_stream<EPI_SILU_MUL, 2>(p, num_cus, st);
        default: return hipErrorInvalidValue;
    }
}
