// chainbench3.hip -- round 4: how cheaply can ONE wave be fed the per-lane operands of a k-ordered f32 chain? (gfx950)
//   plain / 16-lane / DPP forms of the dependent add;
//   products fetched from the LDS (ds_read_b128 per 4 steps) with 64 / 32 / 16 lanes active;
//   products made by the MATRIX pipe: v_mfma_f32_32x32x16_bf16 with a one-hot "selector" A operand (x_k at one k per D row,
//   zeros elsewhere) and the raw bf16 weights as B: D[r][row] = x_k(r) * w[row][k(r)] exactly -- 16 products per lane per issue slot.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chainbench3.hip -o tools/chainbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)

// ---- 1. dependent adds ---------------------------------------------------------------------------
template <int LANES>
__global__ void k_plain(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < LANES) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) asm volatile(REP16("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\t") : "+v"(acc) : "v"(p0), "v"(p1));
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void k_dpp_bcast(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t") : "+v"(acc) : "v"(p0), "v"(p1));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void k_dpp_quad(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t") : "+v"(acc) : "v"(p0), "v"(p1));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
// 16 dependent adds + FILL independent VALU ops (the producer work a DPP chain wave does for itself)
template <int FILL>
__global__ void k_plain_fill(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f, f0 = 1.0f, f1 = 2.0f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters * 2; i++) {
        asm volatile(REP4("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\t") : "+v"(acc) : "v"(p0), "v"(p1));
        if (FILL >= 1) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(f0) : "v"(p0));
        if (FILL >= 2) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(f1) : "v"(p1));
        if (FILL >= 4) asm volatile("v_mul_f32 %0, %1, %1\n\tv_mul_f32 %0, %1, %1" : "=v"(f1) : "v"(p1));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = acc + f0 + f1;
}

// ---- 2. products from the LDS: 8-deep ring of ds_read_b128, 4 dependent adds per read -------------------------------
#define LQ(r0, r1, r2, r3, off) \
    "s_waitcnt lgkmcnt(7)\n\t" \
    "v_add_f32 %[acc], v" #r0 ", %[acc]\n\tv_add_f32 %[acc], v" #r1 ", %[acc]\n\tv_add_f32 %[acc], v" #r2 ", %[acc]\n\tv_add_f32 %[acc], v" #r3 ", %[acc]\n\t" \
    "ds_read_b128 v[" #r0 ":" #r3 "], %[ad] offset:" #off "\n\t"
#define LP(r0, r3, off) "ds_read_b128 v[" #r0 ":" #r3 "], %[ad] offset:" #off "\n\t"
template <int LANES>
__global__ void k_lds(float* out, long long* ticks, int iters, float a, float b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = b * (float)(i & 7);
    __syncthreads();
    float acc = a;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < LANES) {
        const unsigned ad = threadIdx.x * 16u;         // 16 B per lane, conflict-free
        int n = iters;
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile(
            LP(40, 43, 0) LP(44, 47, 1024) LP(48, 51, 2048) LP(52, 55, 3072) LP(56, 59, 4096) LP(60, 63, 5120) LP(64, 67, 6144) LP(68, 71, 7168)
            "L_loop_%=:\n\t"
            LQ(40, 41, 42, 43, 8192) LQ(44, 45, 46, 47, 9216) LQ(48, 49, 50, 51, 10240) LQ(52, 53, 54, 55, 11264)
            LQ(56, 57, 58, 59, 12288) LQ(60, 61, 62, 63, 13312) LQ(64, 65, 66, 67, 14336) LQ(68, 69, 70, 71, 15360)
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_loop_%=\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
              "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "scc", "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + (threadIdx.x & 63)] = acc;
}


// ---- 2b. products from the LDS, delivered to the chain by DPP: one ds_read_b128 feeds 16 steps (quad_perm: 16 rows x 4 lanes per wave)
//          or 64 steps (row_newbcast: 4 rows x 16 lanes per wave) ------------------------------------------------------------------
#define QP(j) " quad_perm:[" #j "," #j "," #j "," #j "] row_mask:0xf bank_mask:0xf\n\t"
#define QADD4(j, r0, r1, r2, r3) "v_add_f32_dpp %[acc], v" #r0 ", %[acc]" QP(j) "v_add_f32_dpp %[acc], v" #r1 ", %[acc]" QP(j) "v_add_f32_dpp %[acc], v" #r2 ", %[acc]" QP(j) "v_add_f32_dpp %[acc], v" #r3 ", %[acc]" QP(j)
#define QG(r0, r1, r2, r3, off) "s_waitcnt lgkmcnt(3)\n\ts_nop 1\n\t" QADD4(0, r0, r1, r2, r3) QADD4(1, r0, r1, r2, r3) QADD4(2, r0, r1, r2, r3) QADD4(3, r0, r1, r2, r3) \
    "ds_read_b128 v[" #r0 ":" #r3 "], %[ad] offset:" #off "\n\t"
__global__ void k_lds_quad(float* out, long long* ticks, int iters, float a, float b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = b * (float)(i & 7);
    __syncthreads();
    float acc = a;
    long long t0 = 0, t1 = 0;
    if (threadIdx.x < 64) {
        const unsigned ad = threadIdx.x * 16u;
        int n = iters / 2;                                   // 64 steps per loop iteration
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile(
            LP(40, 43, 0) LP(44, 47, 1024) LP(48, 51, 2048) LP(52, 55, 3072)
            "L_loop_%=:\n\t"
            QG(40, 41, 42, 43, 4096) QG(44, 45, 46, 47, 5120) QG(48, 49, 50, 51, 6144) QG(52, 53, 54, 55, 7168)
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_loop_%=\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "scc", "memory");
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        // helper waves: keep writing products into the other half of the LDS at about the production rate of a 24-row CU
        float4 v = make_float4(a, b, a, b);
        for (int i = 0; i < iters * 2; i++) {
            *(float4*)(lds + 8192 + ((threadIdx.x * 4 + i * 1024) & 8191)) = v; asm volatile("" ::: "memory");
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + (threadIdx.x & 63)] = acc;
}
#define RB(j) " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
#define BADD16(r) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(0) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(1) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(2) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(3) \
    "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(4) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(5) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(6) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(7) \
    "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(8) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(9) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(10) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(11) \
    "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(12) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(13) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(14) "v_add_f32_dpp %[acc], v" #r ", %[acc]" RB(15)
#define BG(r0, r1, r2, r3, off) "s_waitcnt lgkmcnt(1)\n\ts_nop 1\n\t" BADD16(r0) BADD16(r1) BADD16(r2) BADD16(r3) "ds_read_b128 v[" #r0 ":" #r3 "], %[ad] offset:" #off "\n\t"
__global__ void k_lds_bcast(float* out, long long* ticks, int iters, float a, float b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = b * (float)(i & 7);
    __syncthreads();
    float acc = a;
    long long t0 = 0, t1 = 0;
    if (threadIdx.x < 64) {
        const unsigned ad = threadIdx.x * 16u;
        int n = iters / 4;                                   // 128 steps per loop iteration
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile(
            LP(40, 43, 0) LP(44, 47, 1024)
            "L_loop_%=:\n\t"
            BG(40, 41, 42, 43, 4096) BG(44, 45, 46, 47, 5120)
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_loop_%=\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "scc", "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + (threadIdx.x & 63)] = acc;
}

// ---- 3. products from the matrix pipe ----------------------------------------------------------------------------------
// selector A for MFMA lane l (row i = l & 31 of A, k-group h = l >> 5 holding k = 8h .. 8h+7): D row i carries k-slot
// pi(i) = 4 (i >> 3) + (i & 3)  (both D lane halves then hold slots 0..15 in register order); A[i][kk] = x[kk] iff kk == pi(i).
__device__ __forceinline__ bf16x8 selector(int lane, const uint16_t* x16) {
    const int i = lane & 31, h = lane >> 5, pi = 4 * (i >> 3) + (i & 3);
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (8 * h + e == pi) ? x16[pi] : (unsigned short)0;
    u32x4 r = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void adds16(float& acc, const f32x16& d) {
    asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t"
                 "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0\n\t"
                 "v_add_f32 %0, %9, %0\n\tv_add_f32 %0, %10, %0\n\tv_add_f32 %0, %11, %0\n\tv_add_f32 %0, %12, %0\n\t"
                 "v_add_f32 %0, %13, %0\n\tv_add_f32 %0, %14, %0\n\tv_add_f32 %0, %15, %0\n\tv_add_f32 %0, %16, %0\n\t"
                 : "+v"(acc) : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]),
                   "v"(d[8]), "v"(d[9]), "v"(d[10]), "v"(d[11]), "v"(d[12]), "v"(d[13]), "v"(d[14]), "v"(d[15]));
}
// MODE 0: bare (operands fixed in registers); 1: + one ds_read_b128 per MFMA (the selector from an LDS table);
//      2: + one global_load_dwordx4 per MFMA (the weights, L2-resident ring of 8 in flight) as well
template <int MODE>
__global__ void k_mfma(float* out, long long* ticks, int iters, const uint16_t* x16, const u32x4* wsrc) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    u32x4* tab = (u32x4*)lds;                                // [64 blocks][64 lanes] selectors (this bench: one entry per lane and block)
    for (int b = 0; b < 64; b++) tab[b * 64 + lane] = __builtin_bit_cast(u32x4, selector(lane, x16 + 16 * (b & 3)));
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 asel = selector(lane, x16);
    u32x4 wv = wsrc[lane];
    float acc = 0.0f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asel, __builtin_bit_cast(bf16x8, wv), zero, 0, 0, 0), d1;
    u32x4 a_nx = tab[lane], w_nx = wsrc[64 + lane];
    for (int i = 0; i < iters; i++) {
        // block 2i+1's products are made while block 2i's are added, and vice versa
        if (MODE >= 1) { asel = __builtin_bit_cast(bf16x8, a_nx); a_nx = tab[((2 * i + 1) & 63) * 64 + lane]; }
        if (MODE >= 2) { wv = w_nx; w_nx = wsrc[(size_t)((2 * i + 2) & 255) * 64 + lane]; }
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asel, __builtin_bit_cast(bf16x8, wv), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        adds16(acc, d0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 1) { asel = __builtin_bit_cast(bf16x8, a_nx); a_nx = tab[((2 * i + 2) & 63) * 64 + lane]; }
        if (MODE >= 2) { wv = w_nx; w_nx = wsrc[(size_t)((2 * i + 3) & 255) * 64 + lane]; }
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asel, __builtin_bit_cast(bf16x8, wv), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        adds16(acc, d1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = acc + d0[0];
}

// exactness: D[r][row] from the matrix pipe against the plain f32 product, random bf16 incl. subnormals / zeros / negatives
__global__ void k_mfma_exact(const uint16_t* w /*[32 rows][16 k]*/, const uint16_t* x16 /*[16]*/, uint32_t* got /*[64][16]*/, uint32_t* want) {
    const int lane = threadIdx.x & 63, row = lane & 31, h = lane >> 5;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bf16x8 asel = selector(lane, x16);
    const u32x4 wv = *(const u32x4*)(w + (size_t)row * 16 + 8 * h);       // B lane (col = row, k-group h): k = 8h .. 8h+7
    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asel, __builtin_bit_cast(bf16x8, wv), zero, 0, 0, 0);
    for (int r = 0; r < 16; r++) {
        got[lane * 16 + r] = __float_as_uint(d[r]);
        const float xv = __uint_as_float((uint32_t)x16[r] << 16), wf = __uint_as_float((uint32_t)w[row * 16 + r] << 16);
        want[lane * 16 + r] = __float_as_uint(xv * wf);
    }
}

static double g_ghz = 2.4;
template <typename F, typename... A> static void timeit(const char* name, F kern, int block, size_t lds, int steps_per_iter, A... args) {
    float* out; long long* ticks;
    (void)hipMalloc((void**)&out, 1 << 20); (void)hipMalloc((void**)&ticks, 4096);
    const int iters = 8192;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(block), lds < 100 * 1024 ? 100 * 1024 : lds, 0, out, ticks, iters, args...);   // 100 KiB LDS: one workgroup per CU
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    hipError_t e = hipGetLastError();
    long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double ns = ms * 1e6 / ((double)iters * steps_per_iter);
    printf("%-44s: %.3f ns/step (%.2f cyc @%.1f GHz; %.2f memtime ticks/step)%s\n", name, ns, ns * g_ghz, g_ghz, (double)h / ((double)iters * steps_per_iter), e == hipSuccess ? "" : hipGetErrorString(e));
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    timeit("v_add_f32 dependent, 64 lanes", k_plain<64>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("v_add_f32 dependent, 16 lanes", k_plain<16>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("v_add_f32_dpp row_newbcast", k_dpp_bcast, 64, 0, 32, 1.0f, 1e-3f);
    timeit("v_add_f32_dpp quad_perm", k_dpp_quad, 64, 0, 32, 1.0f, 1e-3f);
    timeit("16 adds + 0 independent VALU", k_plain_fill<0>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("16 adds + 1 independent VALU", k_plain_fill<1>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("16 adds + 2 independent VALU", k_plain_fill<2>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("16 adds + 4 independent VALU", k_plain_fill<4>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("LDS-fed: ds_read_b128 / 4 adds, 64 lanes", k_lds<64>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("LDS-fed: ds_read_b128 / 4 adds, 32 lanes", k_lds<32>, 64, 0, 32, 1.0f, 1e-3f);
    timeit("LDS-fed: ds_read_b128 / 4 adds, 16 lanes", k_lds<16>, 64, 0, 32, 1.0f, 1e-3f);

    timeit("LDS-fed quad DPP: ds_read_b128 / 16 adds", k_lds_quad, 64, 0, 32, 1.0f, 1e-3f);
    timeit("  ... with 4 helper waves writing the LDS", k_lds_quad, 320, 0, 32, 1.0f, 1e-3f);
    timeit("LDS-fed row_newbcast: ds_read_b128 / 64 adds", k_lds_bcast, 64, 0, 32, 1.0f, 1e-3f);

    // matrix-pipe products
    std::vector<uint16_t> hw(32 * 16), hx(64);
    uint64_t s = 0x1234567ull;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    uint16_t *dx, *dw; u32x4* dwsrc; uint32_t *dgot, *dwant;
    (void)hipMalloc((void**)&dx, 128); (void)hipMalloc((void**)&dw, 32 * 16 * 2); (void)hipMalloc((void**)&dwsrc, 256 * 64 * 16);
    (void)hipMalloc((void**)&dgot, 64 * 16 * 4); (void)hipMalloc((void**)&dwant, 64 * 16 * 4);
    long long bad = 0, total = 0, nsub = 0;
    for (int trial = 0; trial < 2000; trial++) {
        for (auto& v : hw) {
            const uint32_t r = rnd();
            if (trial % 4 == 0) v = (uint16_t)r;                                                        // any bit pattern but inf / NaN
            else if (trial % 4 == 1) v = (uint16_t)((r & 0x807F) | ((1 + r % 40) << 7));                // tiny: products underflow into subnormals
            else v = (uint16_t)((r & 0x807F) | ((100 + (r >> 16) % 50) << 7));                          // ordinary range
            if (((v >> 7) & 0xFF) == 0xFF) v &= 0xBFFF;
        }
        for (auto& v : hx) {
            const uint32_t r = rnd();
            v = trial % 4 == 1 ? (uint16_t)((r & 0x807F) | ((60 + r % 40) << 7)) : (trial % 4 == 0 ? (uint16_t)r : (uint16_t)((r & 0x807F) | ((110 + (r >> 16) % 30) << 7)));
            if (((v >> 7) & 0xFF) == 0xFF) v &= 0xBFFF;
            if (trial % 7 == 3 && (r & 3) == 0) v = (uint16_t)(r & 0x8000);                             // +-0
        }
        (void)hipMemcpy(dx, hx.data(), 128, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), 32 * 16 * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma_exact, dim3(1), dim3(64), 0, 0, dw, dx, dgot, dwant);
        std::vector<uint32_t> g(64 * 16), w(64 * 16);
        (void)hipMemcpy(g.data(), dgot, g.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(w.data(), dwant, w.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < g.size(); i++) {
            total++;
            const uint32_t ex = (w[i] >> 23) & 0xFF;
            if (ex == 0 && (w[i] & 0x7FFFFF)) nsub++;
            const bool same = g[i] == w[i] || ((g[i] | w[i]) << 1) == 0;                                // +0 and -0 products are interchangeable in the chain
            if (!same) { if (bad < 8) printf("  mismatch trial %d lane %zu reg %zu: mfma %08x  mul %08x\n", trial, i / 16, i % 16, g[i], w[i]); bad++; }
        }
    }
    printf("matrix-pipe selector products vs v_mul_f32: %lld / %lld differ (%lld subnormal products among them)\n", bad, total, nsub);
    std::vector<uint32_t> fill(256 * 64 * 4);
    for (auto& v : fill) { const uint32_t r = rnd(); v = ((r & 0x807F) | (120u << 7)) | (((r >> 16 & 0x807F) | (121u << 7)) << 16); }
    (void)hipMemcpy(dwsrc, fill.data(), fill.size() * 4, hipMemcpyHostToDevice);
    for (auto& v : hx) v = (uint16_t)(0x3F80 + (rnd() & 0x3F));
    (void)hipMemcpy(dx, hx.data(), 128, hipMemcpyHostToDevice);
    timeit("MFMA-fed: 1 mfma / 16 adds (bare)", k_mfma<0>, 64, 0, 32, (const uint16_t*)dx, (const u32x4*)dwsrc);
    timeit("MFMA-fed: + ds_read_b128 selector", k_mfma<1>, 64, 0, 32, (const uint16_t*)dx, (const u32x4*)dwsrc);
    timeit("MFMA-fed: + selector + global weights", k_mfma<2>, 64, 0, 32, (const uint16_t*)dx, (const u32x4*)dwsrc);
    return 0;
}
