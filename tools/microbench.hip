// microbench.hip -- gfx950 measurements that size the exact-order design:
//   1. dependent v_fmac_f32 chain latency (cycles per dependent op) at 1/2/4 waves per SIMD
//   2. the real inner-loop mix (unpack + fmac, 2 VALU per element) per k-step
//   3. v_dot2c_f32_bf16 with a zero partner: latency, and whether it is bit-identical to the fmaf chain
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_dep_fma(float* out, long long* cyc, int iters, float a, float b) {
    float acc = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) acc = fmaf(a, b, acc);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dep_mix(float* out, long long* cyc, int iters, const uint32_t* w, float a) {
    float acc = threadIdx.x;
    uint32_t v = w[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            float lo = __uint_as_float((v + j) << 16), hi = __uint_as_float((v + j) & 0xffff0000u);
            acc = fmaf(a, lo, acc); acc = fmaf(a, hi, acc);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__global__ void k_dep_dot2(float* out, long long* cyc, int iters, const uint32_t* w, uint32_t xlo, uint32_t xhi) {
    float acc = threadIdx.x;
    uint32_t v = w[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            uint32_t wv = v + j;
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, wv), __builtin_bit_cast(v2bf, xlo), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, wv), __builtin_bit_cast(v2bf, xhi), acc, false);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// exactness: chain of K steps with random bf16 data, fmaf vs dot2c(zero partner)
__global__ void k_exact(const uint32_t* w, const uint32_t* x, int K2, uint32_t* o_fma, uint32_t* o_dot) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.f, d = 0.f;
    for (int k = 0; k < K2; k++) {
        uint32_t wv = w[(size_t)t * K2 + k], xv = x[k];
        a = fmaf(__uint_as_float(xv << 16), __uint_as_float(wv << 16), a);
        a = fmaf(__uint_as_float(xv & 0xffff0000u), __uint_as_float(wv & 0xffff0000u), a);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, wv), __builtin_bit_cast(v2bf, xv & 0xffffu), d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, wv), __builtin_bit_cast(v2bf, xv & 0xffff0000u), d, false);
    }
    o_fma[t] = __float_as_uint(a); o_dot[t] = __float_as_uint(d);
}


#define DPPQ(j) " quad_perm:[" #j "," #j "," #j "," #j "] row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void fmac16_dpp(float& acc, const float4& x4, const float4& w0, const float4& w1, const float4& w2, const float4& w3) {
    asm volatile("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %1, %5" DPPQ(0)  "v_fmac_f32_dpp %0, %2, %6" DPPQ(0)  "v_fmac_f32_dpp %0, %3, %7" DPPQ(0)  "v_fmac_f32_dpp %0, %4, %8" DPPQ(0)
        "v_fmac_f32_dpp %0, %1, %9" DPPQ(1)  "v_fmac_f32_dpp %0, %2, %10" DPPQ(1) "v_fmac_f32_dpp %0, %3, %11" DPPQ(1) "v_fmac_f32_dpp %0, %4, %12" DPPQ(1)
        "v_fmac_f32_dpp %0, %1, %13" DPPQ(2) "v_fmac_f32_dpp %0, %2, %14" DPPQ(2) "v_fmac_f32_dpp %0, %3, %15" DPPQ(2) "v_fmac_f32_dpp %0, %4, %16" DPPQ(2)
        "v_fmac_f32_dpp %0, %1, %17" DPPQ(3) "v_fmac_f32_dpp %0, %2, %18" DPPQ(3) "v_fmac_f32_dpp %0, %3, %19" DPPQ(3) "v_fmac_f32_dpp %0, %4, %20" DPPQ(3)
        : "+v"(acc)
        : "v"(x4.x), "v"(x4.y), "v"(x4.z), "v"(x4.w),
          "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w), "v"(w1.x), "v"(w1.y), "v"(w1.z), "v"(w1.w),
          "v"(w2.x), "v"(w2.y), "v"(w2.z), "v"(w2.w), "v"(w3.x), "v"(w3.y), "v"(w3.z), "v"(w3.w));
}
__device__ __forceinline__ void fmac16_plain(float& acc, const float4& x4, const float4& w0, const float4& w1, const float4& w2, const float4& w3) {
    asm volatile(
        "v_fmac_f32 %0, %1, %5\n\tv_fmac_f32 %0, %2, %6\n\tv_fmac_f32 %0, %3, %7\n\tv_fmac_f32 %0, %4, %8\n\t"
        "v_fmac_f32 %0, %1, %9\n\tv_fmac_f32 %0, %2, %10\n\tv_fmac_f32 %0, %3, %11\n\tv_fmac_f32 %0, %4, %12\n\t"
        "v_fmac_f32 %0, %1, %13\n\tv_fmac_f32 %0, %2, %14\n\tv_fmac_f32 %0, %3, %15\n\tv_fmac_f32 %0, %4, %16\n\t"
        "v_fmac_f32 %0, %1, %17\n\tv_fmac_f32 %0, %2, %18\n\tv_fmac_f32 %0, %3, %19\n\tv_fmac_f32 %0, %4, %20\n\t"
        : "+v"(acc)
        : "v"(x4.x), "v"(x4.y), "v"(x4.z), "v"(x4.w),
          "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w), "v"(w1.x), "v"(w1.y), "v"(w1.z), "v"(w1.w),
          "v"(w2.x), "v"(w2.y), "v"(w2.z), "v"(w2.w), "v"(w3.x), "v"(w3.y), "v"(w3.z), "v"(w3.w));
}
// mode 0: registers only, dpp;  1: registers only, plain fmac;  2: LDS operands, prefetch distance 1, dpp;  3: distance 2, dpp
// 4: LDS operands distance 2, plain fmac
template <int MODE>
__global__ void k_block(float* out, long long* cyc, int iters, int active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    float acc = 0.f;
    const float* px = (const float*)smem + (lane & 3) * 4;
    const char* src = smem + 16384 + lane * 16;
    long long t0 = clock64();
    if (lane < active) {
        if (MODE <= 1) {
            float4 x = *(const float4*)px, w0 = *(const float4*)src, w1 = *(const float4*)(src + 1024), w2 = *(const float4*)(src + 2048), w3 = *(const float4*)(src + 3072);
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int g = 0; g < 16; g++) { if (MODE == 0) fmac16_dpp(acc, x, w0, w1, w2, w3); else fmac16_plain(acc, x, w0, w1, w2, w3); }
            }
        } else {
            constexpr int DIST = (MODE == 2) ? 1 : 2;
            for (int i = 0; i < iters; i++) {
                float4 xq[3], w[3][4];
#pragma unroll
                for (int d = 0; d < DIST; d++) { xq[d] = *(const float4*)(px + d * 16); for (int j = 0; j < 4; j++) w[d][j] = *(const float4*)(src + ((4 * d + j) * 16) * 16); }
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int cur = g % (DIST + 1), nxt = (g + DIST) % (DIST + 1);
                    if (g + DIST < 16) { xq[nxt] = *(const float4*)(px + (g + DIST) * 16); for (int j = 0; j < 4; j++) w[nxt][j] = *(const float4*)(src + ((4 * (g + DIST) + j) * 16) * 16); }
                    if (MODE == 4) fmac16_plain(acc, xq[cur], w[cur][0], w[cur][1], w[cur][2], w[cur][3]);
                    else fmac16_dpp(acc, xq[cur], w[cur][0], w[cur][1], w[cur][2], w[cur][3]);
                }
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc; uint32_t* w;
    CHK(hipMalloc(&out, 1 << 20)); CHK(hipMalloc(&cyc, 4096 * 8)); CHK(hipMalloc(&w, 1 << 16));
    CHK(hipMemset(w, 0x3f, 1 << 16));
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k_block<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); hipFuncSetAttribute((const void*)k_block<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); hipFuncSetAttribute((const void*)k_block<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); hipFuncSetAttribute((const void*)k_block<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); hipFuncSetAttribute((const void*)k_block<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
    int blocks[] = {1, 256};
    int threads[] = {64, 256, 512, 1024};
    for (int kind = 0; kind < 3; kind++) for (int bi = 0; bi < 2; bi++) for (int ti = 0; ti < 4; ti++) {
        int nb = blocks[bi], nt = threads[ti];
        long long h[4096];
        for (int rep = 0; rep < 2; rep++) {
            if (kind == 0) hipLaunchKernelGGL(k_dep_fma, dim3(nb), dim3(nt), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
            if (kind == 1) hipLaunchKernelGGL(k_dep_mix, dim3(nb), dim3(nt), 0, 0, out, cyc, iters, w, 1.0001f);
            if (kind == 2) hipLaunchKernelGGL(k_dep_dot2, dim3(nb), dim3(nt), 0, 0, out, cyc, iters, w, 0x00003f80u, 0x3f800000u);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < nb; i++) if (h[i] > mx) mx = (double)h[i];
        const char* names[] = {"dep_fmac", "unpack+fmac (per element)", "dot2c zero-partner (per element)"};
        printf("%-34s blocks=%3d threads=%4d (waves/SIMD=%.2f): %.2f clk64-ticks per chain step\n", names[kind], nb, nt, nt / 256.0, mx / (iters * 64.0));
    }

    {
        const char* nm[] = {"fmac_dpp regs", "fmac plain regs", "fmac_dpp LDS operands dist1", "fmac_dpp LDS operands dist2", "fmac plain LDS operands dist2"};
        for (int mode = 0; mode < 5; mode++) for (int act = 16; act <= 64; act += 48) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k_block<0>, dim3(256), dim3(64), 98304, 0, out, cyc, 200, act);
                if (mode == 1) hipLaunchKernelGGL(k_block<1>, dim3(256), dim3(64), 98304, 0, out, cyc, 200, act);
                if (mode == 2) hipLaunchKernelGGL(k_block<2>, dim3(256), dim3(64), 98304, 0, out, cyc, 200, act);
                if (mode == 3) hipLaunchKernelGGL(k_block<3>, dim3(256), dim3(64), 98304, 0, out, cyc, 200, act);
                if (mode == 4) hipLaunchKernelGGL(k_block<4>, dim3(256), dim3(64), 98304, 0, out, cyc, 200, act);
                CHK(hipDeviceSynchronize());
            }
            long long h[256]; CHK(hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost));
            double mx = 0; for (int i = 0; i < 256; i++) if (h[i] > mx) mx = (double)h[i];
            printf("%-34s active lanes=%2d: %.2f ticks per chain step\n", nm[mode], act, mx / (200.0 * 256));
        }
    }
    // clock64 tick rate vs wall: report both
    {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dep_fma, dim3(256), dim3(256), 0, 0, out, cyc, 20000, 1.0001f, 0.5f); CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize()); float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        long long h; CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("dep_fmac 256x256: %.3f ms wall for %d dependent ops -> %.3f ns/op ; clock64 ticks/ns = %.4f\n", ms, 20000 * 64, ms * 1e6 / (20000.0 * 64), (double)h / (ms * 1e6));
    }
    // exactness of dot2c vs fmaf
    {
        const int T = 4096, K2 = 2048;
        std::vector<uint32_t> hw((size_t)T * K2), hx(K2);
        uint64_t s = 88172645463325252ULL;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        auto rbf = [&]() { uint32_t e = 100 + (uint32_t)(rnd() % 40); uint32_t m = (uint32_t)(rnd() & 0x7f); uint32_t sg = (uint32_t)(rnd() & 1); return (sg << 15) | (e << 7) | m; };
        for (auto& v : hw) v = rbf() | (rbf() << 16);
        for (auto& v : hx) v = rbf() | (rbf() << 16);
        uint32_t *dw, *dx, *o1, *o2;
        CHK(hipMalloc(&dw, hw.size() * 4)); CHK(hipMalloc(&dx, hx.size() * 4)); CHK(hipMalloc(&o1, T * 4)); CHK(hipMalloc(&o2, T * 4));
        CHK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_exact, dim3(T / 256), dim3(256), 0, 0, dw, dx, K2, o1, o2);
        CHK(hipDeviceSynchronize());
        std::vector<uint32_t> a(T), d(T);
        CHK(hipMemcpy(a.data(), o1, T * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(d.data(), o2, T * 4, hipMemcpyDeviceToHost));
        int diff = 0; for (int i = 0; i < T; i++) diff += a[i] != d[i];
        printf("dot2c(zero partner) vs fmaf chain: %d / %d chains differ (K=%d, exponents 2^-27..2^12)\n", diff, T, 2 * K2);
    }
    return 0;
}
