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

int main() {
    float* out; long long* cyc; uint32_t* w;
    CHK(hipMalloc(&out, 1 << 20)); CHK(hipMalloc(&cyc, 4096 * 8)); CHK(hipMalloc(&w, 1 << 16));
    CHK(hipMemset(w, 0x3f, 1 << 16));
    const int iters = 2000;
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
