// mfma_x1_probe.hip -- v_mfma_f32_16x16x1_f32 (FOUR 16x16 blocks, one k per instruction) with the B operand broadcast from one 16-lane row (BLGP 4..7):
// a matrix-core feed for the CHAIN-tiled weight layouts that needs NO transpose.  gemm_stream_kernel's SRC 2 (lnb_batch_kernels.h) loads, per lane, eight
// consecutive k of one weight row and spends 16+ cross-row ops per 8 k-groups turning them into 16x16x4 A operands (k = 4g + lane row).  With the four
// blocks of 16x16x1 as four WEIGHT TILES -- lane (i, b) = row i of tile b, every lane holding the SAME eight k -- the A operand of step k is one unpack op,
// and the B operand of steps 4g..4g+3 is ONE LDS dword per lane (lane (n, q) = activation row n at k = 4g + q), broadcast by BLGP = 4 + q.
// This probe answers what a rewrite would stand on:
//   1. exactness: is acc = mfma_16x16x1(a_k, b_k, acc) over k ascending bit-identical to the reference's chain acc = acc + a_k * b_k (f32, RNE)?
//   2. BLGP 4..7 really broadcast row q of the B register to all four blocks (and the D layout per block is the 16x16x4 one)?
//   3. rate: cycles per instruction with 1 / 2 / 4 / 8 independent accumulator sets per wave (each set a dependent chain), one wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_x1_probe.hip -o tools/mfma_x1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s failed: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a[(b * 16 + i) * K + k]: weight tile b, row i;  x[n * K + k]: activation row n.  out[((b * 16 + row) * 16 + n)]
__global__ __launch_bounds__(64) void exact_kernel(const float* a, const float* x, float* out, int K) {
    const int lane = threadIdx.x, i = lane & 15, b = lane >> 4;
    f32x16 acc = {0};
    for (int g = 0; g < K / 4; g++) {
        const float bv = x[i * K + 4 * g + b];               // lane (n = i, q = b): activation row n at k = 4g + q
        const float a0 = a[(b * 16 + i) * K + 4 * g + 0], a1 = a[(b * 16 + i) * K + 4 * g + 1], a2 = a[(b * 16 + i) * K + 4 * g + 2], a3 = a[(b * 16 + i) * K + 4 * g + 3];
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, bv, acc, 0, 0, 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, bv, acc, 0, 0, 5);
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a2, bv, acc, 0, 0, 6);
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a3, bv, acc, 0, 0, 7);
    }
    // D layout (assumed; the comparison tells): register 4 * blk + r of lane (n, q) = block blk, row 4 q + r, column n
    for (int blk = 0; blk < 4; blk++)
        for (int r = 0; r < 4; r++) out[((blk * 16 + 4 * b + r) * 16) + i] = acc[4 * blk + r];
}

// The other way round: ONE weight tile, FOUR batch tiles (blocks = batch tiles; CBSZ = 2 broadcasts the A operand of block ABID to all four).  Lane (i, r) holds
// eight consecutive k of weight row i -- k = 32 j + 8 r + e, exactly what gemm_stream_kernel's chain-layout loads put there -- and ABID = r picks whose
// element e is the step's A operand: k ascending = j, then r, then e.  Lane (n, b) of the B operand = batch tile b, row n at that k.
// a[i * K + k]: weight row i;  x[(b * 16 + n) * K + k];  out[(row * 64) + b * 16 + n]
template <int R> __device__ __forceinline__ f32x16 step8(f32x16 acc, const float (&av)[8], const float* xr, int k0) {
#pragma unroll
    for (int e = 0; e < 8; e++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av[e], xr[k0 + 8 * R + e], acc, 2, R, 0);
    return acc;
}
__global__ __launch_bounds__(64) void exact_cbsz_kernel(const float* a, const float* x, float* out, int K) {
    const int lane = threadIdx.x, i = lane & 15, r = lane >> 4;
    f32x16 acc = {0};
    const float* xr = x + (size_t)lane * K;                  // lane (n, b) = row 16 b + n of the 64 activation rows
    for (int j = 0; j < K / 32; j++) {
        float av[8];
        for (int e = 0; e < 8; e++) av[e] = a[i * K + 32 * j + 8 * r + e];
        acc = step8<0>(acc, av, xr, 32 * j); acc = step8<1>(acc, av, xr, 32 * j); acc = step8<2>(acc, av, xr, 32 * j); acc = step8<3>(acc, av, xr, 32 * j);
    }
    for (int blk = 0; blk < 4; blk++)
        for (int q = 0; q < 4; q++) out[(4 * r + q) * 64 + blk * 16 + i] = acc[4 * blk + q];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* sink, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; q++) acc[q] = f32x16{0};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 4);
#pragma unroll
        for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 5);
#pragma unroll
        for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 6);
#pragma unroll
        for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 7);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < NACC; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// the instruction the library uses today, same harness: v_mfma_f32_16x16x4_f32 (one block, four k)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void rate4_kernel(float* sink, long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; q++) acc[q] = f32x4{0};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < NACC; q++) for (int r = 0; r < 4; r++) s += acc[q][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static float bf16_val(uint32_t& s, float scale) {           // a bf16-representable value (what the path multiplies): random sign / exponent spread / 8-bit mantissa
    s = s * 1664525u + 1013904223u;
    const uint32_t m = (s >> 9) & 0x7f, e = 120 + ((s >> 20) % 12), sg = (s >> 31);
    uint32_t bits = (sg << 31) | (e << 23) | (m << 16);
    float f; memcpy(&f, &bits, 4);
    return f * scale;
}
template <int NACC> static void rate(const char* what, bool x4, float* sink, long long* cyc) {
    const int iters = 2000, NWG = 256;
    for (int rep = 0; rep < 2; rep++) {
        if (x4) hipLaunchKernelGGL(rate4_kernel<NACC>, dim3(NWG), dim3(256), 0, 0, sink, cyc, iters);
        else hipLaunchKernelGGL(rate_kernel<NACC>, dim3(NWG), dim3(256), 0, 0, sink, cyc, iters);
        CHK(hipDeviceSynchronize());
    }
    std::vector<long long> h(NWG * 4);
    CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (long long v : h) avg += (double)v; avg /= h.size();
    // clock64 = s_memtime (constant 100 MHz on gfx9) -- report it per instruction in ns and in shader cycles at 2.25 GHz is left to the reader; MACs per instruction 1024 either way
    const double per = avg / ((double)iters * 4 * NACC);
    printf("  %-34s %d accumulator set(s) per wave, 4 waves per CU on 256 CUs: %.3f clock64 ticks per instruction\n", what, NACC, per);
}
int main() {
    const int K = 4096;
    std::vector<float> a(64 * K), x(16 * K), ref(64 * 16), got(64 * 16);
    uint32_t s = 12345;
    for (auto& v : a) v = bf16_val(s, 0.015625f);             // (a power of two: the value stays bf16-representable, so every product is exact in f32)
    for (auto& v : x) v = bf16_val(s, 1.0f);
    for (int r = 0; r < 64; r++)
        for (int n = 0; n < 16; n++) {
            float acc = 0.f;
            for (int k = 0; k < K; k++) { volatile float p = a[r * K + k] * x[n * K + k]; acc = acc + p; }      // operations_lineartransform.go:46-65 (product exact in f32)
            ref[r * 16 + n] = acc;
        }
    float *da, *dx, *dout, *sink; long long* cyc;
    CHK(hipMalloc((void**)&da, a.size() * 4)); CHK(hipMalloc((void**)&dx, x.size() * 4)); CHK(hipMalloc((void**)&dout, got.size() * 4));
    CHK(hipMalloc((void**)&sink, 256 * 256 * 4)); CHK(hipMalloc((void**)&cyc, 256 * 4 * 8));
    CHK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, da, dx, dout, K);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, first = -1;
    for (int q = 0; q < 64 * 16; q++) if (memcmp(&got[q], &ref[q], 4) != 0) { if (first < 0) first = q; bad++; }
    printf("exactness: 64 weight rows (4 blocks) x 16 activation rows, K = %d, v_mfma_f32_16x16x1_f32 with BLGP 4..7 against the sequential f32 chain: %d of 1024 outputs differ", K, bad);
    if (bad) printf(" (first: output %d, device %.9g, chain %.9g)", first, got[first], ref[first]);
    printf("\n");
    {   // one weight tile x four batch tiles (CBSZ = 2, ABID = lane row): 16 weight rows, 64 activation rows
        std::vector<float> x4(64 * K), ref4(16 * 64), got4(16 * 64);
        for (auto& v : x4) v = bf16_val(s, 1.0f);
        for (int r = 0; r < 16; r++)
            for (int n = 0; n < 64; n++) {
                float acc = 0.f;
                for (int k = 0; k < K; k++) { volatile float p = a[r * K + k] * x4[n * K + k]; acc = acc + p; }
                ref4[r * 64 + n] = acc;
            }
        float *dx4, *dout4;
        CHK(hipMalloc((void**)&dx4, x4.size() * 4)); CHK(hipMalloc((void**)&dout4, got4.size() * 4));
        CHK(hipMemcpy(dx4, x4.data(), x4.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(exact_cbsz_kernel, dim3(1), dim3(64), 0, 0, da, dx4, dout4, K);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(got4.data(), dout4, got4.size() * 4, hipMemcpyDeviceToHost));
        int bad4 = 0, first4 = -1;
        for (int q = 0; q < 16 * 64; q++) if (memcmp(&got4[q], &ref4[q], 4) != 0) { if (first4 < 0) first4 = q; bad4++; }
        printf("exactness: 16 weight rows x 64 activation rows (4 blocks = 4 batch tiles), CBSZ = 2 with ABID = the lane row that holds the step's k: %d of 1024 outputs differ", bad4);
        if (bad4) printf(" (first: output %d, device %.9g, chain %.9g)", first4, got4[first4], ref4[first4]);
        printf("\n");
    }
    printf("rate (clock64 ticks; both instructions are 1024 multiply-accumulates):\n");
    rate<1>("v_mfma_f32_16x16x1_f32 blgp", false, sink, cyc); rate<2>("v_mfma_f32_16x16x1_f32 blgp", false, sink, cyc);
    rate<4>("v_mfma_f32_16x16x1_f32 blgp", false, sink, cyc); rate<8>("v_mfma_f32_16x16x1_f32 blgp", false, sink, cyc);
    rate<1>("v_mfma_f32_16x16x4_f32", true, sink, cyc); rate<2>("v_mfma_f32_16x16x4_f32", true, sink, cyc);
    rate<4>("v_mfma_f32_16x16x4_f32", true, sink, cyc); rate<8>("v_mfma_f32_16x16x4_f32", true, sink, cyc);
    return 0;
}
