// mfma_lds_bench.hip -- what does an LDS read cost the f32 matrix pipe?  One workgroup on one CU; every wave runs ITER blocks of
// 16 independent v_mfma_f32_16x16x4_f32 (8 accumulators x 2) with LDS reads placed in different ways; cycles per block from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_bench.hip -o tools/mfma_lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M2(i) "v_mfma_f32_16x16x4_f32 %" #i ", %13, %14, %" #i "\n"
#define SH_M2(i) "v_lshlrev_b32 %12, 16, %14\n s_nop 1\n v_mfma_f32_16x16x4_f32 %" #i ", %13, %12, %" #i "\n"
#define RD128A "ds_read_b128 %8, %15\n"
#define RD128B "ds_read_b128 %9, %15 offset:16\n"
#define RD128C "ds_read_b128 %10, %15 offset:32\n"
#define RD128D "ds_read_b128 %11, %15 offset:48\n"
#define RD32(o) "ds_read_b32 %12, %15 offset:" #o "\n"
#define WAIT0 "s_waitcnt lgkmcnt(0)\n"
#define ALL8 M2(0) M2(1) M2(2) M2(3) M2(4) M2(5) M2(6) M2(7)
#define SH_ALL8 SH_M2(0) SH_M2(1) SH_M2(2) SH_M2(3) SH_M2(4) SH_M2(5) SH_M2(6) SH_M2(7)

template <int MODE> __global__ __launch_bounds__(1024) void k(long long* out, int iters, float av, float bv) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0, r0 = c0, r1 = c0, r2 = c0, r3 = c0;
    float tmp = 0.f;
    const unsigned addr = (unsigned)(threadIdx.x & 63) * 80u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(tmp) : "v"(av), "v"(bv), "v"(addr)
        if (MODE == 0) asm volatile(ALL8 ALL8 OPS);
        if (MODE == 1) asm volatile(RD128A RD128B ALL8 ALL8 WAIT0 OPS);
        if (MODE == 2) asm volatile(RD128A RD128B WAIT0 ALL8 ALL8 OPS);
        if (MODE == 3) asm volatile(M2(0) M2(1) M2(2) M2(3) RD128A M2(4) M2(5) M2(6) M2(7) M2(0) M2(1) M2(2) M2(3) RD128B M2(4) M2(5) M2(6) M2(7) WAIT0 OPS);
        if (MODE == 4) asm volatile(RD32(0) M2(0) M2(1) RD32(4) M2(2) M2(3) RD32(8) M2(4) M2(5) RD32(12) M2(6) M2(7) RD32(16) M2(0) M2(1) RD32(20) M2(2) M2(3) RD32(24) M2(4) M2(5) RD32(28) M2(6) M2(7) WAIT0 OPS);
        if (MODE == 5) asm volatile(RD128A RD128B RD128C RD128D ALL8 ALL8 WAIT0 OPS);
        if (MODE == 6) asm volatile(SH_ALL8 SH_ALL8 OPS);
        if (MODE == 7) asm volatile(RD128A RD128B SH_ALL8 SH_ALL8 WAIT0 OPS);
        if (MODE == 8) asm volatile(RD128A RD128B "s_waitcnt lgkmcnt(1)\n" M2(0) M2(1) M2(2) M2(3) "s_waitcnt lgkmcnt(0)\n" M2(4) M2(5) M2(6) M2(7) ALL8 OPS);
        if (MODE == 9) asm volatile("s_nop 0\n" ALL8 "s_nop 0\n" ALL8 OPS);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (r0[0] + r1[1] + r2[2] + r3[3] + tmp + c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] == 12345.f) out[63] = 1;
}
template <int MODE> static void run(const char* what, long long* d, int threads) {
    const int iters = 2000;
    long long h[8];
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0f, 1e-3f);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d %-62s %d waves/SIMD: %7.1f cycles per 16 MFMAs (ideal %d)\n", MODE, what, threads / 256, (double)h[0] / iters, 512 * threads / 256);
}
template <int MODE> static void run_chip(const char* what, long long* d, int threads) {
    // whole chip: 1024 workgroups, event-timed -> TFLOP/s of 16x16x4 f32 MFMAs (2048 flop each per wave)
    const int iters = 4000, blocks = 1024;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0f, 1e-3f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-3f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chip mode %d %-40s %4d threads/WG x %d WGs: %.2f ms, %.1f TFLOP/s\n", MODE, what, threads, blocks, ms,
           (double)blocks * (threads / 64) * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    long long* d; (void)hipMalloc((void**)&d, 64 * 8);
    for (int threads = 256; threads <= 1024; threads += 256) run_chip<0>("bare MFMAs", d, threads);
    run_chip<6>("shift + nop per MFMA", d, 512);
    run_chip<1>("2 ds_read_b128 per 16", d, 512);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0>("bare MFMAs", d, threads);
        run<9>("one s_nop 0 per 8 MFMAs", d, threads);
        run<6>("v_lshlrev + s_nop 1 in front of every MFMA", d, threads);
        run<1>("2 x ds_read_b128 at the head, wait at the tail", d, threads);
        run<2>("2 x ds_read_b128 at the head, wait lgkmcnt(0) at once", d, threads);
        run<8>("2 x ds_read_b128 at the head, staggered waits", d, threads);
        run<3>("ds_read_b128 after MFMA 4 and after MFMA 12", d, threads);
        run<4>("8 x ds_read_b32, one per 2 MFMAs", d, threads);
        run<5>("4 x ds_read_b128 at the head", d, threads);
        run<7>("2 x ds_read_b128 at the head + shifts/nops", d, threads);
    }
    return 0;
}
