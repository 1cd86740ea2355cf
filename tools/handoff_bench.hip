// handoff_bench.hip -- what would ONE launch for two dependent decode kernels (wo -> ffn_norm + w1|w3) buy on this chip?  (VERDICT r4 #2e)
//
// Every decode launch needs ALL outputs of its predecessor (the next product's input vector is the previous product's 4096 outputs, 16 per CU),
// so a fused launch replaces the kernel boundary by a grid-wide hand-off: every workgroup publishes its rows (write-through stores), arrives on
// a counter, waits until all 256 have arrived, acquires, gathers the 8 KB vector.  This tool measures exactly that trade on the decode kernels'
// launch shape -- 256 workgroups of 512 threads, one per CU (100 KB of LDS requested), a phase = a dependent f32 add chain of `steps` steps per
// wave (the chain-bound kernels' main loop) -- in three forms, per PAIR of phases:
//   two launches     : phase A | kernel boundary | phase B                       (what the library does: captured graph, back-to-back launches)
//   fused, flat      : phase A | release + arrive on ONE counter | poll | acquire + gather | phase B
//   fused, per XCD   : ... arrive on the workgroup's XCD counter (id % 8), the last arriver of an XCD arrives on the global one; poll the global
// and reports pair time - 2 x (a phase alone, measured inside the kernel on the wall clock) = what the boundary / the hand-off costs.
// build: hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s failed: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float chain(float acc, float p, int steps) {            // `steps` dependent adds (one wave-instruction each)
    for (int i = 0; i < steps; i += 8) {
        asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\t"
                     "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0" : "+v"(acc) : "v"(p));
    }
    return acc;
}
// a phase: gather the 8 KB input vector (16 B per thread), run the chain, publish this workgroup's 16 outputs (32 B); write-through when `wt`
__device__ __forceinline__ void phase(const uint4* in, uint16_t* out, int steps, bool wt, bool sc1_in) {
    extern __shared__ char smem[];
    uint4 v;
    if (sc1_in) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(in + threadIdx.x) : "memory");
    else v = in[threadIdx.x];
    ((uint4*)smem)[threadIdx.x] = v;
    __syncthreads();
    float acc = chain(__uint_as_float(v.x & 0x3f800000u), 1.0f, steps);
    if ((threadIdx.x & 31) == 0) {
        const uint16_t r = (uint16_t)(__float_as_uint(acc) >> 16);
        uint16_t* o = out + blockIdx.x * 16 + (threadIdx.x >> 5);
        if (wt) asm volatile("global_store_short %0, %1, off sc0 sc1" :: "v"(o), "v"((unsigned)r) : "memory"); else *o = r;
    }
}
__global__ __launch_bounds__(512) void phase_kernel(const uint4* in, uint16_t* out, int steps) { phase(in, out, steps, false, false); }

// the in-kernel time of ONE phase on the wall clock (for the subtraction): max over workgroups is taken on the host
__global__ __launch_bounds__(512) void phase_timed_kernel(const uint4* in, uint16_t* out, int steps, long long* t) {
    const long long t0 = wall_clock64();
    phase(in, out, steps, false, false);
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
}
template <int XCD>
__global__ __launch_bounds__(512) void fused_kernel(const uint4* in, uint16_t* mid, uint16_t* out, int steps, unsigned* counters, unsigned epoch) {
    phase(in, mid, steps, true, false);
    // ---- hand-off: all stores of this workgroup have left (vmcnt counts stores on gfx9), one lane releases and arrives
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = gridDim.x;
        if (XCD) {                                           // workgroup w runs on XCD w % 8: arrive there, the XCD's last arriver tells the global counter
            const unsigned x = blockIdx.x & 7u, per = (n + 7u - x) / 8u;
            const unsigned got = __hip_atomic_fetch_add(counters + 16 * (1 + x), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (got == epoch * per) __hip_atomic_fetch_add(counters, per, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * n) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    phase((const uint4*)mid, out, steps, false, true);       // gathers the vector the other workgroups just published (sc1 loads: served by the L2 / fabric, never a stale L1 line)
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2048, iters = 400, NWG = 256;
    const size_t lds = 100 * 1024;
    uint4* in; uint16_t *mid, *out; unsigned* cnt; long long* tt;
    CHK(hipMalloc((void**)&in, 8192)); CHK(hipMalloc((void**)&mid, 8192)); CHK(hipMalloc((void**)&out, 8192)); CHK(hipMalloc((void**)&cnt, 16 * 9 * 4)); CHK(hipMalloc((void**)&tt, NWG * 8));
    CHK(hipMemset(in, 0x3f, 8192)); CHK(hipMemset(mid, 0, 8192));
    CHK(hipFuncSetAttribute((const void*)phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)phase_timed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int khz = 0; CHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    // a phase alone, in-kernel
    double phase_us = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(phase_timed_kernel, dim3(NWG), dim3(512), lds, st, in, mid, steps, tt);
        CHK(hipStreamSynchronize(st));
        std::vector<long long> h(NWG); CHK(hipMemcpy(h.data(), tt, NWG * 8, hipMemcpyDeviceToHost));
        long long mx = 0; for (long long v : h) mx = v > mx ? v : mx;
        phase_us = (double)mx / khz * 1e3;
    }
    auto timed = [&](int form) -> double {                   // microseconds per PAIR of phases, as a captured graph of `iters` pairs
        CHK(hipMemsetAsync(cnt, 0, 16 * 9 * 4, st));
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; i++) {
            if (form == 0) { hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(512), lds, st, in, mid, steps); hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(512), lds, st, (const uint4*)mid, out, steps); }
            else if (form == 1) hipLaunchKernelGGL(fused_kernel<0>, dim3(NWG), dim3(512), lds, st, in, mid, out, steps, cnt, (unsigned)(i + 1));
            else hipLaunchKernelGGL(fused_kernel<1>, dim3(NWG), dim3(512), lds, st, in, mid, out, steps, cnt, (unsigned)(i + 1));
        }
        CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipMemsetAsync(cnt, 0, 16 * 9 * 4, st));
            CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms * 1e3 / iters < best ? ms * 1e3 / iters : best;
        }
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        return best;
    };
    const double two = timed(0), flat = timed(1), xcd = timed(2);
    printf("phase = %d dependent adds per wave, 256 workgroups x 512 threads, one per CU: a phase alone takes %.2f us in the kernel (slowest workgroup, wall clock)\n", steps, phase_us);
    printf("per PAIR of dependent phases (graph of %d pairs):\n", iters);
    printf("  two launches                          %7.2f us   = 2 phases + %.2f us   (two kernel boundaries per pair in a chain of launches: %.2f us each)\n", two, two - 2 * phase_us, (two - 2 * phase_us) / 2);
    printf("  one launch, flat counter hand-off     %7.2f us   = 2 phases + %.2f us   (one boundary + one in-kernel hand-off: the hand-off costs %.2f us)\n", flat, flat - 2 * phase_us, flat - 2 * phase_us - (two - 2 * phase_us) / 2);
    printf("  one launch, per-XCD counter hand-off  %7.2f us   = 2 phases + %.2f us   (the hand-off costs %.2f us)\n", xcd, xcd - 2 * phase_us, xcd - 2 * phase_us - (two - 2 * phase_us) / 2);
    printf("  fusing a pair of launches changes the pair by %+.2f us (flat) / %+.2f us (per XCD)\n", flat - two, xcd - two);
    return 0;
}
