// mall_prefetch_bench.hip -- does a weight matrix that was touched shortly before stream faster out of the 256 MiB Infinity Cache than out of HBM?
// (the exact decode's chain-bound kernels -- wq|wk|wv 24 us, wo 15 us, w2 44 us -- leave the HBM 2/3 idle; gate|up then waits 35 us for its 235 MB.)
// Sequence per measurement: flush (stream 1 GiB of other memory), prefetch the first X MB of W with the given load policy, then TIME a full
// streaming read of W (235 MB) the way the decode kernels read it (16 B per lane, nt or default policy).
// build: hipcc --offload-arch=gfx950 -O3 tools/mall_prefetch_bench.hip -o tools/mall_prefetch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int POLICY> __device__ inline u32x4 ld16(const void* p) {
    u32x4 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// every workgroup streams its contiguous slice, 8 loads of 16 B per lane in flight
template <int POLICY> __global__ __launch_bounds__(256) void stream_kernel(const char* w, size_t bytes, unsigned* sink) {
    const size_t per = bytes / gridDim.x;
    const char* base = w + (size_t)blockIdx.x * per + (size_t)threadIdx.x * 16;
    unsigned acc = 0;
    for (size_t off = 0; off + 8 * 4096 <= per; off += 8 * 4096) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ld16<POLICY>(base + off + (size_t)i * 4096);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) acc ^= v[i][0] ^ v[i][3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// a light prefetcher: `waves` waves per CU touch one 16-B piece per 128-B line (line fill is what matters), default policy
template <int POLICY> __global__ __launch_bounds__(64) void touch_kernel(const char* w, size_t bytes, unsigned* sink) {
    const size_t per = bytes / gridDim.x;
    const char* base = w + (size_t)blockIdx.x * per + (size_t)threadIdx.x * 128;
    unsigned acc = 0;
    for (size_t off = 0; off + 8 * 8192 <= per; off += 8 * 8192) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ld16<POLICY>(base + off + (size_t)i * 8192);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) acc ^= v[i][0];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int P> static float timed_stream(const char* w, size_t bytes, unsigned* sink, hipEvent_t e0, hipEvent_t e1) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream_kernel<P>), dim3(1024), dim3(256), 0, 0, w, bytes, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const size_t WB = (size_t)235 << 20, JB = (size_t)1 << 30;
    char *w, *junk; unsigned* sink;
    (void)hipMalloc((void**)&w, WB); (void)hipMalloc((void**)&junk, JB); (void)hipMalloc((void**)&sink, 64);
    (void)hipMemset(w, 1, WB); (void)hipMemset(junk, 2, JB);
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("alloc: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    hipLaunchKernelGGL((stream_kernel<0>), dim3(1024), dim3(256), 0, 0, junk, JB, sink); printf("stream default: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    hipLaunchKernelGGL((stream_kernel<1>), dim3(1024), dim3(256), 0, 0, w, WB, sink); printf("stream nt: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    hipLaunchKernelGGL((touch_kernel<0>), dim3(2048), dim3(64), 0, 0, w, WB, sink); printf("touch default: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    hipLaunchKernelGGL((touch_kernel<2>), dim3(2048), dim3(64), 0, 0, w, WB, sink); printf("touch sc1: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    const char* pol[4] = {"default", "nt", "sc1", "sc0 sc1"};
    auto flush = [&]() { hipLaunchKernelGGL((stream_kernel<0>), dim3(1024), dim3(256), 0, 0, junk, JB, sink); };
    for (int rp = 0; rp < 2; rp++)                          // read policy of the timed stream: default, nt
        for (int pp = -1; pp < 4; pp++) {                   // prefetch policy (-1: none)
            if (pp == 2 && rp == 1) continue;
            for (int xmb : {64, 128, 200, 235}) {
                if (pp < 0 && xmb != 64) continue;
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    flush();
                    const size_t xb = (size_t)xmb << 20;
                    if (pp == 0) hipLaunchKernelGGL((touch_kernel<0>), dim3(2048), dim3(64), 0, 0, w, xb, sink);
                    if (pp == 1) hipLaunchKernelGGL((touch_kernel<1>), dim3(2048), dim3(64), 0, 0, w, xb, sink);
                    if (pp == 2) hipLaunchKernelGGL((touch_kernel<2>), dim3(2048), dim3(64), 0, 0, w, xb, sink);
                    if (pp == 3) hipLaunchKernelGGL((stream_kernel<0>), dim3(1024), dim3(256), 0, 0, w, xb, sink);      // full-width default read as the prefetch
                    const float ms = rp == 0 ? timed_stream<0>(w, WB, sink, e0, e1) : timed_stream<1>(w, WB, sink, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("timed read %-7s | prefetch %-22s %3d MB: %7.1f us = %5.2f TB/s\n", pol[rp], pp < 0 ? "none (cold)" : pp == 3 ? "full-width default read" : pol[pp], pp < 0 ? 0 : xmb,
                       1e3 * best, (double)WB / (best * 1e-3) / 1e12);
            }
        }
    // how long does the light prefetcher itself take, alone (it would run beside a chain-bound kernel)?
    for (int xmb : {64, 128, 235}) {
        flush();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((touch_kernel<0>), dim3(2048), dim3(64), 0, 0, w, (size_t)xmb << 20, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("touch_kernel alone, %3d MB (one 16-B load per 128-B line, 2048 waves): %.1f us = %.2f TB/s of line fills\n", xmb, 1e3 * ms, (double)((size_t)xmb << 20) / (ms * 1e-3) / 1e12);
    }
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
