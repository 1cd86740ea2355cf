// chainbench.hip -- what one "chain wave" step costs on gfx950: 16 dependent v_add_f32 per 4 ds_read_b128 (the GEMV inner loop),
// under different read shapes and with/without producer waves hammering the LDS.  Prints s_memtime ticks per k-step.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chainbench.hip -o tools/chainbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DEVINL __device__ __forceinline__
DEVINL float add4(float acc, const float4& p) { acc += p.x; acc += p.y; acc += p.z; acc += p.w; return acc; }
DEVINL void touch16(const float4& a, const float4& b, const float4& c, const float4& d) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w),
                 "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
}
// mode: 0 adds only | 1 b128 reads, all lanes distinct rows (RW=64) | 2 b128 reads, 16 distinct rows replicated x4 (RW=16)
//       3 b128 reads issued by lanes 0..15 only (exec masked), adds by all | 4 like 2 but reads are 8 x ds_read_b64
//       5 like 1 with reads only every other group (half the LDS instructions)
// nprod producer waves write float4s into a disjoint LDS region in a loop while the chain runs.
template <int MODE> __global__ __launch_bounds__(512) void k_chain(float* out, long long* ticks, int groups, int nprod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* ring = (float*)smem;                         // 16 KB: [j][row][4] like the product ring
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ring[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    if (wave != 0) {
        if (wave <= nprod) {
            float4* dst = (float4*)(smem + 16384 + (wave - 1) * 8192);
            float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
            for (int g = 0; g < groups; g++) {
#pragma unroll
                for (int u = 0; u < 8; u++) dst[u * 64 + lane] = v;
                v.x += 1.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    const int RW = (MODE == 2 || MODE == 3 || MODE == 4) ? 16 : 64;
    const int row = lane & (RW - 1);
    const char* src = smem + row * 16;
    float acc = 0.0f;
    float4 pb[2][4];
    auto load = [&](float4 (&d)[4], int g) {
        const char* s = src + (g & 3) * 4 * RW * 16;
        if (MODE == 0) return;
        if (MODE == 3) { if (lane < 16) { for (int j = 0; j < 4; j++) d[j] = *(const float4*)(s + j * RW * 16); } return; }
        if (MODE == 4) { for (int j = 0; j < 4; j++) { const float2 a = *(const float2*)(s + j * RW * 16), b = *(const float2*)(s + j * RW * 16 + 8); d[j] = make_float4(a.x, a.y, b.x, b.y); } return; }
        if (MODE == 5 && (g & 1)) return;
        for (int j = 0; j < 4; j++) d[j] = *(const float4*)(s + j * RW * 16);
    };
    for (int j = 0; j < 4; j++) { pb[0][j] = make_float4(1e-3f, 2e-3f, 0.f, 1e-3f); pb[1][j] = pb[0][j]; }
    load(pb[0], 0);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; g += 2) {
        load(pb[1], g + 1);
        __builtin_amdgcn_sched_barrier(0);
        touch16(pb[0][0], pb[0][1], pb[0][2], pb[0][3]);
        __builtin_amdgcn_sched_barrier(0);
        acc = add4(acc, pb[0][0]); acc = add4(acc, pb[0][1]); acc = add4(acc, pb[0][2]); acc = add4(acc, pb[0][3]);
        __builtin_amdgcn_sched_barrier(0);
        load(pb[0], g + 2);
        __builtin_amdgcn_sched_barrier(0);
        touch16(pb[1][0], pb[1][1], pb[1][2], pb[1][3]);
        __builtin_amdgcn_sched_barrier(0);
        acc = add4(acc, pb[1][0]); acc = add4(acc, pb[1][1]); acc = add4(acc, pb[1][2]); acc = add4(acc, pb[1][3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { ticks[blockIdx.x] = t1 - t0; }
    out[blockIdx.x * 64 + lane] = acc;
}

// 32-step groups: 8 x ds_read_b128 in flight while 32 adds of the previous group run (prefetch distance = 32 adds)
__global__ __launch_bounds__(512) void k_chain32(float* out, long long* ticks, int groups, int nprod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* ring = (float*)smem;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ring[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    if (wave != 0) {
        if (wave <= nprod) {
            float4* dst = (float4*)(smem + 32768 + (wave - 1) * 8192);
            float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
            for (int g = 0; g < groups; g++) {
#pragma unroll
                for (int u = 0; u < 2; u++) dst[u * 64 + lane] = v;     // realistic rate: 2 x b128 writes per 16 k-steps per helper
                v.x += 1.0f;
                for (int w = 0; w < 8; w++) v.y = v.y * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    const int RW = 16, row = lane & (RW - 1);
    const char* src = smem + row * 16;
    float acc = 0.0f;
    float4 pb[2][8];
    auto load = [&](float4 (&d)[8], int g) {
        const char* s = src + (g & 3) * 8 * RW * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) d[j] = *(const float4*)(s + j * RW * 16);
    };
    load(pb[0], 0);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups / 2; g += 2) {
        load(pb[1], g + 1);
        __builtin_amdgcn_sched_barrier(0);
        touch16(pb[0][0], pb[0][1], pb[0][2], pb[0][3]); touch16(pb[0][4], pb[0][5], pb[0][6], pb[0][7]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j++) acc = add4(acc, pb[0][j]);
        __builtin_amdgcn_sched_barrier(0);
        load(pb[0], g + 2);
        __builtin_amdgcn_sched_barrier(0);
        touch16(pb[1][0], pb[1][1], pb[1][2], pb[1][3]); touch16(pb[1][4], pb[1][5], pb[1][6], pb[1][7]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j++) acc = add4(acc, pb[1][j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { ticks[blockIdx.x] = t1 - t0; }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE> static void run(const char* name, int nprod, int nblocks) {
    float* out; long long* ticks;
    hipMalloc((void**)&out, nblocks * 64 * 4); hipMalloc((void**)&ticks, nblocks * 8);
    const int groups = 4096;                             // 65536 k-steps
    hipFuncSetAttribute((const void*)k_chain<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_chain<MODE>, dim3(nblocks), dim3(512), 96 * 1024, 0, out, ticks, groups, nprod);
    hipDeviceSynchronize();
    std::vector<long long> h(nblocks);
    hipMemcpy(h.data(), ticks, nblocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-58s producers %d  blocks %3d : %.2f ticks / k-step\n", name, nprod, nblocks, s / nblocks / (groups * 16.0));
    hipFree(out); hipFree(ticks);
}

static void run32(int nprod, int nblocks) {
    float* out; long long* ticks;
    hipMalloc((void**)&out, nblocks * 64 * 4); hipMalloc((void**)&ticks, nblocks * 8);
    const int groups = 4096;
    hipFuncSetAttribute((const void*)k_chain32, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_chain32, dim3(nblocks), dim3(512), 96 * 1024, 0, out, ticks, groups, nprod);
    hipDeviceSynchronize();
    std::vector<long long> h(nblocks);
    hipMemcpy(h.data(), ticks, nblocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-58s producers %d  blocks %3d : %.2f ticks / k-step\n", "8 x ds_read_b128 / 32 adds (32-step groups), mild producers", nprod, nblocks, s / nblocks / (groups / 2 * 32.0));
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int nprod : {0, 2, 6}) run32(nprod, 256);
    for (int nprod : {0, 2, 6}) {
        run<0>("adds only", nprod, 256);
        run<1>("4 x ds_read_b128 / 16 adds, 64 rows", nprod, 256);
        run<2>("4 x ds_read_b128 / 16 adds, 16 rows replicated", nprod, 256);
        run<3>("4 x ds_read_b128 by lanes 0..15 only", nprod, 256);
        run<4>("8 x ds_read_b64 / 16 adds, 16 rows replicated", nprod, 256);
        run<5>("2 x ds_read_b128 / 16 adds (half the reads)", nprod, 256);
    }
    return 0;
}
