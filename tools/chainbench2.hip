// chainbench2.hip -- dependent-chain cost of v_pk_add_f32 and of v_add_f32 with a DPP row broadcast operand (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k_pk(float* out, long long* ticks, int iters, float a, float b) {
    f32x2 acc = {a, b}, p0 = {b, a}, p1 = {a * 0.5f, b * 0.25f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" "\n\t" "v_pk_add_f32 %0, %0, %1" "\n\t" "v_pk_add_f32 %0, %0, %2" : "+v"(acc) : "v"(p0), "v"(p1));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = acc.x + acc.y;
}
__global__ void k_dpp(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" "\n\t" "v_add_f32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p0), "v"(p1));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = acc;
}
__global__ void k_plain(float* out, long long* ticks, int iters, float a, float b) {
    float acc = a, p0 = b + threadIdx.x, p1 = a * 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" "\n\t" "v_add_f32 %0, %1, %0" "\n\t" "v_add_f32 %0, %2, %0" : "+v"(acc) : "v"(p0), "v"(p1));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = acc;
}
template <typename F> static void timeit(const char* name, F kern) {
    float* out; long long* ticks;
    (void)hipMalloc((void**)&out, 1 << 20); (void)hipMalloc((void**)&ticks, 64);
    const int iters = 32768;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64), 100 * 1024, 0, out, ticks, iters, 1.0f, 1e-3f);   // 100 KiB LDS: one wave per CU
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-28s: %.3f ns/op (= %.2f cycles at 2.4 GHz), %.2f s_memtime ticks/op\n", name, ms * 1e6 / (iters * 32.0), ms * 1e6 / (iters * 32.0) * 2.4, (double)h / (iters * 32.0));
}
int main() {
    timeit("v_add_f32 dependent", k_plain);
    timeit("v_pk_add_f32 dependent", k_pk);
    timeit("v_add_f32_dpp newbcast", k_dpp);
    timeit("v_add_f32 dependent (again)", k_plain);
    return 0;
}
