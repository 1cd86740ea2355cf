// chainbench4.hip -- round 4: what a single wave pays for scalar work and for VALU <-> SALU hand-offs (the RMSNorm walker's item loop)
// build: hipcc --offload-arch=gfx950 -O3 tools/chainbench4.hip -o tools/chainbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define KERNEL(name, setup, body, fin)                                                        \
__global__ void name(unsigned* out, long long* ticks, int iters, unsigned a) {                \
    unsigned v = a + threadIdx.x * 0, w = a * 3u; unsigned s = a, s2 = 5; setup               \
    const long long t0 = __builtin_amdgcn_s_memtime();                                        \
    for (int i = 0; i < iters; i++) { body }                                                  \
    const long long t1 = __builtin_amdgcn_s_memtime();                                        \
    fin                                                                                       \
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                        \
    out[threadIdx.x] = v + w + s + s2;                                                        \
}
// 16 dependent SALU adds
KERNEL(k_salu, , asm volatile(REP16("s_add_u32 %0, %0, %1\n\t") : "+s"(s) : "s"(s2) : "scc");, )
// 16 dependent VALU adds (integer)
KERNEL(k_valu, , asm volatile(REP16("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(w));, )
// 16 x (VALU -> SGPR -> VALU): v_readfirstlane then v_add with the SGPR
KERNEL(k_rfl, , asm volatile(REP16("v_readfirstlane_b32 %1, %0\n\tv_add_u32 %0, %1, %2\n\t") : "+v"(v), "+s"(s) : "v"(w));, )
// 16 x (VALU -> SGPR -> SALU -> VALU): readfirstlane, s_add, v_mov
KERNEL(k_rfl_salu, , asm volatile(REP16("v_readfirstlane_b32 %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_mov_b32 %0, %1\n\t") : "+v"(v), "+s"(s) :: "scc");, )
// 16 x (SALU lane select -> v_readlane -> SALU use): s_and lane, v_readlane, s_add
KERNEL(k_rl_dyn, , asm volatile(REP16("s_and_b32 %1, %2, 63\n\ts_nop 3\n\tv_readlane_b32 %2, %0, %1\n\ts_add_u32 %2, %2, 1\n\t") : "+v"(v), "+s"(s), "+s"(s2) :: "scc");, )
// 16 x v_readlane with a CONSTANT lane feeding a dependent VALU op
KERNEL(k_rl_const, , asm volatile(REP16("v_readlane_b32 %1, %0, 5\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(s));, )
// 16 x (VALU compare -> vcc -> s_cbranch not taken)
KERNEL(k_vcc_branch, , asm volatile(REP16("v_cmp_eq_u32 vcc, 0x7fffffff, %0\n\ts_cbranch_vccnz L_never_%=\n\tv_add_u32 %0, 1, %0\n\t") "s_branch L_end_%=\nL_never_%=:\n\tv_add_u32 %0, 7, %0\nL_end_%=:\n\t" : "+v"(v) :: "vcc");, )
// 16 x (readfirstlane -> s_cmp -> s_cbranch not taken)
KERNEL(k_rfl_branch, , asm volatile(REP16("v_readfirstlane_b32 %1, %0\n\ts_cmp_eq_u32 %1, 0x7fffffff\n\ts_cbranch_scc1 L_never_%=\n\tv_add_u32 %0, 1, %0\n\t") "s_branch L_end_%=\nL_never_%=:\n\tv_add_u32 %0, 7, %0\nL_end_%=:\n\t" : "+v"(v), "+s"(s) :: "scc");, )
// 16 taken branches (each to the next line)
KERNEL(k_taken, , asm volatile(REP16("s_branch 1f\n\ts_nop 0\n1:\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v));, )
// f32 add with an SGPR operand, dependent (the replay's v_readlane + v_add pair)
KERNEL(k_rl_fadd, , asm volatile(REP16("v_readlane_b32 %1, %2, 7\n\tv_add_f32 %0, %1, %0\n\t") : "+v"(v), "+s"(s) : "v"(w));, )

template <typename F> static void timeit(const char* name, F kern, int per_iter) {
    unsigned* out; long long* ticks;
    (void)hipMalloc((void**)&out, 4096); (void)hipMalloc((void**)&ticks, 4096);
    const int iters = 4096;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, ticks, iters, 3u); (void)hipDeviceSynchronize(); }
    long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-64s: %.2f cycles per unit\n", name, (double)h / ((double)iters * per_iter));
}
int main() {
    timeit("dependent s_add_u32", k_salu, 16);
    timeit("dependent v_add_u32", k_valu, 16);
    timeit("v_readfirstlane -> v_add(sgpr)", k_rfl, 16);
    timeit("v_readfirstlane -> s_add -> v_mov", k_rfl_salu, 16);
    timeit("s_and -> (s_nop 3) -> v_readlane(dyn) -> s_add", k_rl_dyn, 16);
    timeit("v_readlane(const) -> v_add(sgpr)", k_rl_const, 16);
    timeit("v_cmp -> s_cbranch_vccnz (not taken) + v_add", k_vcc_branch, 16);
    timeit("v_readfirstlane -> s_cmp -> s_cbranch (not taken) + v_add", k_rfl_branch, 16);
    timeit("taken s_branch + v_add", k_taken, 16);
    timeit("v_readlane(const) -> v_add_f32(sgpr), dependent", k_rl_fadd, 16);
    return 0;
}
