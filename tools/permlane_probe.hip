// permlane_probe.hip -- the row-swap instructions new in gfx950 (v_permlane16_swap_b32, v_permlane32_swap_b32), as the hardware executes them:
// gemm_stream_kernel's chain-layout feed (lnb_batch_kernels.h, SRC 2) builds its 4 x 4 row transpose on them.  Prints, per 16-lane row, which
// (operand, row) each result row came from.   build: hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x, a = l, b = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1]; o[128 + l] = s[0]; o[192 + l] = s[1];
}
int main() {
    unsigned* d; unsigned h[256];
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[4] = {"permlane16_swap result 0 (first operand)", "permlane16_swap result 1 (second operand)", "permlane32_swap result 0", "permlane32_swap result 1"};
    for (int q = 0; q < 4; q++) {
        printf("%-44s rows:", names[q]);
        for (int row = 0; row < 4; row++) { unsigned v = h[q * 64 + row * 16]; printf("  %s.row%u", v >= 100 ? "B" : "A", (v % 100) / 16); }
        bool lanes_ok = true;
        for (int l = 0; l < 64; l++) if ((h[q * 64 + l] % 100) % 16 != (unsigned)(l % 16)) lanes_ok = false;
        printf("   (lane within row preserved: %s)\n", lanes_ok ? "yes" : "NO");
    }
    return 0;
}
