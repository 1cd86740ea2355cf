// gemmbench.hip -- where does gemm_mfma_kernel's time go?  Includes the product kernel source and times one launch shape with
// parts of the slab loop compiled out (-DGM_DBG: 1 = stage only the first slab, 2 = no barriers, 4 = no LDS fragment reads).
// build: for d in 0 1 3 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGM_DBG=$d -Illama-nuts-and-bolts_amd/csrc -Iinclude tools/gemmbench.hip -o tools/gemmbench_$d; done
#include "../llama-nuts-and-bolts_amd/csrc/lnb_kernels.hip"
#include <cstdio>
#define GM_STR2(x) #x
#define GM_STR(x) GM_STR2(x)
#ifdef GM_PD_SET
#define GM_PD_STR GM_STR(GM_PD_SET)
#else
#define GM_PD_STR "default"
#endif
#include <vector>
int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    uint16_t *w, *x, *out;
    (void)hipMalloc((void**)&w, (size_t)N * K * 2); (void)hipMalloc((void**)&x, (size_t)S * K * 2); (void)hipMalloc((void**)&out, (size_t)S * N * 2);
    std::vector<uint16_t> h((size_t)N * K);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 24));
    (void)hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(x, h.data(), (size_t)S * K * 2, hipMemcpyHostToDevice);
    GemmParams p{}; p.w = w; p.rw = 16; p.nch = 1; p.x = x; p.K = K; p.n_rows = N; p.S = S; p.out = out;
    const int WN = argc > 5 ? atoi(argv[5]) : 4;                                       // 4: 64-row tiles; 1: 16-row tiles, waves split the batch rows
    const int MB = argc > 6 ? atoi(argv[6]) : 128;                                     // batch rows per workgroup (64: WN = 4 only)
    auto k4 = WN == 4 ? (MB == 64 ? gemm_mfma_kernel<EPI_STORE, 1, 4, 64> : gemm_mfma_kernel<EPI_STORE, 1, 4, 128>) : gemm_mfma_kernel<EPI_STORE, 1, 1, 128>;
    const size_t lds = (argc > 4 && atoi(argv[4]) > 0) ? (size_t)atoi(argv[4]) : gemm_lds_bytes(1, WN, MB);      // (a larger value forces one workgroup per CU)
    (void)hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid((N + 16 * WN - 1) / (16 * WN), (S + MB - 1) / MB);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k4, grid, dim3(256), lds, 0, p);
    (void)hipEventRecord(e0, 0);
    const int IT = 10;
    for (int i = 0; i < IT; i++) hipLaunchKernelGGL(k4, grid, dim3(256), lds, 0, p);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("GM_DBG=%d PD=%s WN=%d MB=%d lds=%zu S=%d N=%d K=%d: %.1f us per launch, %.1f TFLOP/s (%.0f %% of 157.3)  err=%s\n", GM_DBG, GM_PD_STR, WN, MB, lds, S, N, K, 1e3 * ms / IT,
           2.0 * S * N * K / (ms / IT * 1e-3) / 1e12, 100.0 * 2.0 * S * N * K / (ms / IT * 1e-3) / 1e12 / 157.3, hipGetErrorString(hipGetLastError()));
    return 0;
}
