// gemmstream_bench.hip -- where does gemm_stream_kernel's time go?  Includes the product kernel source and times one launch shape, with
// parts of the chunk loop compiled out (-DGS_DBG: 1 = stage only the first two chunks, 2 = no barriers, 4 = no LDS operand reads).
// build: for d in 0 1 3 4 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGS_DBG=$d -Illama-nuts-and-bolts_amd/csrc -Iinclude tools/gemmstream_bench.hip -o tools/gemmstream_bench_$d; done
// run:   tools/gemmstream_bench_0 S N K [ntw [nch [lds_bytes]]]     (ntw 0 = the launcher's own choice);   tools/gemmstream_bench_0 peak
#include "../llama-nuts-and-bolts_amd/csrc/lnb_kernels.hip"
#include <cstdio>
#include <cstring>
#include <vector>
template <int NCH, int NTW> static void launch(dim3 grid, size_t lds, const GemmParams& q) {
    constexpr int EPI = NCH == 2 ? EPI_SILU_MUL : EPI_STORE;              // (two chains: the gate|up epilogue, so that both chains are live)
    (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, NCH, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((gemm_stream_kernel<EPI, NCH, NTW>), grid, dim3(256), lds, 0, q);
}
// the chip's f32 matrix rate under load: NACC independent accumulators per wave, nothing but matrix instructions
template <int NACC> __global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x, b = 1.0f + (float)blockIdx.x;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
static void peak(int wgs_per_cu) {
    float* d; (void)hipMalloc((void**)&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, NACC = 8;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_peak_kernel<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2048.0 * 8 * NACC * iters * 4 * 256 * wgs_per_cu;
    printf("f32 16x16x4 matrix instructions only, %d wave(s) per SIMD on 256 CUs, %.0f ms: %.1f TFLOP/s (%.0f %% of 157.3 = 2.4 GHz)\n", wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / 1e12 / 157.3);
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "peak")) { peak(1); peak(2); return 0; }
    const int S = argc > 1 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    int ntw = argc > 4 ? atoi(argv[4]) : 0;
    const int nch = argc > 5 ? atoi(argv[5]) : 1;            // 2: N counts gate|up PAIRS of rows
    const size_t lds_force = argc > 6 ? (size_t)atoi(argv[6]) : 0;
    const int num_cus = 256, n_tiles = (N + 15) / 16;
    uint16_t *w, *x, *out;
    const size_t welems = (size_t)n_tiles * nch * 16 * K;
    (void)hipMalloc((void**)&w, welems * 2); (void)hipMalloc((void**)&x, (size_t)S * K * 2); (void)hipMalloc((void**)&out, (size_t)S * N * 2);
    std::vector<uint16_t> h(welems > (size_t)S * K ? welems : (size_t)S * K);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 24));
    (void)hipMemcpy(w, h.data(), welems * 2, hipMemcpyHostToDevice);          // (timing only: any bits do as an M16 image)
    (void)hipMemcpy(x, h.data(), (size_t)S * K * 2, hipMemcpyHostToDevice);
    float* silu; (void)hipMalloc((void**)&silu, 65536 * 4); (void)hipMemset(silu, 0x3c, 65536 * 4);
    GemmParams p{}; p.silu = silu; p.w16 = w; p.nch = nch; p.x = x; p.K = K; p.n_rows = N; p.S = S; p.out = out;
    if (ntw == 0) ntw = lnb_gemm_stream_ntw(n_tiles, (S + 15) / 16, nch, num_cus);
    const int rows_wg = 16 * ntw;
    unsigned gx = (unsigned)((n_tiles + 3) / 4); if (gx > (unsigned)num_cus) gx = num_cus;
    const dim3 grid(gx, (unsigned)((S + rows_wg - 1) / rows_wg));
    p.rows_fastest = getenv("GS_ORDER") ? atoi(getenv("GS_ORDER")) : lnb_gemm_stream_rows_fastest((int)grid.y);
    const size_t lds = lds_force ? lds_force : (size_t)2 * rows_wg * GS_PITCH * 4;
    auto go = [&]() {
        if (nch == 1) switch (ntw) { case 1: launch<1, 1>(grid, lds, p); break; case 2: launch<1, 2>(grid, lds, p); break; default: launch<1, 4>(grid, lds, p); }
        else switch (ntw) { case 1: launch<2, 1>(grid, lds, p); break; case 2: launch<2, 2>(grid, lds, p); break; default: launch<2, 4>(grid, lds, p); }
    };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) go();
    (void)hipEventRecord(e0, 0);
    const int IT = 10;
    for (int i = 0; i < IT; i++) go();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * S * N * nch * K / (ms / IT * 1e-3) / 1e12;
    printf("GS_DBG=%d occ=%d/%d R=%d/%d nch=%d ntw=%d rows_wg=%d grid=(%u,%u) lds=%zu S=%d N=%d K=%d: %.1f us per launch, %.1f TFLOP/s (%.0f %% of 157.3)  err=%s\n", GS_DBG, GS_OCC1, GS_OCC2, GS_R1, GS_R2, nch, ntw, rows_wg, grid.x, grid.y, lds,
           S, N, K, 1e3 * ms / IT, tf, 100.0 * tf / 157.3, hipGetErrorString(hipGetLastError()));
    return 0;
}
