// mfma_bf16_exact.hip -- are the bf16 matrix-core instructions of gfx950 bit-identical to the reference's k-ordered
// chain  acc = acc + (x*w)  (f32 accumulator rounded after every term)?  Probes 16x16x16, 32x32x8, 4x4x4 (_1k forms)
// and the gfx950 16x16x32 form.  Companion of mfma_exact.hip (the f32 16x16x4 form IS identical).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_bf16_exact.hip -o tools/mfma_bf16_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A: [M][K] bf16 bits, B: [K][N] bf16 bits (row-major), D: [M][N] f32
__global__ void k_16x16x16(const uint16_t* A, const uint16_t* B, int K, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        s16x4 a, b;
        for (int e = 0; e < 4; e++) { a[e] = (short)A[i * K + k0 + 4 * g + e]; b[e] = (short)B[(k0 + 4 * g + e) * 16 + i]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(g * 4 + r) * 16 + i] = acc[r];
}
__global__ void k_32x32x8(const uint16_t* A, const uint16_t* B, int K, float* D) {
    const int l = threadIdx.x, i = l & 31, g = l >> 5;
    f32x16 acc; for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        s16x4 a, b;
        for (int e = 0; e < 4; e++) { a[e] = (short)A[i * K + k0 + 4 * g + e]; b[e] = (short)B[(k0 + 4 * g + e) * 32 + i]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
    }
    // D layout: col = lane & 31, row = 8*(r/4) + 4*... : row = (r / 4) * 8 + g * 4 + (r % 4)
    for (int r = 0; r < 16; r++) D[((r >> 2) * 8 + g * 4 + (r & 3)) * 32 + i] = acc[r];
}
__global__ void k_16x16x32(const uint16_t* A, const uint16_t* B, int K, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        union { bf16x8 v; uint16_t u[8]; } a, b;
        for (int e = 0; e < 8; e++) { a.u[e] = A[i * K + k0 + 8 * g + e]; b.u[e] = B[(k0 + 8 * g + e) * 16 + i]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(g * 4 + r) * 16 + i] = acc[r];
}

static float wide(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t trunc16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

template <class L> static int run(const char* name, int M, int N, int K, L launch) {
    std::vector<uint16_t> A((size_t)M * K), B((size_t)K * N);
    std::vector<float> D((size_t)M * N), R((size_t)M * N);
    uint64_t s = 777;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((int64_t)(s >> 33) - (1ll << 30)) / (float)(1 << 30); };
    int bad_total = 0;
    for (int trial = 0; trial < 4; trial++) {
        // trial 0: one instruction's worth of K only (isolates the in-instruction order); later trials: long chains
        const int Kt = trial == 0 ? 32 : K;
        for (auto& v : A) v = trunc16(rnd() * exp2f(rnd() * 6.0f));
        for (auto& v : B) v = trunc16(rnd() * exp2f(rnd() * 3.0f));
        std::vector<uint16_t> At((size_t)M * Kt), Bt((size_t)Kt * N);
        for (int i = 0; i < M; i++) for (int k = 0; k < Kt; k++) At[(size_t)i * Kt + k] = A[(size_t)i * K + k];
        for (int k = 0; k < Kt; k++) for (int j = 0; j < N; j++) Bt[(size_t)k * N + j] = B[(size_t)k * N + j];
        for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < Kt; k++) { const float p = wide(At[(size_t)i * Kt + k]) * wide(Bt[(size_t)k * N + j]); acc = acc + p; }
            R[(size_t)i * N + j] = acc;
        }
        uint16_t *dA, *dB; float* dD;
        (void)hipMalloc((void**)&dA, At.size() * 2); (void)hipMalloc((void**)&dB, Bt.size() * 2); (void)hipMalloc((void**)&dD, D.size() * 4);
        (void)hipMemcpy(dA, At.data(), At.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
        launch(dA, dB, Kt, dD);
        (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t q = 0; q < D.size(); q++) if (memcmp(&D[q], &R[q], 4)) bad++;
        printf("%-28s K=%5d: %d / %d outputs differ\n", name, Kt, bad, (int)D.size());
        bad_total += bad;
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    }
    return bad_total;
}

int main() {
    int b1 = run("v_mfma_f32_16x16x16_bf16", 16, 16, 4096, [](uint16_t* a, uint16_t* b, int K, float* d) { hipLaunchKernelGGL(k_16x16x16, dim3(1), dim3(64), 0, 0, a, b, K, d); });
    int b2 = run("v_mfma_f32_32x32x8_bf16", 32, 32, 4096, [](uint16_t* a, uint16_t* b, int K, float* d) { hipLaunchKernelGGL(k_32x32x8, dim3(1), dim3(64), 0, 0, a, b, K, d); });
    int b3 = run("v_mfma_f32_16x16x32_bf16", 16, 16, 4096, [](uint16_t* a, uint16_t* b, int K, float* d) { hipLaunchKernelGGL(k_16x16x32, dim3(1), dim3(64), 0, 0, a, b, K, d); });
    printf("16x16x16: %s\n32x32x8: %s\n16x16x32: %s\n", b1 ? "NOT identical" : "IDENTICAL", b2 ? "NOT identical" : "IDENTICAL", b3 ? "NOT identical" : "IDENTICAL");
    return 0;
}
