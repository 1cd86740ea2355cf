// mfma_exact.hip -- is v_mfma_f32_16x16x4_f32 bit-identical to the reference's k-ordered chain  acc = acc + (x*w)
// (bf16-valued operands, product exact in f32)?  16x16 outputs, K = 4096, operands with a wide dynamic range.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_exact.hip -o tools/mfma_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float* A /*[16][K]*/, const float* B /*[K][16]*/, int K, float* D /*[16][16]*/) {
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < K / 4; g++) {
        const float a = A[i * K + 4 * g + kk], b = B[(4 * g + kk) * 16 + i];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];          // row = (lane>>4)*4 + r, col = lane & 15
}
static float bf(float f) { uint32_t u; memcpy(&u, &f, 4); u &= 0xFFFF0000u; memcpy(&f, &u, 4); return f; }
int main() {
    const int K = 4096;
    std::vector<float> A(16 * K), B(K * 16), D(256), R(256);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((int64_t)(s >> 33) - (1ll << 30)) / (float)(1 << 30); };
    int bad_total = 0;
    for (int trial = 0; trial < 8; trial++) {
        for (auto& v : A) { float e = rnd() * 12.0f; v = bf(rnd() * exp2f(e)); }
        for (auto& v : B) { float e = rnd() * 6.0f; v = bf(rnd() * exp2f(e)); }
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) { const float p = A[i * K + k] * B[k * 16 + j]; acc = acc + p; }   // the reference's order (two roundings; product exact)
            R[i * 16 + j] = acc;
        }
        float *dA, *dB, *dD;
        (void)hipMalloc((void**)&dA, A.size() * 4); (void)hipMalloc((void**)&dB, B.size() * 4); (void)hipMalloc((void**)&dD, 1024);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, K, dD);
        (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int q = 0; q < 256; q++) if (memcmp(&D[q], &R[q], 4)) bad++;
        printf("trial %d: %d / 256 outputs differ\n", trial, bad);
        bad_total += bad;
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    }
    printf("%s\n", bad_total ? "NOT bit-identical" : "v_mfma_f32_16x16x4_f32 == k-ordered f32 chain, bit for bit");
    return bad_total != 0;
}
