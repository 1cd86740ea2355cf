// Host harness for csrc/lnb_seqsum.h: emulates the 64-lane device algorithm with loops so the exact-parallel
// sequential sum can be fuzzed against the plain sequential f32 loop on the CPU (tests/test_seqsum.py).
// build: g++ -O2 -ffp-contract=off -shared -fPIC tests/native/seqsum_host.cpp -o tests/native/libseqsum_host.so
#include "../../llama-nuts-and-bolts_amd/csrc/lnb_seqsum.h"
#include <vector>

extern "C" float seqsum_ref(const float* p, int K) {
    float s = 0.0f;
    for (int k = 0; k < K; k++) s += p[k];
    return s;
}

// same structure as rms_scale_scan() in lnb_kernels.hip: nb blocks of bs = K/nb terms
extern "C" float seqsum_scan(const float* p, int K, int nb, int* n_fast_out) {
    const int bs = K / nb;
    std::vector<float> P(nb), S(nb + 1);
    for (int l = 0; l < nb; l++) { float a = 0.0f; for (int i = 0; i < bs; i++) a += p[l * bs + i]; P[l] = a; }
    S[0] = 0.0f;
    for (int l = 0; l < nb; l++) S[l + 1] = S[l] + P[l];          // (device: wave prefix scan; any order is fine, it is only a guess)
    std::vector<SeqBlock> B(nb);
    for (int l = 0; l < nb; l++) {
        SeqBlock b; b.c0 = 0; b.c1 = 0; b.e = seq_guess(S[l], S[l + 1]); b.ok = b.e != 0;
        if (b.ok) for (int i = 0; i < bs; i++) { seq_step(b, p[l * bs + i]); if ((uint32_t)b.c0 > 0x2000000u || (uint32_t)b.c1 > 0x2000000u) b.ok = 0; }
        B[l] = b;
    }
    float s = 0.0f; int fast = 0;
    for (int l = 0; l < nb; l++) {
        if (seq_apply(s, B[l])) { fast++; continue; }
        for (int i = 0; i < bs; i++) s += p[l * bs + i];
    }
    if (n_fast_out) *n_fast_out = fast;
    return s;
}
