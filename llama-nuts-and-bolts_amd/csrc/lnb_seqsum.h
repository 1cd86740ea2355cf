// lnb_seqsum.h -- EXACT parallel evaluation of the reference's sequential f32 sum of NON-NEGATIVE terms.
//
// The reference's RMSNorm computes   sum = 0f; for k ascending: sum += x_k^2   in float32
// (src/ml/operations_impl.go:236-251 on the exact squares of :197-217): 4096 dependent roundings per norm, 65 norms
// per token.  Rounding is not associative, but for non-negative terms the running sum s is monotone, so it sits in
// one binade [2^e, 2^(e+1)) for long stretches, and INSIDE a binade an f32 add is integer arithmetic:
//     s = M * ulp,  M in [2^23, 2^24);   p = q * ulp (q a dyadic rational);   fl(s + p) = ulp * RNE(M + q)
//     RNE(M + q) = M + floor(q) + [frac > 1/2] + [frac == 1/2 and (M + floor(q)) odd]
// i.e. each step is a map  M -> M + c_{M mod 2}  that depends on M only through its PARITY.  Such maps are closed
// under composition ((c0,c1) pairs), so a block of steps can be collapsed WITHOUT knowing M, all blocks in parallel,
// and the serial part only walks block by block:  M += c_{M&1}.
// The binade of a block is guessed from approximate prefix sums; the guess is VERIFIED when the block is applied
// (exponent of the true running sum must match, and the result must stay below 2^24): on any mismatch the block is
// replayed with plain sequential adds.  The result is therefore bit-identical to the sequential loop for every
// input; the guesses only decide how many blocks take the fast path.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SEQ_HD __host__ __device__ __forceinline__
#else
#define SEQ_HD static inline
#endif

struct SeqBlock {
    int32_t c0, c1;     // offset added to M when M is even / odd on entry
    int32_t e;          // biased f32 exponent field the block was evaluated for
    int32_t ok;         // 0: a term was too large for that binade (or e not a normal exponent) -> replay sequentially
};

SEQ_HD uint32_t seq_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SEQ_HD float seq_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// fold the add of term p (>= 0, finite) into the block map, for running sums in the binade with biased exponent b.e.
// Branch-free: a shift of 25 or more means q < 1/2 (the term is absorbed), so clamping the shift to 25 gives
// floor = 0, rem = mp < half = 2^24 without a special case.
SEQ_HD void seq_step(SeqBlock& b, float p) {
    const uint32_t u = seq_f2u(p);
    const uint32_t ep = (u >> 23) & 0xFFu;
    const uint32_t mp = (u & 0x7FFFFFu) | (ep != 0u ? 0x800000u : 0u);   // subnormals: no implicit bit, scale of exponent 1
    const int32_t sh = b.e - (int32_t)(ep > 1u ? ep : 1u);               // q = mp >> sh
    b.ok &= (sh > 0 && ep != 0xFFu) ? 1 : 0;                             // p >= 2^e (or inf/nan): the sum leaves the binade
    const uint32_t shc = (uint32_t)(sh < 1 ? 1 : (sh > 25 ? 25 : sh));
    const uint32_t fl = mp >> shc, rem = mp & ((1u << shc) - 1u), half = 1u << (shc - 1u);
    const int32_t a = (int32_t)fl + (rem > half ? 1 : 0);
    const int32_t tie = rem == half ? 1 : 0;
    // parity of M after the steps folded so far: par ^ (c & 1)
    b.c0 = b.c0 + a + (tie & ((b.c0 ^ a) & 1));              // entry parity 0
    b.c1 = b.c1 + a + (tie & ((1 ^ b.c1 ^ a) & 1));          // entry parity 1
}

// apply a block map to the running sum s; returns 0 if the block must be replayed sequentially
SEQ_HD int seq_apply(float& s, const SeqBlock& b) {
    const uint32_t u = seq_f2u(s);
    if (!b.ok || (int32_t)((u >> 23) & 0xFF) != b.e || (u >> 31)) return 0;
    const uint32_t M = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t odd = 0u - (M & 1u);                          // branch- and table-free select (hipcc turned ?: into a scratch lookup)
    const uint32_t Mn = M + (((uint32_t)b.c1 & odd) | ((uint32_t)b.c0 & ~odd));
    if (Mn >= 0x1000000u) return 0;                           // left the binade somewhere inside the block
    s = seq_u2f(((uint32_t)b.e << 23) | (Mn & 0x7FFFFFu));
    return 1;
}

// biased exponent guess for a block whose entry sum is about `lo` and exit sum about `hi` (both approximate);
// returns 0 when the block should not be trusted to the fast path (near a binade edge, or spanning one)
SEQ_HD int32_t seq_guess(float lo, float hi) {
    const uint32_t ul = seq_f2u(lo * 0.9995f), uh = seq_f2u(hi * 1.0005f);
    const int32_t el = (int32_t)((ul >> 23) & 0xFF), eh = (int32_t)((uh >> 23) & 0xFF);
    if (el != eh || el == 0 || el == 0xFF) return 0;
    return el;
}
