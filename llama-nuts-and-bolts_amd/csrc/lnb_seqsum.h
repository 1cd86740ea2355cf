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

// ---- packed nodes and their composition (the multi-wave tree form used by the RMSNorm prologue) --------------------
// A node is the parity map of a run of consecutive terms evaluated for ONE binade e:  M -> M + c_{M&1}.
// Packed in two words: a = (e << 24) | c0, b = c1, with c0, c1 < 2^24; e == 0 means "invalid, replay the terms".
// Such maps compose:  (f then g)(par) = f.c[par] + g.c[par ^ (f.c[par] & 1)]  when both were evaluated for the same e.
struct SeqNode { uint32_t a, b; };
// an INVALID node (e == 0: no binade could be guessed) whose terms are all exactly +0: still cut out of every run, but the walker
// skips it instead of replaying it (x + 0 == x).  Rows that start with zeros have no binade until the first non-zero term.
#define SEQ_ZERO_LEAF 1u

SEQ_HD SeqNode seq_pack(const SeqBlock& blk) {
    SeqNode n; n.a = 0; n.b = 0;
    if (blk.ok && blk.e > 0 && blk.e < 0xFF && (uint32_t)blk.c0 < 0x1000000u && (uint32_t)blk.c1 < 0x1000000u) {
        n.a = ((uint32_t)blk.e << 24) | (uint32_t)blk.c0; n.b = (uint32_t)blk.c1;
    }
    return n;
}
SEQ_HD SeqNode seq_compose(const SeqNode& f, const SeqNode& g) {
    const uint32_t ef = f.a >> 24, eg = g.a >> 24;
    const uint32_t fc0 = f.a & 0xFFFFFFu, fc1 = f.b, gc0 = g.a & 0xFFFFFFu, gc1 = g.b;
    const uint32_t o0 = 0u - (fc0 & 1u), o1 = 0u - (fc1 & 1u);           // mask selects (no ?: -> no scratch lookup tables)
    const uint32_t c0 = fc0 + ((gc1 & o0) | (gc0 & ~o0));                // entry parity 0 -> parity fc0&1 in front of g
    const uint32_t c1 = fc1 + ((gc0 & o1) | (gc1 & ~o1));                // entry parity 1 -> parity 1^(fc1&1)
    const uint32_t ok = (ef != 0u && ef == eg && c0 < 0x1000000u && c1 < 0x1000000u) ? 0xFFFFFFFFu : 0u;
    SeqNode h; h.a = ((ef << 24) | c0) & ok; h.b = c1 & ok;
    return h;
}
// apply a node to the running sum given as f32 bits (sign 0); returns 0 if the terms must be replayed
SEQ_HD int seq_apply_node(uint32_t& sb, const SeqNode& n) {
    const uint32_t e = n.a >> 24, es = sb >> 23;
    if (e == 0u || e != es) return 0;
    const uint32_t M = (sb & 0x7FFFFFu) | 0x800000u;
    const uint32_t Mn = M + ((M & 1u) ? n.b : (n.a & 0xFFFFFFu));
    if (Mn >= 0x1000000u) return 0;                                       // left the binade somewhere inside the run
    sb = (es << 23) | (Mn & 0x7FFFFFu);
    return 1;
}
// leaf size of the multi-wave form: the K terms are split over at most `lanes` leaves (one per folding lane); a multiple of
// 4 (float4 reads), at least 8.  The last leaf may run past K: the terms there are +0 and change nothing.
SEQ_HD int seq_leaf_size(int K, int lanes) { int l = (K + lanes - 1) / lanes; l = (l + 3) & ~3; return l < 8 ? 8 : l; }
// fold `n` consecutive terms into a leaf node for the binade guessed from the approximate sums around them
SEQ_HD SeqNode seq_leaf_steps(const float* p, int n, float lo, float hi) {      // the integer evaluation, term by term (kept as the checker of seq_leaf)
    SeqBlock b; b.c0 = 0; b.c1 = 0; b.e = seq_guess(lo, hi); b.ok = b.e != 0;
    for (int i = 0; i < n; i++) seq_step(b, p[i]);
    return seq_pack(b);
}
// The same node from TWO FLOATING-POINT SUMS (round 4; ~2 operations per term instead of ~27): the map M -> M + c_{M&1} holds for every
// M of the binade, so c0 and c1 can be read off two simulated running sums, one started at the binade's smallest even significand
// (s = 2^e, M = 2^23) and one at the smallest odd one (M = 2^23 + 1) -- the hardware's own round-to-nearest-even add does the work:
//     c0 = bits(fl(..fl(2^e + p_0) + ..p_{n-1})) - bits(2^e),   c1 likewise from 2^e (1 + 2^-23).
// Valid iff both sums are still inside the binade at the end (the terms are non-negative, so then they never left it; a term >= 2^e, an
// infinity or a NaN ends outside).  f32 adds must be IEEE (denormals honoured, no contraction): -ffp-contract=off, as everywhere.
SEQ_HD void seq_sim_init(int32_t e, float& s0, float& s1) { s0 = seq_u2f((uint32_t)e << 23); s1 = seq_u2f(((uint32_t)e << 23) | 1u); }
SEQ_HD SeqNode seq_sim_node(int32_t e, float s0, float s1) {
    const uint32_t u0 = seq_f2u(s0), u1 = seq_f2u(s1), B0 = (uint32_t)e << 23;
    SeqNode n; n.a = 0; n.b = 0;
    if (e > 0 && (u0 >> 23) == (uint32_t)e && (u1 >> 23) == (uint32_t)e) { n.a = ((uint32_t)e << 24) | (u0 - B0); n.b = u1 - (B0 | 1u); }
    return n;
}
SEQ_HD SeqNode seq_leaf(const float* p, int n, float lo, float hi) {
    const int32_t e = seq_guess(lo, hi);
    float s0, s1; seq_sim_init(e, s0, s1);
    for (int i = 0; i < n; i++) { s0 = s0 + p[i]; s1 = s1 + p[i]; }
    return seq_sim_node(e, s0, s1);
}

// ---- items of the branch-free walk (round 4) ----------------------------------------------------------------------------
// The serial part of the multi-wave form used to visit its items with scalar code and to REPLAY every leaf that crosses into the next
// binade term by term; both are slow for reasons that have nothing to do with arithmetic (a v_readlane with a computed lane ~50 cycles,
// a branch on a vector result ~40, the replay's cold path ~500).  An ITEM advances the running sum s (f32 bits, s >= +0) by
//     u = bits(f32(s) + x);   t = u + c0 + (u & 1) * d;        valid iff u and t lie in binade e
// -- one exact f32 add followed by a parity map -- which covers both kinds of step: a run of leaves (x = +0: the add is the identity) and
// a leaf that CROSSES a binade edge, split at the crossing term x*: (0, map of the terms before x*, e) then (x*, map of the terms after
// x*, e + 1).  Where the sum crosses is guessed from the approximate prefix sums like every binade here, and like every guess it is
// verified when applied: the first item's check says every term before x* kept the sum inside binade e, the second's that s + x* is in
// binade e + 1 and stays there.  Every item of a row is applied unconditionally (no branch), the checks are OR-ed, and a row with any
// failed check -- or with a leaf that is neither a run member nor cleanly split -- is walked again the old way (seq_apply_node + replays).
struct SeqItem { uint32_t x, c0; int32_t d; uint32_t e; };
SEQ_HD SeqItem seq_item_of_node(const SeqNode& n) {             // a node of the scan (its b may carry the run's first leaf in the top byte)
    SeqItem it; it.x = 0u; it.c0 = n.a & 0xFFFFFFu; it.d = (int32_t)(n.b & 0xFFFFFFu) - (int32_t)it.c0; it.e = n.a >> 24;
    return it;
}
// Round 6: a SINGLE-TERM item (x = the term, c0 = d = 0, e = SEQ_ANY_BINADE) is one exact f32 add with no assumption about the binade at all -- what a leaf
// that sits too close to a binade edge for any guess is replayed as, INSIDE the branch-free walk instead of condemning the row to the record walk.
#define SEQ_ANY_BINADE 0u
SEQ_HD uint32_t seq_item_apply(uint32_t s, const SeqItem& it, uint32_t& bad) {
    const uint32_t u = seq_f2u(seq_u2f(s) + seq_u2f(it.x));
    const uint32_t t = u + it.c0 + (uint32_t)((int32_t)(u & 1u) * it.d);
    bad |= ((t ^ u) >> 23) | (it.e != SEQ_ANY_BINADE ? (it.e ^ (u >> 23)) : 0u);
    return t;
}
SEQ_HD SeqItem seq_item_of_term(float x) { SeqItem it; it.x = seq_f2u(x); it.c0 = 0u; it.d = 0; it.e = SEQ_ANY_BINADE; return it; }
// binade guess with a margin just above the error of the approximate prefix sums (f32 tree sums of a few thousand non-negative terms:
// ~1e-6 relative): a wrong guess costs a fallback, never a wrong result
// Round 6: the margin is a parameter.  2e-6 was both too wide (1-2 % of gaussian rows had a crossing "too close to call") and too narrow (1-3 % had a leaf whose
// TRUE f32 running sum -- which wanders ~sqrt(n) 6e-8 from the exact prefix -- crossed on the other side of the guess and failed the walk's check); either way the
// whole row fell back to the record walk: 2.7 % of rows at K = 4096, 6.8 % at 8192.  With single-term items a leaf near an edge costs LEAF items instead of the
// row, so the margin can be generous: SEQ_MARGIN = 8e-6.
#ifndef SEQ_MARGIN
#define SEQ_MARGIN 8e-6f
#endif
SEQ_HD int32_t seq_guess_tight(float lo, float hi) {
    const uint32_t ul = seq_f2u(lo * (1.0f - SEQ_MARGIN)), uh = seq_f2u(hi * (1.0f + SEQ_MARGIN));
    const int32_t el = (int32_t)((ul >> 23) & 0xFF), eh = (int32_t)((uh >> 23) & 0xFF);
    if (el != eh || el == 0 || el == 0xFF) return 0;
    return el;
}
// a leaf whose approximate prefix sums say "enters the next binade here": split at the term that takes the approximate running sum across
// the edge, if that is unambiguous (no approximate sum of the leaf -- the one in front of it included -- within SEQ_MARGIN (relative) of the edge:
// several times the error of the approximate prefix).  One pass, no data-dependent branch: the device runs it for a whole wave at once (terms in
// front of the crossing feed the sums of binade el, terms behind it those of el + 1; + 0 leaves a sum unchanged).  p: 16-byte aligned,
// n a multiple of 4.  A wrong split costs a fallback, never a wrong result -- the walk verifies every item.
struct SeqSplit { int ok; SeqItem a, b; };
#define SEQ_FMIN(a, b) __builtin_fminf(a, b)
#define SEQ_FABS(a) __builtin_fabsf(a)
SEQ_HD int seq_split_candidate(float lo, float hi) {
    const uint32_t el = (seq_f2u(lo) >> 23) & 0xFFu, eh = (seq_f2u(hi) >> 23) & 0xFFu;
    return (el >= 1u && eh > el && eh < 0xFFu) ? 1 : 0;       // (one term may lift the sum several binades: an outlier channel)
}
SEQ_HD SeqSplit seq_split_leaf(const float* p, int n, float lo, float hi) {
    SeqSplit r;
    const int cand = seq_split_candidate(lo, hi);
    const int32_t el = cand ? (int32_t)((seq_f2u(lo) >> 23) & 0xFFu) : 1, eh = cand ? (int32_t)((seq_f2u(hi) >> 23) & 0xFFu) : 2;
    const float edge = seq_u2f((uint32_t)eh << 23);
    float a0, a1, b0, b1; seq_sim_init(el, a0, a1); seq_sim_init(eh, b0, b1);
    // ten operations per term on the device: add, compare, three selects (one mask operation), subtract, min, two packed adds
    float run = lo, xs = 0.0f, clear = edge - lo;                    // clear: the smallest |approximate sum - edge| seen (lo < edge: el < eh)
    bool was = false;
    const float* q = (const float*)__builtin_assume_aligned(p, 16);
    for (int i = 0; i < n; i += 4) {
        const float vv[4] = {q[i], q[i + 1], q[i + 2], q[i + 3]};
        for (int j = 0; j < 4; j++) {
            const float v = vv[j], nx = run + v;
            const bool is = nx >= edge;
            const float ta = is ? 0.0f : v, tb = was ? v : 0.0f;
            xs = (is && !was) ? v : xs;
            clear = SEQ_FMIN(clear, SEQ_FABS(nx - edge));
            a0 = a0 + ta; a1 = a1 + ta; b0 = b0 + tb; b1 = b1 + tb;
            run = nx; was = is;
        }
    }
    const SeqNode na = seq_sim_node(el, a0, a1), nb = seq_sim_node(eh, b0, b1);
    r.ok = (cand && was && clear > edge * SEQ_MARGIN && (na.a >> 24) != 0u && (nb.a >> 24) != 0u) ? 1 : 0;
    r.a = seq_item_of_node(na); r.b = seq_item_of_node(nb); r.b.x = seq_f2u(xs);
    return r;
}

// ---- segmented inclusive scan of leaf maps (one wave = 64 leaves) -------------------------------------------------
// a leaf starts a new run when it or its left neighbour is invalid, when the binade changes, or when forced
SEQ_HD int seq_is_start(int lane, const SeqNode& me, const SeqNode& left, int forced) {
    const uint32_t e = me.a >> 24, el = left.a >> 24;
    return (lane == 0 || e == 0u || el == 0u || e != el || forced) ? 1 : 0;
}
// one Hillis-Steele step for a lane that is not cut yet (f == 0): absorb the run that ends right in front of this one
// returns 0 when the composition did not go through (cannot happen inside a verified binade): the lane is cut there, the leaves in
// front of it are then covered by no item, and the folding wave reports it so that the walker checks every record's `start` (rms_walk_heap)
SEQ_HD int seq_scan_step(SeqNode& n, int& f, int& start, const SeqNode& o, int of, int ostart) {
    const SeqNode h = seq_compose(o, n);
    if (h.a >> 24) { n = h; f = of; start = ostart; return 1; }
    f = 1;
    return 0;
}
