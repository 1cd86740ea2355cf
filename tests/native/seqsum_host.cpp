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


// ---- the multi-wave form (rms_fold / rms_scale_wide in lnb_kernels.hip), emulated lane by lane -------------------------
// NH folding waves of 64 lanes; lane = one leaf of LEAF = seq_leaf_size(K, NH*64) terms (zero padded past K).  Per wave
// ("heap" of 64 leaves) a segmented inclusive scan composes the leaf maps of every run of equal-binade valid leaves; the
// walker adds the first `head` terms one by one, then visits only the run ends and the invalid leaves (item mask),
// applying the run's composed map when it starts exactly at the walker's position and verifies, else replaying the
// leaves' terms.  Returns the sum; *visits = items visited, *raw = leaves replayed.
static long leaf_mismatch = 0;
extern "C" long seqsum_leaf_mismatches() { return leaf_mismatch; }
extern "C" float seqsum_tree(const float* pin, int K, int NH, int head_terms, int* visits, int* raw) {
    const int LEAF = seq_leaf_size(K, NH * 64), nleaf = (K + LEAF - 1) / LEAF, TB = NH * 64;
    std::vector<float> pv((size_t)TB * LEAF + 64, 0.0f);
    for (int k = 0; k < K; k++) pv[k] = pin[k];
    const float* p = pv.data();
    std::vector<float> bs(TB, 0.0f), lo(TB), hi(TB);
    for (int b = 0; b < nleaf; b++) {
        const float* q = p + (size_t)b * LEAF;
        float v = 0.0f;
        for (int i = 0; i < LEAF; i += 4) v += (q[i] + q[i + 1]) + (q[i + 2] + q[i + 3]);       // (only a guess: any order)
        bs[b] = v;
    }
    float run = 0.0f;                                                      // approximate prefix: wave scans + wave totals
    for (int g = 0; g < NH; g++) {
        float incl = 0.0f;
        for (int l = 0; l < 64; l++) { const int b = g * 64 + l; lo[b] = run + incl; incl += bs[b]; hi[b] = run + incl; }
        run += incl;
    }
    int head = head_terms < K ? head_terms : K;
    const int headleaf = head / LEAF;
    head = headleaf * LEAF;
    std::vector<SeqNode> rec((size_t)TB);
    std::vector<uint64_t> items(NH);
    std::vector<int> scan_failed(NH, 0);
    for (int g = 0; g < NH; g++) {
        SeqNode n[64]; int f[64], st[64];
        for (int l = 0; l < 64; l++) {
            const int b = g * 64 + l;
            n[l].a = 0; n[l].b = 0;
            if (b < nleaf) {
                n[l] = seq_leaf(p + (size_t)b * LEAF, LEAF, lo[b], hi[b]);
                // the two-sums evaluation (device path) against the integer one, term by term: same node whenever the integer one is valid;
                // where only the simulated one is valid (it may accept what the conservative integer test rejects) the walk still verifies it
                const SeqNode chk = seq_leaf_steps(p + (size_t)b * LEAF, LEAF, lo[b], hi[b]);
                if ((chk.a >> 24) && (chk.a != n[l].a || chk.b != n[l].b)) leaf_mismatch++;
            }
            if (b < nleaf && bs[b] == 0.0f && !(n[l].a >> 24)) n[l].a = SEQ_ZERO_LEAF;
        }
        for (int l = 0; l < 64; l++) f[l] = seq_is_start(l, n[l], l ? n[l - 1] : n[l], g * 64 + l == headleaf);
        uint64_t mask = 0;
        for (int l = 0; l < 64; l++) if (!(n[l].a >> 24) || l == 63 || f[l + 1]) mask |= 1ull << l;
        for (int l = 0; l < 64; l++) st[l] = l;
        for (int d = 1; d < 64; d <<= 1) {                                 // Hillis-Steele, all lanes in lock step
            SeqNode nn[64]; int ff[64], ss[64];
            for (int l = 0; l < 64; l++) {
                nn[l] = n[l]; ff[l] = f[l]; ss[l] = st[l];
                if (l >= d && !f[l]) scan_failed[g] |= !seq_scan_step(nn[l], ff[l], ss[l], n[l - d], f[l - d], st[l - d]);
            }
            for (int l = 0; l < 64; l++) { n[l] = nn[l]; f[l] = ff[l]; st[l] = ss[l]; }
        }
        for (int l = 0; l < 64; l++) { rec[(size_t)g * 64 + l].a = n[l].a; rec[(size_t)g * 64 + l].b = n[l].b | ((uint32_t)st[l] << 24); }
        items[g] = mask;
    }
    int nv = 0, nr = 0;
    float s = 0.0f;
    for (int k = 0; k < head; k++) s += p[k];
    uint32_t sb = seq_f2u(s);
    for (int g = 0; g < NH; g++) {
        int nloc = nleaf - g * 64; nloc = nloc < 0 ? 0 : (nloc > 64 ? 64 : nloc);
        int pos = headleaf - g * 64; pos = pos < 0 ? 0 : pos;
        uint64_t mask = items[g], zm = 0;
        for (int l = 0; l < 64; l++) if (rec[(size_t)g * 64 + l].a == SEQ_ZERO_LEAF) zm |= 1ull << l;
        mask &= ~zm;                                                       // leaves of exact zeros: stepped over, not visited
        while (mask) {
            const int i = __builtin_ctzll(mask); mask &= mask - 1;
            if (i < pos) continue;
            if (i >= nloc) break;
            nv++;
            SeqNode n = rec[(size_t)g * 64 + i];
            int st = (int)(n.b >> 24); n.b &= 0xFFFFFFu;
            // rms_walk_fast (lnb_kernels.hip): the items tile the leaves by construction unless a composition of the scan failed, so the
            // record's start is only checked then (rms_walk_heap); a replay starts behind the previous item either way
            if (!scan_failed[g]) st = pos;
            else if (st > pos && ((((1ull << st) - 1ull) & (~0ull << pos)) & ~zm) == 0ull) pos = st;   // zeros: nothing to add
            if (!(st == pos && seq_apply_node(sb, n))) {
                float f = seq_u2f(sb);
                for (int l = pos; l <= i; l++) { const float* q = p + ((size_t)g * 64 + l) * LEAF; for (int t = 0; t < LEAF; t++) f += q[t]; nr++; }
                sb = seq_f2u(f);
            }
            pos = i + 1;
        }
    }
    if (visits) *visits = nv;
    if (raw) *raw = nr;
    return seq_u2f(sb);
}


// ---- round 4: the branch-free walk over a row's ITEM LIST (rms_fold's emission + rms_scale_wide's fast path), emulated -------------------
// Same leaves, same per-wave segmented scan as seqsum_tree; binades guessed with the tight margin; a leaf that crosses a binade edge is
// split (seq_split_leaf) instead of being replayed.  Items in leaf order: the end of every run (x = 0, the run's composed map) and the two
// items of every split leaf; leaves of exact zeros emit nothing; anything else that is not covered marks the row for the old walk.
// Every item is applied unconditionally, the checks OR-ed: a row whose checks all pass is done, any other row is handed to seqsum_tree.
// *fast = 1 if the list was enough, *n_items = its length.
static int dbg_reason = 0;
extern "C" int seqsum_dbg_reason() { return dbg_reason; }
extern "C" float seqsum_items(const float* pin, int K, int NH, int head_terms, int* fast, int* n_items) {
    const int LEAF = seq_leaf_size(K, NH * 64), nleaf = (K + LEAF - 1) / LEAF, TB = NH * 64;
    std::vector<float> pv((size_t)TB * LEAF + 64, 0.0f);
    for (int k = 0; k < K; k++) pv[k] = pin[k];
    const float* p = pv.data();
    std::vector<float> bs(TB, 0.0f), lo(TB), hi(TB);
    for (int b = 0; b < nleaf; b++) { const float* q = p + (size_t)b * LEAF; float v = 0.0f; for (int i = 0; i < LEAF; i += 4) v += (q[i] + q[i + 1]) + (q[i + 2] + q[i + 3]); bs[b] = v; }
    float run = 0.0f;
    for (int g = 0; g < NH; g++) { float incl = 0.0f; for (int l = 0; l < 64; l++) { const int b = g * 64 + l; lo[b] = run + incl; incl += bs[b]; hi[b] = run + incl; } run += incl; }
    int head = head_terms < K ? head_terms : K;
    const int headleaf = head / LEAF;
    head = headleaf * LEAF;
    std::vector<SeqItem> list;
    bool row_bad = false;
    dbg_reason = 0;
    for (int g = 0; g < NH; g++) {
        SeqNode n[64]; int f[64], st[64]; SeqSplit sp[64]; bool zero[64];
        for (int l = 0; l < 64; l++) {
            const int b = g * 64 + l;
            n[l].a = 0; n[l].b = 0; sp[l].ok = 0; zero[l] = false;
            if (b < nleaf) {
                const int32_t e = seq_guess_tight(lo[b], hi[b]);
                float s0, s1; seq_sim_init(e, s0, s1);
                for (int i = 0; i < LEAF; i++) { s0 = s0 + p[(size_t)b * LEAF + i]; s1 = s1 + p[(size_t)b * LEAF + i]; }
                n[l] = seq_sim_node(e, s0, s1);
                if (bs[b] == 0.0f && !(n[l].a >> 24)) { n[l].a = SEQ_ZERO_LEAF; zero[l] = true; }
                else if (!(n[l].a >> 24)) sp[l] = seq_split_leaf(p + (size_t)b * LEAF, LEAF, lo[b], hi[b]);
            }
        }
        for (int l = 0; l < 64; l++) f[l] = seq_is_start(l, n[l], l ? n[l - 1] : n[l], g * 64 + l == headleaf);
        uint64_t mask = 0;
        for (int l = 0; l < 64; l++) if (!(n[l].a >> 24) || l == 63 || f[l + 1 < 64 ? l + 1 : 63] || l == 63) mask |= 1ull << l;
        for (int l = 0; l + 1 < 64; l++) if (f[l + 1]) mask |= 1ull << l;
        for (int l = 0; l < 64; l++) st[l] = l;
        int failed = 0;
        for (int d = 1; d < 64; d <<= 1) {
            SeqNode nn[64]; int ff[64], ss[64];
            for (int l = 0; l < 64; l++) { nn[l] = n[l]; ff[l] = f[l]; ss[l] = st[l]; if (l >= d && !f[l]) failed |= !seq_scan_step(nn[l], ff[l], ss[l], n[l - d], f[l - d], st[l - d]); }
            for (int l = 0; l < 64; l++) { n[l] = nn[l]; f[l] = ff[l]; st[l] = ss[l]; }
        }
        if (failed) { row_bad = true; dbg_reason |= 1; }
        for (int l = 0; l < 64; l++) {
            const int b = g * 64 + l;
            if (!((mask >> l) & 1) || b < headleaf || b >= nleaf || zero[l]) continue;
            if (n[l].a >> 24) list.push_back(seq_item_of_node(n[l]));            // the end of a run: its composed map
            else if (sp[l].ok) { list.push_back(sp[l].a); list.push_back(sp[l].b); }
            else {                                                                // round 6: a leaf no guess covers (too close to an edge): its terms one by one, as single-term items
                const float* q = p + (size_t)b * LEAF;
                bool finite = true;
                for (int i = 0; i < LEAF; i++) if (!(q[i] <= 3.4028234e38f)) finite = false;      // inf / NaN squares: the record walk's business
                if (!finite) { row_bad = true; dbg_reason |= 2; }
                else for (int i = 0; i < LEAF; i++) list.push_back(seq_item_of_term(q[i]));
            }
        }
    }
    if (n_items) *n_items = (int)list.size();
    float s = 0.0f;
    for (int k = 0; k < head; k++) s += p[k];
    uint32_t sb = seq_f2u(s), bad = 0;
    for (const SeqItem& it : list) sb = seq_item_apply(sb, it, bad);
    if (bad) dbg_reason |= 4;
    if (row_bad || bad || list.size() > 64) { if (fast) *fast = 0; return seqsum_tree(pin, K, NH, head_terms, nullptr, nullptr); }
    if (fast) *fast = 1;
    return seq_u2f(sb);
}
