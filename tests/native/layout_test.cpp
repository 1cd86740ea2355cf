// layout_test.cpp -- host-side check of the HBM layout functions the re-tilers and the kernels share (csrc/lnb_device.h): the tiled GEMV
// layout, the M16 matrix-core layout of the weights and the B-operand layout of the batch's activations.  Pure index arithmetic: runs
// without a GPU (tests/test_layouts.py, g++).
#include "../../llama-nuts-and-bolts_amd/csrc/lnb_device.h"
#include <cstdio>
#include <vector>
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 10) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)
int main() {
    // ---- M16: [tile n/16][chain][chunk k/128][m = (k%16)/4][i = n%16][kk = k%4][e = (k%128)/16] -------------------------------------
    for (int NCH = 1; NCH <= 2; NCH++)
        for (int K : {128, 384, 4096})
            for (int N : {16, 48, 100}) {                    // 100: the last tile is padded to 16 rows
                const size_t total = m16_elems(N, K, NCH);
                CHECK(total == (size_t)((N + 15) / 16) * 16 * NCH * K, "m16_elems");
                std::vector<char> seen(total, 0);
                const int Np = (N + 15) / 16 * 16;
                for (int n = 0; n < Np; n++)
                    for (int c = 0; c < NCH; c++)
                        for (int k = 0; k < K; k++) {
                            const size_t ix = m16_index(n, k, c, K, NCH);
                            CHECK(ix < total, "m16_index out of range n=%d k=%d", n, k);
                            if (ix < total) { CHECK(!seen[ix], "m16_index not injective n=%d k=%d c=%d", n, k, c); seen[ix] = 1; }
                        }
                // a 16-byte unit = the eight k of one row with the same k % 16 (e = 0..7 contiguous); a wave-wide load of unit (C, m) of a
                // tile-chain is 1 KiB contiguous and lane (i, kk) finds its eight elements at byte offset ((i * 4 + kk) * 16) inside it
                for (int t = 0; t < Np / 16; t++)
                    for (int c = 0; c < NCH; c++)
                        for (int C = 0; C < K / 128; C++)
                            for (int m = 0; m < 4; m++) {
                                const size_t unit0 = m16_index(16 * t, 128 * C + 4 * m, c, K, NCH);
                                CHECK(unit0 % 512 == 0, "unit (C, m) not 1 KiB aligned");
                                CHECK(unit0 == ((((size_t)t * NCH + c) * (K / 128) + C) * 4 + m) * 512, "chain-major order: chunk after chunk, m inside");
                                for (int i = 0; i < 16; i++)
                                    for (int kk = 0; kk < 4; kk++)
                                        for (int e = 0; e < 8; e++) {
                                            const int k = 128 * C + 16 * e + 4 * m + kk;            // k-group g = 4e + m: k = 128C + 4g + kk
                                            CHECK(m16_index(16 * t + i, k, c, K, NCH) == unit0 + (size_t)(i * 4 + kk) * 8 + e, "lane (i, kk) element e");
                                        }
                            }
            }
    // ---- xt: [chunk][m][kk][sequence 0..15][e] ---------------------------------------------------------------------------------------
    for (int K : {128, 896, 4096}) {
        std::vector<char> seen((size_t)16 * K, 0);
        for (int s = 0; s < 16; s++)
            for (int k = 0; k < K; k++) {
                const size_t ix = xt_index(s, k);
                CHECK(ix < (size_t)16 * K, "xt_index out of range");
                if (ix < (size_t)16 * K) { CHECK(!seen[ix], "xt_index not injective"); seen[ix] = 1; }
            }
        for (int C = 0; C < K / 128; C++)
            for (int m = 0; m < 4; m++)
                for (int kk = 0; kk < 4; kk++)
                    for (int s = 0; s < 16; s++)
                        for (int e = 0; e < 8; e++)
                            CHECK(xt_index(s, 128 * C + 16 * e + 4 * m + kk) == ((((size_t)C * 4 + m) * 4 + kk) * 16 + s) * 8 + e, "xt unit layout");
    }
    // ---- xt for more than 16 sequences (round 4, 17 .. 32 sequences): group s / 16 is a whole layout of its own, 16 * K elements further --------
    for (int K : {128, 4096}) {
        std::vector<char> seen((size_t)32 * K, 0);
        for (int s = 0; s < 32; s++)
            for (int k = 0; k < K; k++) {
                const size_t ix = xt_group(s, K) + xt_index(s & 15, k);
                CHECK(ix < (size_t)32 * K, "grouped xt index out of range");
                CHECK((ix >= (size_t)16 * K) == (s >= 16), "group 1 lives behind group 0");
                if (ix < (size_t)32 * K) { CHECK(!seen[ix], "grouped xt index not injective"); seen[ix] = 1; }
            }
    }
    // ---- tiled GEMV layout [N/RW][K/8][NCH][RW][8] (RW 4: the row-broadcast layout [N/4][K/128][row%4][k%16][(k%128)/16], one chain): a
    // bijection into tiled_elems for every row-block width the library uses; a lane's 16-byte piece = eight k of its own row ---------------
    for (int RW : {4, 16, 28, 32, 56, 64})
        for (int NCH = 1; NCH <= (RW == 4 ? 1 : 2); NCH++) {
            const int K = 256, N = 3 * RW + (RW == 4 ? 1 : 0);                       // (13 rows at RW 4: padded to whole 16-row workgroups)
            const size_t total = tiled_elems(N, K, RW, NCH);
            std::vector<char> seen(total, 0);
            for (int n = 0; n < N; n++)
                for (int c = 0; c < NCH; c++)
                    for (int k = 0; k < K; k++) {
                        const size_t ix = tiled_index(n, k, c, K, RW, NCH);
                        CHECK(ix < total, "tiled_index out of range RW=%d", RW);
                        if (ix < total) { CHECK(!seen[ix], "tiled_index not injective RW=%d", RW); seen[ix] = 1; }
                    }
            const int n5 = 5 % N;
            if (RW == 4) {                                   // the eight k congruent modulo 16 inside a 128-step chunk, ascending
                for (int k0 = 0; k0 < K; k0 += 128)
                    for (int j = 0; j < 16; j++)
                        for (int e = 1; e < 8; e++) CHECK(tiled_index(n5, k0 + j + 16 * e, 0, K, RW, NCH) == tiled_index(n5, k0 + j, 0, K, RW, NCH) + e, "row-broadcast unit");
            } else {
                for (int k8 = 0; k8 < K / 8; k8++)           // eight consecutive k
                    for (int j = 1; j < 8; j++) CHECK(tiled_index(n5, 8 * k8 + j, 0, K, RW, NCH) == tiled_index(n5, 8 * k8, 0, K, RW, NCH) + j, "8 consecutive k contiguous");
            }
        }
    // ---- launch plan of the streaming product: the choices measured on the 8B shapes (profiles/r03_gemmstream_bench.log) ---------------------
    {
        const int cus = 256;
        struct { int n_rows, nch; } shp[4] = {{6144, 1}, {4096, 1}, {14336, 2}, {4096, 1}};      // wq|wk|wv, wo, gate|up pairs, w2
        for (int S : {16, 17, 64, 128, 256, 512, 1024, 4096})
            for (auto sh : shp) {
                const int nt = (sh.n_rows + 15) / 16, ct = (S + 15) / 16, ntw = lnb_gemm_stream_ntw(nt, ct, sh.nch, cus);
                CHECK(ntw == 1 || ntw == 2 || ntw == 4, "ntw %d", ntw);
                CHECK(ntw <= (ct >= 4 ? 4 : ct >= 2 ? 2 : 1), "more batch tiles per wave than the call has: S=%d ntw=%d", S, ntw);
                if (S >= 512) CHECK(ntw == 4, "S=%d rows=%d: four batch tiles per wave", S, sh.n_rows);
                if (S == 128) CHECK(ntw == (sh.nch == 2 ? 4 : 1), "S=128 rows=%d nch=%d: ntw %d", sh.n_rows, sh.nch, ntw);
                if (S == 256) CHECK(ntw == (sh.nch == 2 ? 4 : 2), "S=256 rows=%d nch=%d: ntw %d", sh.n_rows, sh.nch, ntw);
                const int groups = (S + 16 * ntw - 1) / (16 * ntw);
                CHECK(lnb_gemm_stream_rows_fastest(groups) == (S >= 1024 ? 1 : 0), "dispatch order at S=%d (%d row groups)", S, groups);
            }
    }
    CHECK(LNB_BATCH_MAX >= LNB_STREAM_COLS && LNB_STREAM_COLS == 16, "batch limits");
    printf(fails ? "layout_test: %d FAILURES\n" : "layout_test: ok\n", fails);
    return fails ? 1 : 0;
}
