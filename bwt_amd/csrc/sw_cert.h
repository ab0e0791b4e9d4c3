// sw_cert.h -- a certificate that sw:sw/2 (src/sw.erl:47-160) answers {2 Lq - 3 m, "<Lq>M"} on a SQUARE window whose only
// full diagonal has m mismatches, decided on bit masks without filling the table.  Plain C++ on purpose: the kernel
// (k_sw_cert, kernels.hip.h) and the CPU check against the oracle's full table (tests/cert_check/) compile THIS text.
//
// Setting.  Rows i = 1..L (read bases), columns j = 1..L (window bases), diagonal o = j - i; the main diagonal (o = 0) scores
// d = 2 L - 3 m.  By DESIGN.md 4b step 1 every value of the reference's table T is the affine score (match 2, mismatch -1, gap
// open -2, extend -1) of a real path from row 0 to its cell, and T(i,i) >= the main diagonal's prefix score.  So if
//   (I)  every path into (L, L) other than the main diagonal scores <= d, and
//   (II) every path into a last-row cell (L, j), j < L, scores < d,
// then T(L,L) = d is the leftmost maximum of the last row, every T(i,i) equals its prefix score through its up_left candidate
// (a larger T(i,i) would carry through to T(L,L) > d), ties prefer up_left (sw.erl:90), and the traceback is the diagonal.
//
// Gain of a path P against the main diagonal.  Every row is consumed by one diagonal move or one up move (insertion); left
// moves (deletions) consume columns.  With s_i = 2 / -1 the main diagonal's score in row i:
//   a diagonal move on diagonal o != 0:  +3 where the main diagonal mismatches and o matches ("save"), -3 where the main diagonal
//                                        matches and o does not ("bad"), 0 otherwise;
//   an up move:  -1 - s_i = 0 in a mismatch row ("free"), -3 in a match row;  a left move: -1;  every gap run: -1 more.
//   gain(P) = 3 (saves - bads) - runs - 3 (up moves in match rows) - left moves.
// P starts on a diagonal >= 0 and ends on one <= 0; up moves step one diagonal down, left moves one up; P != main has >= 1 up move.
//
// Outside the band.  A path that touches a diagonal |o| > X makes u >= X + 1 up moves; f of them free, each on a mismatch row of
// its own that is then not saved: gain <= 3 (m - f) - 3 (u - f) - 1 <= 3 (m - X - 1) - 1 < 0 when m <= X + 1.
//
// Inside the band: an upper bound by a coarse DP.  The mismatch rows p_1 < ... < p_m cut the rows into m + 1 stretches of
// match rows.  State: g[o] >= the gain of any partial path that is on diagonal o (gu[o]: ... and whose last move was an up move).
//   stretch (n rows, b[o] of them bad on o):  staying on o costs 3 b[o] -- or 6 at most, the least a detour (one left run, one
//     up run over match rows) costs; arriving from a lower diagonal costs >= 1 + the distance (one left run), from a higher one
//     >= 3 per step (up moves in match rows; the run's own -1 is charged where a free up move opens it, or not at all: lower bounds);
//   mismatch row p:  a diagonal move on o gains 3 where o matches; a free up move from o + 1 costs 1 if it opens a run, 0 if it
//     extends one that stood at the row before (gu: after a mismatch row the paths that made an up move there; after a stretch
//     whatever arrived by up moves inside it -- a superset of the runs that reach its last row).
// Every real path inside the band maps to a coarse path with at least its gain (each coarse charge is a lower bound of what the real
// moves cost; cells outside the window are allowed: more paths, not fewer).  The certificate holds when, after the last stretch,
// g[0] <= 0 and g[o] <= -1 for every o < 0 -- and 2 L - 3 m stays above the abort threshold of sw.erl:62,69.
#pragma once
#include <stdint.h>

#ifndef SWC_HD
#ifdef __HIPCC__
#define SWC_HD __host__ __device__ __forceinline__
#else
#define SWC_HD static inline
#endif
#endif

// bits [sh, sh + 32) of the 64-bit value hi:lo, 0 <= sh < 32
SWC_HD uint32_t swc_funnel(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
SWC_HD int swc_max(int a, int b) { return a > b ? a : b; }
SWC_HD int swc_min(int a, int b) { return a < b ? a : b; }

#define SWC_NEG (-100)

// X: diagonals -X..X are looked at; NW: 32-bit words per read plane (L <= 32 NW).
// qlo / qhi: the read's code planes, bit b of word w = base 32 w + b.   rlo / rhi: the window's, word t = the 32 reference bases
// from window start + 32 (t - 1): word 0 lies in front of the window, word NW + 1 may lie behind it.
// -> true: the answer is {2 L - 3 m, "<L>M"}; m_out = mismatches on the main diagonal either way.
// g_out (optional, 2 X + 1 ints): the final bounds, for the CPU check.
template <int X, int NW>
SWC_HD bool sw_diag_cert(const uint32_t (&qlo)[NW], const uint32_t (&qhi)[NW], const uint32_t (&rlo)[NW + 2],
                         const uint32_t (&rhi)[NW + 2], int L, int &m_out, int *g_out = nullptr) {
    constexpr int ND = 2 * X + 1;
    constexpr int NQ = (NW + 1) / 2;      // the masks as 64-bit words: the loop below runs per word and, on the GPU, as long as the lane of
                                          // its wave with the most mismatch rows in that word -- half as many words, fewer idle turns
    uint64_t E[ND][NQ];                   // E[X + o][q]: rows where diagonal o matches
    uint64_t vm[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int left = L - 64 * q;
        vm[q] = left >= 64 ? ~0ull : left <= 0 ? 0ull : (1ull << left) - 1ull;
    }
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const int o = d - X;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            uint64_t e = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int w = 2 * q + h;
                if (w >= NW) continue;
                uint32_t xl, xh;
                if (o == 0) { xl = rlo[w + 1]; xh = rhi[w + 1]; }
                else if (o > 0) { xl = swc_funnel(rlo[w + 2], rlo[w + 1], (uint32_t)o); xh = swc_funnel(rhi[w + 2], rhi[w + 1], (uint32_t)o); }
                else { xl = swc_funnel(rlo[w + 1], rlo[w], (uint32_t)(32 + o)); xh = swc_funnel(rhi[w + 1], rhi[w], (uint32_t)(32 + o)); }
                e |= (uint64_t)(uint32_t)~((qlo[w] ^ xl) | (qhi[w] ^ xh)) << (32 * h);
            }
            E[d][q] = e;
        }
    }
    int m = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) m += __builtin_popcountll(~E[X][q] & vm[q]);
    m_out = m;
    if (m > X + 1 || 3 * m > 2 * (L / 2)) return false;

    int g[ND], gu[ND], acc[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) { g[d] = d >= X ? 0 : SWC_NEG; gu[d] = SWC_NEG; acc[d] = 0; }
    int nrows = 0;

    auto stretch = [&]() {
        int lf[ND], up[ND];
        lf[0] = SWC_NEG;
#pragma unroll
        for (int d = 1; d < ND; d++) lf[d] = swc_max(lf[d - 1] - 1, g[d - 1] - 2);
        up[ND - 1] = SWC_NEG;
#pragma unroll
        for (int d = ND - 2; d >= 0; d--) up[d] = swc_max(up[d + 1], g[d + 1]) - 3;
        const bool some = nrows > 0;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const int stay = d == X ? g[d] : g[d] - swc_min(3 * acc[d], 6);
            g[d] = swc_max(stay, swc_max(lf[d], up[d]));
            if (some) gu[d] = up[d];          // an up run that reaches the stretch's last row may go on, unopened, in the mismatch row behind it
            acc[d] = 0;
        }
        nrows = 0;
    };

#pragma unroll
    for (int q = 0; q < NQ; q++) {
        uint64_t rem = vm[q];
        uint64_t mw = ~E[X][q] & vm[q];
        while (mw) {
            const uint32_t b = (uint32_t)__builtin_ctzll(mw);
            mw &= mw - 1ull;
            const uint64_t seg = rem & ((1ull << b) - 1ull);
#pragma unroll
            for (int d = 0; d < ND; d++)
                if (d != X) acc[d] += __builtin_popcountll(~E[d][q] & seg);
            nrows += __builtin_popcountll(seg);
            stretch();
            // the mismatch row itself
#pragma unroll
            for (int d = 0; d < ND; d++) {
                const int t = d + 1 < ND ? swc_max(g[d + 1] - 1, gu[d + 1]) : SWC_NEG;
                const int sv = d == X ? 0 : (int)((E[d][q] >> b) & 1ull) * 3;
                g[d] = swc_max(g[d] + sv, t);
                gu[d] = t;
            }
            rem &= ~((2ull << b) - 1ull);
        }
#pragma unroll
        for (int d = 0; d < ND; d++)
            if (d != X) acc[d] += __builtin_popcountll(~E[d][q] & rem);
        nrows += __builtin_popcountll(rem);
    }
    stretch();
    bool ok = g[X] <= 0;
#pragma unroll
    for (int d = 0; d < X; d++) ok = ok && g[d] <= -1;
    if (g_out) {
#pragma unroll
        for (int d = 0; d < ND; d++) g_out[d] = g[d];
    }
    return ok;
}
