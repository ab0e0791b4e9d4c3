// TEST INFRASTRUCTURE.  Checks bwt_amd/csrc/sw_cert.h -- the very text the kernel k_sw_cert compiles -- on the CPU:
//   (1) whenever the certificate holds, the oracle's full table (oracle/bwt_oracle.c: orc_sw, the restatement of src/sw.erl:47-160)
//       answers {2 L - 3 m, "<L>M"};
//   (2) its per-diagonal bounds g[o] are upper bounds: an exact three-state affine DP over the real L x L table gives the best
//       score A(L, L + o) of any path into each last-row cell, and max(g[o], 3 (m - X - 1) - 1) >= A(L, L + o) - (2 L - 3 m) must
//       hold whether or not the certificate does.
// Instances lean towards what could break it: mismatches in clumps, at the read's ends, in adjacent rows; two-letter and periodic
// sequences (neighbouring diagonals match often); reads of 18..160 bases.
//   cert_check <cases> <seed> [plain]  ->  one line of counts, exit code 1 on the first violation.
//   plain = 1: reads of 100 bases from an i.i.d. sequence with 3..6 (X = 5), 3..4 (X = 3), 3..8 (X = 7), 7..10 (X = 9) substitutions at random places -- what
//   the certificate's share of plain-diagonal answers is on ordinary candidates.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>
#include <random>

#include "../../bwt_amd/csrc/sw_cert.h"

extern "C" int orc_sw(const uint8_t *q, int lq, const uint8_t *ref, int w, int *score, char *cigar, int cap);

static const char ALPHA[4] = {'A', 'C', 'G', 'T'};
static int code_of(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

// planes of seq[from, from + 32 * nwords): bases outside [0, n) read as 'A'
static void planes(const std::vector<uint8_t> &seq, long from, int nwords, uint32_t *lo, uint32_t *hi) {
    for (int w = 0; w < nwords; w++) {
        lo[w] = hi[w] = 0;
        for (int b = 0; b < 32; b++) {
            const long k = from + 32L * w + b;
            const int c = (k >= 0 && k < (long)seq.size()) ? code_of(seq[k]) : 0;
            lo[w] |= (uint32_t)(c & 1) << b;
            hi[w] |= (uint32_t)(c >> 1) << b;
        }
    }
}

// best affine score of any path from row 0 into (L, j), j = 0..L (free start column, column 0 = up moves from (0,0))
static void exact_last_row(const uint8_t *q, const uint8_t *r, int L, std::vector<int> &out) {
    const int NEG = -1000000;
    std::vector<int> H(L + 1, 0), E(L + 1, NEG), Hn(L + 1), En(L + 1);
    for (int i = 1; i <= L; i++) {
        En[0] = std::max(H[0] - 2, E[0] - 1); Hn[0] = En[0];
        int F = NEG;
        for (int j = 1; j <= L; j++) {
            En[j] = std::max(H[j] - 2, E[j] - 1);
            F = std::max(Hn[j - 1] - 2, F - 1);
            const int d = H[j - 1] + (q[i - 1] == r[j - 1] ? 2 : -1);
            Hn[j] = std::max(d, std::max(En[j], F));
        }
        H.swap(Hn); E.swap(En);
    }
    out = H;
}

template <int X, int NW>
static int run_case(const std::vector<uint8_t> &q, const std::vector<uint8_t> &ref, long wstart, int L, long counts[4]) {
    uint32_t qlo[NW], qhi[NW], rlo[NW + 2], rhi[NW + 2];
    planes(q, 0, NW, qlo, qhi);
    planes(ref, wstart - 32, NW + 2, rlo, rhi);
    int m = 0, g[2 * X + 1];
    const bool ok = sw_diag_cert<X, NW>(qlo, qhi, rlo, rhi, L, m, g);
    int mm = 0;
    for (int i = 0; i < L; i++) mm += q[i] != ref[wstart + i];
    if (mm != m) { fprintf(stderr, "mismatch count %d vs %d\n", m, mm); return 1; }
    counts[0]++;
    if (m <= X + 1 && 3 * m <= 2 * (L / 2)) {
        counts[1]++;
        std::vector<int> A;
        exact_last_row(q.data(), ref.data() + wstart, L, A);
        const int d = 2 * L - 3 * m;
        for (int o = -X; o <= 0; o++) {
            if (L + o < 0) continue;
            const int gain = A[L + o] - d;
            const int bound = std::max(g[X + o], 3 * (m - X - 1) - 1);
            if (bound < gain) {
                fprintf(stderr, "BOUND VIOLATED  X=%d L=%d m=%d o=%d g=%d exact gain=%d\nq=%.*s\nr=%.*s\n", X, L, m, o, g[X + o], gain, L,
                        (const char *)q.data(), L, (const char *)ref.data() + wstart);
                return 1;
            }
        }
    }
    if (ok) {
        counts[2]++;
        int score = 0;
        char cig[4096];
        const int rc = orc_sw(q.data(), L, ref.data() + wstart, L, &score, cig, sizeof cig);
        char want[32];
        snprintf(want, sizeof want, "%dM", L);
        if (rc != 0 || score != 2 * L - 3 * m || strcmp(cig, want) != 0) {
            fprintf(stderr, "CERTIFICATE WRONG  X=%d L=%d m=%d: oracle rc=%d score=%d cigar=%s\nq=%.*s\nr=%.*s\n", X, L, m, rc, score, cig, L,
                    (const char *)q.data(), L, (const char *)ref.data() + wstart);
            return 1;
        }
    } else if (m >= 1 && m <= X + 1) {
        // how often the certificate fails although the answer is the plain diagonal (its price, not an error)
        int score = 0;
        char cig[4096], want[32];
        snprintf(want, sizeof want, "%dM", L);
        if (orc_sw(q.data(), L, ref.data() + wstart, L, &score, cig, sizeof cig) == 0 && score == 2 * L - 3 * m && !strcmp(cig, want)) counts[3]++;
    }
    return 0;
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    const bool plain = argc > 3 && atoi(argv[3]) != 0;
    std::mt19937_64 rng(seed);
    auto U = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    long counts[6][4] = {};      // per template: cases, bound checks, certified, plain-diagonal answers the certificate missed
    for (long it = 0; it < cases; it++) {
        const int kind = U(0, 9);
        const int L = plain ? 100 : (it % 7 == 0) ? U(18, 160) : (it % 3 == 0 ? 100 : U(24, 128));
        const int margin = 40;
        std::vector<uint8_t> ref(L + 2 * margin);
        // the reference stretch: random / two-letter / periodic with a few point changes
        const int style = plain ? 0 : U(0, 5);
        if (style <= 2) for (auto &c : ref) c = ALPHA[U(0, 3)];
        else if (style == 3) for (auto &c : ref) c = ALPHA[U(0, 9) < 9 ? U(0, 1) * 3 : U(0, 3)];     // AT-rich
        else {
            const int per = U(1, 9);
            std::vector<uint8_t> unit(per);
            for (auto &c : unit) c = ALPHA[U(0, 3)];
            for (size_t k = 0; k < ref.size(); k++) ref[k] = unit[k % per];
            const int chg = U(0, 6);
            for (int k = 0; k < chg; k++) ref[U(0, (int)ref.size() - 1)] = ALPHA[U(0, 3)];
        }
        std::vector<uint8_t> q(ref.begin() + margin, ref.begin() + margin + L);
        // mismatches: how many and where
        const int m = plain ? (it % 4 == 0 ? U(3, 4) : it % 4 == 1 ? U(3, 8) : it % 4 == 3 ? U(7, 10) : U(3, 6)) : kind == 9 ? U(0, 14) : U(1, 10);
        std::vector<int> posn;
        const int place = plain ? 0 : U(0, 5);
        int anchor = place == 1 ? 0 : place == 2 ? L - 1 : U(0, L - 1);
        for (int k = 0; k < m; k++) {
            int p;
            if (place == 0) p = U(0, L - 1);
            else if (place <= 3) p = anchor + (place == 2 ? -1 : 1) * U(0, 2 * m + 2);             // a clump at the start / end / somewhere
            else if (place == 4) p = anchor + k * U(1, 2);                                            // adjacent and next-but-one rows
            else p = (k & 1) ? U(0, std::min(L - 1, 6)) : L - 1 - U(0, std::min(L - 1, 6));           // both ends
            p = std::max(0, std::min(L - 1, p));
            posn.push_back(p);
        }
        for (int p : posn) {
            // half of the time the new base is what a neighbouring diagonal holds there (so that some other diagonal matches)
            uint8_t nb = ALPHA[(code_of(q[p]) + U(1, 3)) & 3];
            if (!plain && U(0, 1)) { const int o = U(-3, 3); const uint8_t cand = ref[margin + p + o]; if (cand != q[p]) nb = cand; }
            q[p] = nb;
        }
        int rc = 0;
        if (L <= 128) {
            rc |= run_case<5, 4>(q, ref, margin, L, counts[0]);
            if (it % 4 == 0) rc |= run_case<3, 4>(q, ref, margin, L, counts[1]);
            if (it % 4 == 1) rc |= run_case<7, 4>(q, ref, margin, L, counts[2]);
        }
        if (it % 4 == 2 || L > 128) rc |= run_case<5, 5>(q, ref, margin, L, counts[3]);
        if (L <= 128 && it % 2 == 1) rc |= run_case<9, 4>(q, ref, margin, L, counts[4]);
        if (it % 4 == 3 || (L > 128 && it % 2 == 0)) rc |= run_case<9, 5>(q, ref, margin, L, counts[5]);
        if (rc) return 1;
    }
    const char *names[6] = {"X5_NW4", "X3_NW4", "X7_NW4", "X5_NW5", "X9_NW4", "X9_NW5"};
    printf("{");
    for (int t = 0; t < 6; t++)
        printf("%s\"%s\": {\"cases\": %ld, \"bound_checks\": %ld, \"certified\": %ld, \"plain_but_not_certified\": %ld}", t ? ", " : "", names[t],
               counts[t][0], counts[t][1], counts[t][2], counts[t][3]);
    printf("}\n");
    return 0;
}
