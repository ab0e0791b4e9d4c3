/* TEST INFRASTRUCTURE ONLY (oracle/): plain-C CPU restatement of the reference's
 * FM-index seed search, seed clustering and Smith-Waterman extension.
 *
 * PARITY UNPINNED: the reference is Erlang; no Erlang runtime exists where this was
 * written and the reference holds no test that records a result for this path
 * (SURVEY.md 8c).  This is a restatement from reading the source, checked against
 * oracle/erl_literal.py (a list-for-list Python restatement) by tests/.
 *
 * "Literal mode": se_down / se_up are the reference's linear scans over the L and I
 * columns, the SW fills the full Lq x (W+1) rectangle with the per-row abort test.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Reference lines followed (relative to /root/reference):
 *   include/bwt.hrl:4-7,14-18      constants
 *   src/bwa.erl:12-45              find_seeds/6, find_seeds/5
 *   src/fmi.erl:25-39,89-100       se_down, se_up, get_seed_ends
 *   src/sga.erl:17-76              sga loop, add_seeds, get_similar, skip_Ns
 *   src/sw.erl:47-202              sw, build_tab, add_row2tab, get_CIGAR, sigma
 *   src/worker_bwt.erl:54-55,84-97 window slicing and the per-read fold
 *   src/seeds.erl:11-85            generate_fs patterns and the generated fs:find_seeds/1
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>

#define MIN_LEN 8
#define MAX_RANGE 45
#define MAX_LEN 15
#define TOLERANCE 20
#define REF_EXT 200

typedef struct {
    uint32_t n;            /* rows */
    const uint8_t *L;      /* row r (1-based) at [r-1] */
    const uint32_t *I;
    const uint32_t *SA;
    uint32_t Pc, Pg, Pt, Last;
    int32_t shift;
} orc_fm;

enum { ORC_SEEDS = 0, ORC_NO_SEEDS = 1, ORC_TOO_MANY = 2, ORC_CRASH = -1, ORC_NOMEM = -2 };

typedef struct { int64_t *v; size_t n, cap; } vec64;
static int vpush(vec64 *a, int64_t x) {
    if (a->n == a->cap) {
        size_t nc = a->cap ? a->cap * 2 : 64;
        int64_t *p = (int64_t *)realloc(a->v, nc * sizeof(int64_t));
        if (!p) return -1;
        a->v = p; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

/* fmi.erl:25-31.  returns 0 = not_found, else 1 and *p */
static int se_down(const orc_fm *fm, uint32_t sp, uint32_t ep, uint8_t c, uint32_t *p) {
    for (;;) {
        if (sp == ep) return 0;
        if (sp < 1 || sp > fm->n) return -1;
        if (fm->L[sp - 1] == c) { *p = fm->I[sp - 1]; return 1; }
        sp++;
    }
}
/* fmi.erl:35-39 */
static int se_up(const orc_fm *fm, uint32_t ep, uint8_t c, uint32_t *p) {
    for (;;) {
        if (ep < 1 || ep > fm->n) return -1;
        if (fm->L[ep - 1] == c) { *p = fm->I[ep - 1]; return 1; }
        ep--;
    }
}

/* fmi.erl:89-100: appends SA[Ep]+N, SA[Ep-1]+N, ..., SA[Sp]+N (that list order) + add */
static int get_seed_ends(const orc_fm *fm, uint32_t sp, uint32_t ep, int n, int64_t add, vec64 *out) {
    if (n < MIN_LEN) return ORC_NO_SEEDS;
    if (sp < 1 || ep > fm->n || sp > ep) return ORC_CRASH;
    for (uint32_t r = ep;; r--) {
        if (vpush(out, (int64_t)fm->SA[r - 1] + n + add)) return ORC_NOMEM;
        if (r == sp) break;
    }
    return ORC_SEEDS;
}

/* bwa.erl:12-45; sub[0..len) is the sub-read, consumed from the right.
 * On ORC_SEEDS the seed ends (+add) are appended to out in reference list order.
 * sp_out/ep_out/n_out report the final state (for the per-task parity tests). */
static int find_seeds(const orc_fm *fm, const uint8_t *sub, int len, int64_t add, vec64 *out,
                      uint32_t *sp_out, uint32_t *ep_out, int *n_out) {
    uint32_t sp, ep;
    uint8_t h = sub[len - 1];
    switch (h) {
    case 'A': case 'N': sp = 1; ep = fm->Pc - 1; break;
    case 'C': sp = fm->Pc; ep = fm->Pg - 1; break;
    case 'G': sp = fm->Pg; ep = fm->Pt - 1; break;
    case 'T': sp = fm->Pt; ep = fm->Last; break;
    default: return ORC_CRASH;                       /* case_clause */
    }
    int n = 1, k = len - 2, st;
    for (;;) {
        if (n == MAX_LEN) {                           /* bwa.erl:31-32 */
            st = (ep - sp > MAX_RANGE && ep > sp) ? ORC_TOO_MANY : get_seed_ends(fm, sp, ep, n, add, out);
            break;
        }
        if (k >= 0 && sub[k] == 'N') {                /* bwa.erl:33-34 */
            st = (ep - sp > MAX_RANGE && ep > sp) ? ORC_TOO_MANY : get_seed_ends(fm, sp, ep, n, add, out);
            break;
        }
        if (k >= 0) {                                 /* bwa.erl:36-43 */
            uint32_t sp1, ep1;
            int f = se_down(fm, sp, ep, sub[k], &sp1);
            if (f < 0) return ORC_CRASH;
            if (f == 0) { st = get_seed_ends(fm, sp, ep, n, add, out); break; }
            if (se_up(fm, ep, sub[k], &ep1) < 0) return ORC_CRASH;
            sp = sp1; ep = ep1; n++; k--;
            continue;
        }
        st = (ep - sp > MAX_RANGE && ep > sp) ? ORC_TOO_MANY : get_seed_ends(fm, sp, ep, n, add, out);
        break;
    }
    if (sp_out) { *sp_out = sp; *ep_out = ep; *n_out = n; }
    return st;
}

int orc_find_seeds(const orc_fm *fm, const uint8_t *sub, int len, int64_t *out, int cap, int *n_out_seeds,
                   uint32_t *sp, uint32_t *ep, int *n) {
    vec64 v = {0, 0, 0};
    int st = find_seeds(fm, sub, len, 0, &v, sp, ep, n);
    *n_out_seeds = (int)v.n;
    if (st == ORC_SEEDS) {
        if ((int)v.n > cap) { free(v.v); return ORC_NOMEM; }
        memcpy(out, v.v, v.n * sizeof(int64_t));
    }
    free(v.v);
    return st;
}

/* sga.erl:69-76.  q[0..len): returns -1 = no_more_subseqs, else Skip (new length = len - Skip) */
static int skip_Ns(const uint8_t *q, int len) {
    if (len < MIN_LEN) return -1;
    int j = 0, skipped = 0, k = len - 1;
    for (;;) {
        if (j == MIN_LEN) return skipped;
        if (k < 0) return -1;
        if (q[k] == 'N') { skipped += j; j = 1; } else j++;
        k--;
    }
}

static int cmp64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

/* sga.erl:17-64.  Returns number of clusters (>=0), or <0 on crash/overflow.
 * S/D are written in the reference's output order (descending position).
 * stats (optional, 3 x int64): LF steps taken is not tracked here; [0]=Qty [1]=#seeds. */
int orc_sga(const orc_fm *fm, const uint8_t *q, int lq, int64_t *S, int32_t *D, int cap, int64_t *stats) {
    vec64 acc = {0, 0, 0};
    int qty = 0, len = lq;
    int64_t shift = 0;
    for (;;) {
        int skip = skip_Ns(q, len);
        if (skip < 0) break;
        int l1 = len - skip;
        int st = find_seeds(fm, q, l1, shift + skip, &acc, 0, 0, 0);
        if (st < 0) { free(acc.v); return st == ORC_NOMEM ? -2 : -1; }
        if (st == ORC_SEEDS) qty++;
        len = l1 - MIN_LEN;
        shift += MIN_LEN + skip;
    }
    if (stats) { stats[0] = qty; stats[1] = (int64_t)acc.n; }
    int nout = 0;
    if (qty >= 2) {
        qsort(acc.v, acc.n, sizeof(int64_t), cmp64);
        int nthr = 2 + qty / 7;
        /* clusters found ascending; the reference prepends, so reverse at the end */
        int64_t p1 = acc.v[0], dist = 0;
        int count = 1;
        for (size_t i = 1; i <= acc.n; i++) {
            int end = (i == acc.n);
            if (!end && acc.v[i] - p1 <= TOLERANCE) {
                count++; dist += acc.v[i] - p1; p1 = acc.v[i];
                continue;
            }
            if (count >= nthr) {
                if (nout >= cap) { free(acc.v); return -2; }
                S[nout] = p1; D[nout] = (int32_t)dist; nout++;
            }
            if (!end) { p1 = acc.v[i]; count = 1; dist = 0; }
        }
        for (int a = 0, b = nout - 1; a < b; a++, b--) {
            int64_t ts = S[a]; S[a] = S[b]; S[b] = ts;
            int32_t td = D[a]; D[a] = D[b]; D[b] = td;
        }
    }
    free(acc.v);
    return nout;
}

/* sw.erl:164-202 */
static double sigma(uint8_t s, uint8_t r) {
    if (s == 'N' && r == 'N') return 1.6;
    if (s == r) return 2;
    if (s == 'N' || r == 'N') return 1.6;
    switch (r) {
    case 'R': return (s == 'A' || s == 'G') ? 2 : -1;
    case 'Y': return (s == 'C' || s == 'T') ? 2 : -1;
    case 'B': return s == 'A' ? -1 : 2;
    case 'D': return s == 'C' ? -1 : 2;
    case 'K': return (s == 'G' || s == 'T') ? 2 : -1;
    case 'M': return (s == 'A' || s == 'C') ? 2 : -1;
    case 'S': return (s == 'G' || s == 'C') ? 2 : -1;
    case 'W': return (s == 'A' || s == 'T') ? 2 : -1;
    case 'V': return s == 'T' ? -1 : 2;
    default: return -1;
    }
}

enum { D_LEFT = 0, D_UP = 1, D_UL = 2, D_UNDEF = 3 };   /* term order left < up < up_left */

/* sw.erl:47-160.  returns 0 ok, 1 no_match, <0 error.  cigar is NUL-terminated text. */
int orc_sw(const uint8_t *q, int lq, const uint8_t *ref, int w, int *score, char *cigar, int cap) {
    if (lq <= 0 || w <= 0) return -1;
    int W1 = w + 1;
    double *prev = (double *)malloc(sizeof(double) * W1 * 2);
    uint8_t *dir = (uint8_t *)malloc((size_t)(lq + 1) * W1);
    if (!prev || !dir) { free(prev); free(dir); return -2; }
    double *cur = prev + W1;
    for (int j = 0; j < W1; j++) { prev[j] = 0; dir[j] = D_UNDEF; }
    double vthr = -2.0 * (lq / 2), vmax = 0;
    int rc = 0;
    for (int i = 1; i <= lq; i++) {
        if (vmax < vthr) { rc = 1; goto done; }               /* sw.erl:69 */
        uint8_t *dp = dir + (size_t)(i - 1) * W1, *dc = dir + (size_t)i * W1;
        cur[0] = prev[0] + (i == 1 ? -2 : -1);                /* sw.erl:73-76 */
        dc[0] = D_UP;
        vmax = cur[0];
        for (int j = 1; j < W1; j++) {
            double d = prev[j - 1] + sigma(q[i - 1], ref[j - 1]);
            double l = cur[j - 1] + (dc[j - 1] == D_LEFT ? -1 : -2);
            double u = prev[j] + (dp[j] == D_UP ? -1 : -2);
            /* lists:max of {V,Dir}: larger V, ties by atom order up_left > up > left */
            double v = d; uint8_t dd = D_UL;
            if (u > v) { v = u; dd = D_UP; }
            if (l > v) { v = l; dd = D_LEFT; }
            cur[j] = v; dc[j] = dd;
            if (v > vmax) vmax = v;                           /* erlang:max keeps first on == */
        }
        vthr += 2;
        double *t = prev; prev = cur; cur = t;
    }
    if (vmax < vthr) { rc = 1; goto done; }
    {
        /* get_starting_point: leftmost column of the last row equal to Vmax */
        int j = 0;
        while (prev[j] != vmax) j++;
        *score = (int)vmax;                                    /* trunc */
        /* traceback; runs collected backwards */
        int i = lq, nrun = 0, runcap = lq + w + 2;
        int *rl = (int *)malloc(sizeof(int) * runcap);
        uint8_t *ro = (uint8_t *)malloc(runcap);
        if (!rl || !ro) { free(rl); free(ro); rc = -2; goto done; }
        while (i > 0) {
            uint8_t d = dir[(size_t)i * W1 + j];
            if (nrun && ro[nrun - 1] == d) rl[nrun - 1]++;
            else { ro[nrun] = d; rl[nrun] = 1; nrun++; }
            if (d == D_UL) { i--; j--; }
            else if (d == D_UP) i--;
            else j--;
        }
        int p = 0;
        for (int k = nrun - 1; k >= 0; k--) {
            char tmp[16];
            int m = snprintf(tmp, sizeof tmp, "%d%c", rl[k], ro[k] == D_UL ? 'M' : ro[k] == D_UP ? 'I' : 'D');
            if (p + m + 1 > cap) { free(rl); free(ro); rc = -3; goto done; }
            memcpy(cigar + p, tmp, m); p += m;
        }
        cigar[p] = 0;
        free(rl); free(ro);
    }
done:
    free(prev < cur ? prev : cur);
    free(dir);
    return rc;
}

/* worker_bwt.erl:84-97 for one read.  ref1 = 200xN ++ .ref ++ 200xN.
 * Outputs in the reference's `Cigars` list order (ascending position):
 *   score[k], pos[k] (= S-Ref_len+Shift), win_off[k] (offset into ref1), win_len[k], cigar text
 *   packed NUL-separated into cigars (cap bytes).
 * returns #candidates kept (>=0); -1 read emits nothing (sga = []) ; <-1 error:
 *   -2 crash/bad base, -3 window out of range, -4 capacity */
int orc_align_read(const orc_fm *fm, const uint8_t *ref1, int64_t ref1_len, const uint8_t *q, int lq,
                   int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len,
                   char *cigars, int cig_cap, int cap, int64_t *cells) {
    /* room for 4096 clusters, and once more with 2^20 when a read has more (low-complexity reads on a repeat-rich genome have
     * tens of thousands: round 3's parity runs met them) */
    int maxc = 4096;
    int64_t *S = (int64_t *)malloc(sizeof(int64_t) * maxc);
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * maxc);
    int nc = orc_sga(fm, q, lq, S, D, maxc, 0);
    if (nc == -2) {
        free(S); free(D);
        maxc = 1 << 20;
        S = (int64_t *)malloc(sizeof(int64_t) * maxc);
        D = (int32_t *)malloc(sizeof(int32_t) * maxc);
        nc = orc_sga(fm, q, lq, S, D, maxc, 0);
    }
    if (nc <= 0) { free(S); free(D); return nc == 0 ? -1 : (nc == -1 ? -2 : -4); }
    int kept = 0, cp = 0;
    /* Seeds1 is in descending order; Cigars is built by prepending => process clusters in
     * list order, emit in reverse.  Collect then reverse. */
    int32_t *sc = (int32_t *)malloc(sizeof(int32_t) * nc);
    int *keep = (int *)malloc(sizeof(int) * nc);
    char **cg = (char **)calloc(nc, sizeof(char *));
    int rc = 0;
    for (int k = 0; k < nc; k++) {
        int64_t ref_len = lq + D[k];
        int64_t start = S[k] - ref_len + REF_EXT;
        if (start < 0 || start + ref_len > ref1_len) { rc = -3; break; }
        cg[k] = (char *)malloc(4 * (size_t)(lq + ref_len) + 16);
        if (cells) *cells += (int64_t)lq * ref_len;
        int r = orc_sw(q, lq, ref1 + start, (int)ref_len, &sc[k], cg[k], 4 * (int)(lq + ref_len) + 16);
        if (r < 0) { rc = -4; break; }
        keep[k] = (r == 0);
    }
    if (rc == 0) {
        for (int k = nc - 1; k >= 0; k--) {
            if (!keep[k]) continue;
            int64_t ref_len = lq + D[k];
            int m = (int)strlen(cg[k]);
            if (kept >= cap || cp + m + 1 > cig_cap) { rc = -4; break; }
            score[kept] = sc[k];
            pos[kept] = S[k] - ref_len + fm->shift;
            win_off[kept] = S[k] - ref_len + REF_EXT;
            win_len[kept] = (int32_t)ref_len;
            memcpy(cigars + cp, cg[k], m + 1); cp += m + 1;
            kept++;
        }
    }
    for (int k = 0; k < nc; k++) free(cg[k]);
    free(cg); free(keep); free(sc); free(S); free(D);
    return rc ? rc : kept;
}

/* ---- batch drivers for bench.py's cpu_baseline leg (read-parallel, static partition) ---- */
typedef struct {
    const orc_fm *fm; const uint8_t *ref1; int64_t ref1_len;
    const uint8_t *bases; const uint32_t *off; uint32_t lo, hi;
    int mode;                 /* 0 = sga only (search+cluster), 1 = full align */
    int64_t aligned, clusters, cells, checksum;
} job_t;

static void *job_run(void *p) {
    job_t *j = (job_t *)p;
    int CAP = 256, CIGCAP = 1 << 16;                    /* grown once for a read with more alignments (repeat-rich genomes) */
    int64_t S[4096]; int32_t D[4096];
    int32_t *score = (int32_t *)malloc(sizeof(int32_t) * CAP), *wlen = (int32_t *)malloc(sizeof(int32_t) * CAP);
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * CAP), *woff = (int64_t *)malloc(sizeof(int64_t) * CAP);
    char *cig = (char *)malloc(CIGCAP);
    for (uint32_t r = j->lo; r < j->hi; r++) {
        const uint8_t *q = j->bases + j->off[r];
        int lq = (int)(j->off[r + 1] - j->off[r]);
        if (j->mode == 0) {
            int nc = orc_sga(j->fm, q, lq, S, D, 4096, 0);
            if (nc > 0) {
                j->clusters += nc;
                for (int k = 0; k < nc; k++) j->checksum += S[k] * 31 + D[k];
            }
        } else {
            int64_t cells0 = j->cells;
            int n = orc_align_read(j->fm, j->ref1, j->ref1_len, q, lq, score, pos, woff, wlen,
                                   cig, CIGCAP, CAP, &j->cells);
            if (n == -4 && CAP < (1 << 20)) {
                free(score); free(wlen); free(pos); free(woff); free(cig);
                CAP = 1 << 20; CIGCAP = 1 << 26;
                score = (int32_t *)malloc(sizeof(int32_t) * CAP); wlen = (int32_t *)malloc(sizeof(int32_t) * CAP);
                pos = (int64_t *)malloc(sizeof(int64_t) * CAP); woff = (int64_t *)malloc(sizeof(int64_t) * CAP);
                cig = (char *)malloc(CIGCAP);
                j->cells = cells0;
                n = orc_align_read(j->fm, j->ref1, j->ref1_len, q, lq, score, pos, woff, wlen, cig, CIGCAP, CAP, &j->cells);
            }
            if (n > 0) {
                j->aligned++; j->clusters += n;
                for (int k = 0; k < n; k++) j->checksum += pos[k] * 31 + score[k];
            }
        }
    }
    free(cig); free(score); free(wlen); free(pos); free(woff);
    return 0;
}

/* out[0]=reads with >=1 alignment, out[1]=clusters/candidates, out[2]=SW cells, out[3]=checksum */
int orc_batch(const orc_fm *fm, const uint8_t *ref1, int64_t ref1_len, const uint8_t *bases,
              const uint32_t *off, uint32_t n_reads, int mode, int nthreads, int64_t *out) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = (job_t *)calloc(nthreads, sizeof(job_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].fm = fm; jobs[t].ref1 = ref1; jobs[t].ref1_len = ref1_len;
        jobs[t].bases = bases; jobs[t].off = off; jobs[t].mode = mode;
        jobs[t].lo = (uint32_t)((uint64_t)n_reads * t / nthreads);
        jobs[t].hi = (uint32_t)((uint64_t)n_reads * (t + 1) / nthreads);
        pthread_create(&th[t], 0, job_run, &jobs[t]);
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], 0);
        out[0] += jobs[t].aligned; out[1] += jobs[t].clusters;
        out[2] += jobs[t].cells; out[3] += jobs[t].checksum;
    }
    free(th); free(jobs);
    return 0;
}

/* Per-read digests of the complete answer (every hit's score, pos, window and CIGAR text, in list order), for at-scale
 * comparisons where aggregates are too coarse:
 *   digest(read) = sum over its hits h (k = 1-based rank in the read's list) of
 *                  k * (pos * 1000003 + score * 10007 + wlen * 101 + woff * 7 + sum_i text[i] * (i + 1)^2)      (mod 2^64)
 * reads without a hit list digest to 0, reads the reference would crash on to 1 (the C ABI reports them). */
typedef struct { const orc_fm *fm; const uint8_t *ref1; int64_t ref1_len; const uint8_t *bases; const uint32_t *off;
                 uint32_t lo, hi; uint64_t *dig; } djob_t;
static void *djob_run(void *p) {
    djob_t *j = (djob_t *)p;
    int CAP = 4096, CIGCAP = 1 << 20;                  /* (a read from a 300-copy repeat has 300 alignments; grown once for the reads that have more) */
    int32_t *score = (int32_t *)malloc(sizeof(int32_t) * CAP), *wlen = (int32_t *)malloc(sizeof(int32_t) * CAP);
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * CAP), *woff = (int64_t *)malloc(sizeof(int64_t) * CAP);
    char *cig = (char *)malloc(CIGCAP);
    for (uint32_t r = j->lo; r < j->hi; r++) {
        const uint8_t *q = j->bases + j->off[r];
        int lq = (int)(j->off[r + 1] - j->off[r]);
        int n = orc_align_read(j->fm, j->ref1, j->ref1_len, q, lq, score, pos, woff, wlen, cig, CIGCAP, CAP, 0);
        if (n == -4 && CAP < (1 << 20)) {               /* more than 4096 alignments: the buffers grow once, the read is done again */
            free(score); free(wlen); free(pos); free(woff); free(cig);
            CAP = 1 << 20; CIGCAP = 1 << 26;
            score = (int32_t *)malloc(sizeof(int32_t) * CAP); wlen = (int32_t *)malloc(sizeof(int32_t) * CAP);
            pos = (int64_t *)malloc(sizeof(int64_t) * CAP); woff = (int64_t *)malloc(sizeof(int64_t) * CAP);
            cig = (char *)malloc(CIGCAP);
            n = orc_align_read(j->fm, j->ref1, j->ref1_len, q, lq, score, pos, woff, wlen, cig, CIGCAP, CAP, 0);
        }
        uint64_t d = 0;
        if (n == -4) d = 2;                              /* the oracle's own capacity: not an answer (never equals a GPU digest) */
        else if (n < -1) d = 1;
        const char *t = cig;
        for (int k = 0; k < n; k++) {
            uint64_t th = 0;
            size_t len = strlen(t);
            for (size_t i = 0; i < len; i++) th += (uint64_t)(uint8_t)t[i] * (uint64_t)((i + 1) * (i + 1));
            d += (uint64_t)(k + 1) * ((uint64_t)pos[k] * 1000003ull + (uint64_t)(int64_t)score[k] * 10007ull + (uint64_t)wlen[k] * 101ull +
                                      (uint64_t)woff[k] * 7ull + th);
            t += len + 1;
        }
        j->dig[r] = d;
    }
    free(cig); free(score); free(wlen); free(pos); free(woff);
    return 0;
}
int orc_batch_digest(const orc_fm *fm, const uint8_t *ref1, int64_t ref1_len, const uint8_t *bases, const uint32_t *off,
                     uint32_t n_reads, int nthreads, uint64_t *dig) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    djob_t *jobs = (djob_t *)calloc(nthreads, sizeof(djob_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].fm = fm; jobs[t].ref1 = ref1; jobs[t].ref1_len = ref1_len; jobs[t].bases = bases; jobs[t].off = off; jobs[t].dig = dig;
        jobs[t].lo = (uint32_t)((uint64_t)n_reads * t / nthreads);
        jobs[t].hi = (uint32_t)((uint64_t)n_reads * (t + 1) / nthreads);
        pthread_create(&th[t], 0, djob_run, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
    return 0;
}

/* The I column from L: I[r] = C0[L[r]] + occ(L[r], r) - 1 with C0 = {$:0, A:1, C:Pc-1, G:Pg-1, T:Pt-1}
 * (bwt.erl:105-112 + fmi.erl:43-60: the K-th row whose L is c pops the K-th 0-based row whose F is c). */
int orc_make_I(const uint8_t *L, uint32_t n, uint32_t Pc, uint32_t Pg, uint32_t Pt, uint32_t *I) {
    uint64_t occ[256] = {0}, c0[256] = {0};
    c0['A'] = 1; c0['C'] = Pc - 1; c0['G'] = Pg - 1; c0['T'] = Pt - 1;
    for (uint32_t r = 0; r < n; r++) {
        uint8_t c = L[r];
        I[r] = (uint32_t)(c0[c] + occ[c]);
        occ[c]++;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * seeds:generate_fs/3 (src/seeds.erl:11-85) and the module `fs` it writes, restated literally:
 * the patterns are enumerated by the same depth-first recursion (gen_pattern/5, :51-62, in the order
 * mutation, insertion, deletion, match) and kept as explicit item strings, and fs:find_seeds/1 slides over the reference exactly as the generated
 * clauses do (seeds.erl:24-41): last window -> test once; a hit records N and jumps Len, a miss
 * advances by one; the accumulator is built by prepending (descending positions).
 * Pattern item: 0..255 = that byte, 256 = `_`. */
typedef struct { uint16_t *v; size_t n, cap; int len; int err; } patset;

/* acc[0..nacc) = the accumulator in the order its items were prepended (acc[nacc-1] is the head of the Erlang list) */
static void pat_emit(patset *ps, const uint16_t *acc, int nacc, const uint8_t *rest, int nrest) {
    /* seeds.erl:51-55: N = Len - length(Acc), Tail = sublist(SubQsec, N), Sequence = reverse(Tail) ++ Acc;
     * print_pattern/2 (:65-74) folds Sequence back into reading order = reverse(Acc) ++ Tail */
    int ntail = ps->len - nacc;
    if (ntail > nrest) { ps->err = 1; return; }            /* a shorter clause: fs.erl would not compile (head mismatch) */
    if (ps->n + (size_t)ps->len > ps->cap) {
        size_t nc = ps->cap ? ps->cap * 2 : 1024;
        while (nc < ps->n + (size_t)ps->len) nc *= 2;
        uint16_t *p = (uint16_t *)realloc(ps->v, nc * sizeof(uint16_t));
        if (!p) { ps->err = 2; return; }
        ps->v = p; ps->cap = nc;
    }
    uint16_t *o = ps->v + ps->n;
    for (int i = 0; i < nacc; i++) o[i] = acc[i];
    for (int i = 0; i < ntail; i++) o[nacc + i] = rest[i];
    ps->n += (size_t)ps->len;
}

static void gen_pattern(patset *ps, uint16_t *acc, int nacc, const uint8_t *rest, int nrest, int defects) {
    if (ps->err) return;
    if (defects == 0) {                                                     /* seeds.erl:51 */
        pat_emit(ps, acc, nacc, rest, nrest);
        return;
    }
    if (nrest > 0 && nacc < ps->len) {                                      /* seeds.erl:57 */
        acc[nacc] = 256; gen_pattern(ps, acc, nacc + 1, rest + 1, nrest - 1, defects - 1);   /* mutation  :58 */
        acc[nacc] = 256; gen_pattern(ps, acc, nacc + 1, rest, nrest, defects - 1);           /* insertion :59 */
        gen_pattern(ps, acc, nacc, rest + 1, nrest - 1, defects - 1);                        /* deletion  :60 */
        acc[nacc] = rest[0]; gen_pattern(ps, acc, nacc + 1, rest + 1, nrest - 1, defects);   /* match     :61 */
    }
    /* seeds.erl:62: anything else -> ok */
}

/* patterns of generate_fs(Qsec, Len, Defects) in generation order; returns the count, -1 = would not compile, -2 = nomem.
 * *out (count x len items) is malloc'ed */
long orc_fs_patterns(const uint8_t *qsec, int qlen, int len, int defects, uint16_t **out) {
    if (len < 1 || len > 255 || defects < 0) return -1;
    patset ps = {0, 0, 0, len, 0};
    uint16_t acc[512];
    int nsub = qlen < len + defects ? qlen : len + defects;                 /* lists:sublist(Qsec, Len+Defects) :43 */
    gen_pattern(&ps, acc, 0, qsec, nsub, defects);
    if (ps.err) { free(ps.v); return ps.err == 1 ? -1 : -2; }
    *out = ps.v;
    return (long)(ps.n / (size_t)len);
}
void orc_free(void *p) { free(p); }

static int fs_matches(const uint16_t *pats, long npat, int len, const uint8_t *w) {
    for (long k = 0; k < npat; k++) {                                       /* clause order; first true wins */
        const uint16_t *p = pats + (size_t)k * (size_t)len;
        int ok = 1;
        for (int i = 0; i < len && ok; i++) ok = p[i] == 256 || p[i] == w[i];
        if (ok) return 1;
    }
    return 0;                                                               /* the final matches(_,...,_) -> false */
}

/* fs:find_seeds/1 on ref[0..n): 1-based positions, newest first.  returns the count (may exceed cap; only cap written) */
long orc_fs_find_seeds(const uint16_t *pats, long npat, int len, const uint8_t *ref, long n, int64_t *out, long cap) {
    long cnt = 0, i = 0;
    int64_t N = 1;
    /* collect ascending, reverse at the end (= prepending) */
    while (n - i >= len) {
        if (n - i == len) {                                                  /* seeds.erl:26-30: exactly Len left */
            if (fs_matches(pats, npat, len, ref + i)) { if (cnt < cap) out[cnt] = N; cnt++; }
            break;
        }
        if (fs_matches(pats, npat, len, ref + i)) {                          /* :33-35 */
            if (cnt < cap) out[cnt] = N;
            cnt++;
            N += len; i += len;
        } else { N += 1; i += 1; }                                           /* :36 */
    }
    long m = cnt < cap ? cnt : cap;
    for (long a = 0, b = m - 1; a < b; a++, b--) { int64_t t = out[a]; out[a] = out[b]; out[b] = t; }
    return cnt;
}
