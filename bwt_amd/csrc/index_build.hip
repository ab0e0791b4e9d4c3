// GPU index builder: reference text -> suffix array -> BWT rank blocks, all in HBM.
//
// Reference functions this replaces (offline path, paths relative to /root/reference):
//   fm_index:make_index/4     src/fm_index.erl:127-141   IUPAC map, N -> random:uniform(4), $ append
//   bwt:fm/1, fm/7            src/bwt.erl:84-120         SA -> rows {F,L,I,SA}
//   dsa:dsa/1                 src/dsa.erl:15-48          suffix order, '$' smallest
//   fmi:assemble_index/8      src/fmi.erl:43-62          I column (not stored: SURVEY 8a-13)
//   fmi:get_index_pointers/1  src/fmi.erl:66-76          Pc, Pg, Pt, Last
// The suffix array of Seq++"$" is unique, so any correct construction reproduces the reference's
// rows; here: prefix doubling on the GPU (21-symbol packed initial keys, then (rank[i], rank[i+h])
// pairs), each round one 64-bit LSD radix sort of (key, position) pairs (radix.hip.h: hand-written
// histogram / scan / stable scatter, 8 bits per pass).
#include <cstring>

#include "common.h"
#include "radix.hip.h"
#include "etf_fm.h"

#include <string>
#include <vector>

// ---- AS183 (Erlang `random` module, OTP 17 stdlib; not vendored by the reference) ----------------
// random:uniform/1 as used by fm_index.erl:131 for 'N' bases.  Default process seed {3172,9814,20125}.
static inline uint32_t as183_uniform4(uint32_t st[3]) {
    st[0] = (st[0] * 171u) % 30269u;
    st[1] = (st[1] * 172u) % 30307u;
    st[2] = (st[2] * 170u) % 30323u;
    double r = st[0] / 30269.0 + st[1] / 30307.0 + st[2] / 30323.0;
    r -= (double)(long long)r;
    return (uint32_t)(long long)(r * 4.0) + 1u;      // trunc(uniform()*N)+1
}

// fm_index.erl:131-137: the base the index is built from; 0 = the reference would error(Err)
static inline uint8_t map_base(uint8_t c, uint32_t st[3]) {
    switch (c) {
    case 'N': { static const char t[4] = {'A', 'C', 'G', 'T'}; return (uint8_t)t[as183_uniform4(st) - 1]; }
    case 'B': return 'C'; case 'D': return 'G'; case 'R': return 'A'; case 'Y': return 'C';
    case 'K': return 'T'; case 'M': return 'A'; case 'S': return 'C'; case 'W': return 'A'; case 'V': return 'A';
    case 'A': case 'C': case 'G': case 'T': return c;
    default: return 0;
    }
}

// ---- kernels -------------------------------------------------------------------------------------
// symbols: 0 = end ('$' and beyond), 1..4 = A,C,G,T  (so plain integer order = '$' < A < C < G < T)
__global__ __launch_bounds__(256) void kb_init_keys(const uint8_t *__restrict__ sym, uint64_t N, uint64_t *__restrict__ key,
                                                     uint32_t *__restrict__ val) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint64_t k = 0;
#pragma unroll
    for (int t = 0; t < 21; t++) k = (k << 3) | sym[i + t];      // sym is zero-padded past N
    key[i] = k;
    val[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void kb_pair_keys(const uint32_t *__restrict__ rank, uint64_t N, uint64_t h,
                                                     uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint64_t second = i + h < N ? (uint64_t)rank[i + h] + 1ull : 0ull;
    key[i] = ((uint64_t)rank[i] << 32) | second;
    val[i] = (uint32_t)i;
}

// head[j] = j if sorted key j starts a new group else 0; counts the groups
__global__ __launch_bounds__(256) void kb_heads(const uint64_t *__restrict__ key, uint64_t N, uint32_t *__restrict__ head,
                                                 unsigned long long *ngroups) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t f = 0;
    if (j < N) {
        f = (j == 0 || key[j] != key[j - 1]) ? 1u : 0u;
        head[j] = f ? (uint32_t)j : 0u;
    }
    unsigned long long c = __popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(ngroups, c);
}

__global__ __launch_bounds__(256) void kb_scatter_rank(const uint32_t *__restrict__ head, const uint32_t *__restrict__ sa,
                                                        uint64_t N, uint32_t *__restrict__ rank) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) rank[sa[j]] = head[j];
}

// one thread per 64-byte rank block: bit-planes of 192 L codes + their counts; the `$` row is reported
__global__ __launch_bounds__(256) void kb_pack_blocks(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ sym,
                                                       uint64_t N, uint32_t nblk, uint32_t *__restrict__ blocks,
                                                       uint32_t *__restrict__ cntA, uint32_t *__restrict__ cntC,
                                                       uint32_t *__restrict__ cntG, uint32_t *__restrict__ cntT,
                                                       uint32_t *dollar) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t *o = blocks + (uint64_t)b * 16;
    for (uint32_t q = 0; q < 4; q++) {
        unsigned long long lo = 0, hi = 0;
        for (uint32_t r = 0; r < 48; r++) {
            const uint64_t row0 = (uint64_t)b * BWT_BLOCK_ROWS + q * 48 + r;
            if (row0 >= N) break;
            const uint32_t p = sa[row0];
            uint32_t code = 0;
            if (p == 0) *dollar = (uint32_t)row0 + 1;        // L = '$' (the preceding char of suffix 0)
            else {
                code = sym[p - 1] - 1u;
                c0 += code == 0; c1 += code == 1; c2 += code == 2; c3 += code == 3;
            }
            lo |= (unsigned long long)(code & 1u) << r;
            hi |= (unsigned long long)(code >> 1) << r;
        }
        o[4 * q + 1] = (uint32_t)lo;
        o[4 * q + 2] = (uint32_t)hi;
        o[4 * q + 3] = (uint32_t)(lo >> 32) | ((uint32_t)(hi >> 32) << 16);
    }
    cntA[b] = c0; cntC[b] = c1; cntG[b] = c2; cntT[b] = c3;
}

__global__ __launch_bounds__(256) void kb_block_counts(uint32_t nblk, uint32_t *__restrict__ blocks,
                                                        const uint32_t *__restrict__ sA, const uint32_t *__restrict__ sC,
                                                        const uint32_t *__restrict__ sG, const uint32_t *__restrict__ sT) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    uint32_t *o = blocks + (uint64_t)b * 16;
    o[0] = sA[b]; o[4] = sC[b]; o[8] = sG[b]; o[12] = sT[b];
}

static inline uint32_t cdiv_u(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

#define RPCHK(expr)                                                                       \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) { rc = fail(BWT_EHIP, std::string(#expr " failed: ") + hipGetErrorString(_e)); goto done; } \
    } while (0)

extern "C" int bwt_build_index(const uint8_t *ref, uint64_t n, int32_t shift, uint32_t rng_state[3], int device,
                               bwt_index **out) {
    if (!ref || !out || n < 8) return fail(BWT_EINVAL, "bwt_build_index: need at least 8 bases");
    if (n + 70001ull >= 0xffffffffull) return fail(BWT_EINVAL, "bwt_build_index: chunk too long for u32 rows");
    uint32_t st[3] = {3172u, 9814u, 20125u};
    if (rng_state) { st[0] = rng_state[0]; st[1] = rng_state[1]; st[2] = rng_state[2]; }
    const uint64_t N = n + 1;
    std::vector<uint8_t> sym(N + 64, 0);
    uint64_t cnt[5] = {0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t b = map_base(ref[i], st);
        if (!b) return fail(BWT_EINVAL, "bwt_build_index: unknown symbol in the reference (fm_index.erl:136 errors)");
        const uint8_t s = b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 3 : 4;
        sym[i] = s; cnt[s]++;
    }
    if (rng_state) { rng_state[0] = st[0]; rng_state[1] = st[1]; rng_state[2] = st[2]; }
    if (!cnt[1] || !cnt[2] || !cnt[3] || !cnt[4]) return fail(BWT_EINVAL, "bwt_build_index: every base must occur");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "bwt_build_index: no such HIP device");
    HIPCHK(hipSetDevice(device));

    int rc = 0;
    bwt_index *ix = new bwt_index();
    ix->device = device;
    DevBuf d_sym, d_key0, d_key1, d_val0, d_val1, d_rank, d_head, d_tmp, d_cnt, d_scan, d_small, d_hist;
    hipStream_t s = nullptr;
    const uint32_t nblk = (uint32_t)(N / BWT_BLOCK_ROWS) + 1;
    {
        if ((rc = d_sym.exact(N + 64)) || (rc = d_key0.exact(N * 8)) || (rc = d_key1.exact(N * 8)) ||
            (rc = d_val0.exact(N * 4)) || (rc = d_val1.exact(N * 4)) || (rc = d_rank.exact(N * 4)) ||
            (rc = d_head.exact(N * 4)) || (rc = d_small.exact(64)))
            goto done;
        RPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        RPCHK(hipMemcpyAsync(d_sym.p, sym.data(), N + 64, hipMemcpyHostToDevice, s));
        uint64_t *kk[2] = {d_key0.as<uint64_t>(), d_key1.as<uint64_t>()};
        uint32_t *vv[2] = {d_val0.as<uint32_t>(), d_val1.as<uint32_t>()};
        int cur = 0;
        const uint64_t sort_tiles = (N + RS_TILE - 1) / RS_TILE;
        const uint64_t tmp_words = std::max(dev_scan_u32_tmp_words(256 * sort_tiles), dev_scan_u32_tmp_words(N));
        if ((rc = d_hist.exact(256 * sort_tiles * 4)) || (rc = d_tmp.exact(tmp_words * 4))) goto done;
        unsigned long long *d_ng = d_small.as<unsigned long long>();
        const uint32_t gridN = cdiv_u(N, 256);

        hipLaunchKernelGGL(kb_init_keys, dim3(gridN), dim3(256), 0, s, d_sym.as<uint8_t>(), N, kk[cur], vv[cur]);
        uint64_t h = 21;
        // a (rank[i], rank[i+h]+1) pair key holds two numbers <= N + 1: only rank_bits of each half need sorting
        int rank_bits = 1;
        while ((1ull << rank_bits) <= N + 1) rank_bits++;
        for (int round = 0; round < 40; round++) {
            if (round == 0) {
                RPCHK(dev_radix_sort_pairs(kk, vv, &cur, N, 0, 63, d_hist.as<uint32_t>(), d_tmp.as<uint32_t>(), s));   // 21 x 3 bits
            } else {
                RPCHK(dev_radix_sort_pairs(kk, vv, &cur, N, 0, rank_bits, d_hist.as<uint32_t>(), d_tmp.as<uint32_t>(), s));
                RPCHK(dev_radix_sort_pairs(kk, vv, &cur, N, 32, 32 + rank_bits, d_hist.as<uint32_t>(), d_tmp.as<uint32_t>(), s));
            }
            RPCHK(hipMemsetAsync(d_ng, 0, 8, s));
            hipLaunchKernelGGL(kb_heads, dim3(gridN), dim3(256), 0, s, kk[cur], N, d_head.as<uint32_t>(), d_ng);
            unsigned long long ng = 0;
            RPCHK(hipMemcpyAsync(&ng, d_ng, 8, hipMemcpyDeviceToHost, s));
            RPCHK(hipStreamSynchronize(s));
            if (ng == N) break;                                   // every suffix has its own rank: vals is the SA
            if (round == 39) { rc = fail(BWT_EHIP, "bwt_build_index: prefix doubling did not converge"); goto done; }
            RPCHK(dev_scan_u32<true>(d_head.as<uint32_t>(), d_head.as<uint32_t>(), N, d_tmp.as<uint32_t>(), s));
            hipLaunchKernelGGL(kb_scatter_rank, dim3(gridN), dim3(256), 0, s, d_head.as<uint32_t>(), vv[cur], N,
                               d_rank.as<uint32_t>());
            hipLaunchKernelGGL(kb_pair_keys, dim3(gridN), dim3(256), 0, s, d_rank.as<uint32_t>(), N, h, kk[cur], vv[cur]);
            h *= 2;
        }
        // SA
        if ((rc = ix->sa.exact(N * 4))) goto done;
        RPCHK(hipMemcpyAsync(ix->sa.p, vv[cur], N * 4, hipMemcpyDeviceToDevice, s));
        // rank blocks
        if ((rc = ix->blocks.exact((size_t)nblk * 64)) || (rc = d_cnt.exact((size_t)nblk * 4 * 4 + 64)) ||
            (rc = d_scan.exact((size_t)nblk * 4 * 4 + 64)))
            goto done;
        RPCHK(hipMemsetAsync(d_small.p, 0, 64, s));
        uint32_t *cA = d_cnt.as<uint32_t>(), *cC = cA + nblk, *cG = cC + nblk, *cT = cG + nblk;
        uint32_t *sA = d_scan.as<uint32_t>(), *sC = sA + nblk, *sG = sC + nblk, *sT = sG + nblk;
        uint32_t *d_dollar = d_small.as<uint32_t>() + 4;
        hipLaunchKernelGGL(kb_pack_blocks, dim3(cdiv_u(nblk, 256)), dim3(256), 0, s, ix->sa.as<uint32_t>(),
                           d_sym.as<uint8_t>(), N, nblk, ix->blocks.as<uint32_t>(), cA, cC, cG, cT, d_dollar);
        RPCHK(dev_scan_u32<false>(cA, sA, nblk, d_tmp.as<uint32_t>(), s));
        RPCHK(dev_scan_u32<false>(cC, sC, nblk, d_tmp.as<uint32_t>(), s));
        RPCHK(dev_scan_u32<false>(cG, sG, nblk, d_tmp.as<uint32_t>(), s));
        RPCHK(dev_scan_u32<false>(cT, sT, nblk, d_tmp.as<uint32_t>(), s));
        hipLaunchKernelGGL(kb_block_counts, dim3(cdiv_u(nblk, 256)), dim3(256), 0, s, nblk, ix->blocks.as<uint32_t>(), sA,
                           sC, sG, sT);
        uint32_t dollar = 0;
        RPCHK(hipMemcpyAsync(&dollar, d_dollar, 4, hipMemcpyDeviceToHost, s));
        // Ref1 = 200 x N ++ .ref (pre-substitution bytes, fm_index.erl:130) ++ 200 x N
        ix->ref1_len = n + 2 * BWT_REF_EXT;
        if ((rc = ix->ref1.exact(ix->ref1_len))) goto done;
        RPCHK(hipMemsetAsync(ix->ref1.p, 'N', ix->ref1_len, s));
        RPCHK(hipMemcpyAsync(ix->ref1.as<uint8_t>() + BWT_REF_EXT, ref, n, hipMemcpyHostToDevice, s));
        RPCHK(hipStreamSynchronize(s));
        RPCHK(hipGetLastError());
        if (!dollar) { rc = fail(BWT_EHIP, "bwt_build_index: no $ row found"); goto done; }
        // pointers (fmi.erl:66-76): row 1 is '$', then the A, C, G, T blocks of F
        ix->Pc = (uint32_t)(2 + cnt[1]); ix->Pg = (uint32_t)(ix->Pc + cnt[2]); ix->Pt = (uint32_t)(ix->Pg + cnt[3]);
        ix->Last = (uint32_t)N; ix->shift = shift;
        RPCHK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
        ix->d.blocks = ix->blocks.as<uint4>(); ix->d.sa = ix->sa.as<uint32_t>(); ix->d.ref1 = ix->ref1.as<uint8_t>();
        ix->d.ref1_len = ix->ref1_len;
        ix->d.Pc = ix->Pc; ix->d.Pg = ix->Pg; ix->d.Pt = ix->Pt; ix->d.Last = ix->Last; ix->d.dollar = dollar; ix->d.nblk = nblk;
        ix->d.c0[0] = 1; ix->d.c0[1] = ix->Pc - 1; ix->d.c0[2] = ix->Pg - 1; ix->d.c0[3] = ix->Pt - 1;
        if ((rc = bwt_index_make_refplanes(ix, s))) goto done;
    }
done:
    d_sym.release(); d_key0.release(); d_key1.release(); d_val0.release(); d_val1.release(); d_rank.release();
    d_head.release(); d_tmp.release(); d_cnt.release(); d_scan.release(); d_small.release(); d_hist.release();
    if (s) (void)hipStreamDestroy(s);
    // the suffix-interval table last: the builder's buffers are gone, so its size is weighed against what is really free
    if (!rc) rc = bwt_index_set_ktable(ix, -1, -1);
    if (rc) { bwt_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

// Writes the index back in the reference's on-disk format: <base>.fm (ETF of {Meta,FM}) and
// <base>.ref -- what fm_index:make_index/4 leaves behind (src/fm_index.erl:129-141).
extern "C" int bwt_index_save(const bwt_index *ix, const char *fm_path, const char *ref_path) {
    if (!ix || !fm_path || !ref_path) return fail(BWT_EINVAL, "bwt_index_save: null argument");
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t n = ix->Last;
    FmHost fm;
    fm.Pc = ix->Pc; fm.Pg = ix->Pg; fm.Pt = ix->Pt; fm.Last = ix->Last; fm.shift = ix->shift;
    fm.SA.resize(n);
    fm.L.resize(n);
    std::vector<uint32_t> blk((size_t)ix->d.nblk * 16);
    HIPCHK(hipMemcpy(fm.SA.data(), ix->sa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(blk.data(), ix->blocks.p, blk.size() * 4, hipMemcpyDeviceToHost));
    static const char tb[4] = {'A', 'C', 'G', 'T'};
    for (uint32_t r0 = 0; r0 < n; r0++) {
        const uint32_t *w = &blk[(size_t)(r0 / BWT_BLOCK_ROWS) * 16];
        const uint32_t i = r0 % BWT_BLOCK_ROWS;
        fm.L[r0] = (uint8_t)tb[blk_get_code(w, i)];
    }
    fm.L[ix->d.dollar - 1] = '$';
    std::string err;
    int rc = fm_write_etf(fm_path, fm, err);
    if (rc) return fail(rc, err);
    std::vector<uint8_t> ref(ix->ref1_len - 2 * BWT_REF_EXT);
    if (!ref.empty())
        HIPCHK(hipMemcpy(ref.data(), ix->ref1.as<uint8_t>() + BWT_REF_EXT, ref.size(), hipMemcpyDeviceToHost));
    FILE *f = fopen(ref_path, "wb");
    if (!f) return fail(BWT_EIO, std::string("cannot create ") + ref_path);
    size_t w = fwrite(ref.data(), 1, ref.size(), f);
    fclose(f);
    if (w != ref.size()) return fail(BWT_EIO, "short write on .ref");
    return 0;
}

extern "C" int bwt_index_export(const bwt_index *ix, uint8_t *L, uint32_t *SA) {
    if (!ix || !L || !SA) return fail(BWT_EINVAL, "bwt_index_export: null argument");
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t n = ix->Last;
    HIPCHK(hipMemcpy(SA, ix->sa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> blk((size_t)ix->d.nblk * 16);
    HIPCHK(hipMemcpy(blk.data(), ix->blocks.p, blk.size() * 4, hipMemcpyDeviceToHost));
    static const char tb[4] = {'A', 'C', 'G', 'T'};
    for (uint32_t r0 = 0; r0 < n; r0++) {
        const uint32_t *w = &blk[(size_t)(r0 / BWT_BLOCK_ROWS) * 16];
        const uint32_t i = r0 % BWT_BLOCK_ROWS;
        L[r0] = (uint8_t)tb[blk_get_code(w, i)];
    }
    L[ix->d.dollar - 1] = '$';
    return 0;
}
