// Constants and device-side structs shared by every translation unit of libbwtgpu.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BWT_MIN_LEN 8
#define BWT_MAX_RANGE 45u
#define BWT_MAX_LEN 15
#define BWT_TOLERANCE 20u
#define BWT_REF_EXT 200
#define BWT_BLOCK_ROWS 192u

// task status codes (t_info low byte)
enum { TS_SEEDS = 0, TS_NO_SEEDS = 1, TS_TOO_MANY = 2, TS_NONE = 3, TS_CRASH = 4 };
// candidate status codes
enum { CS_OK = 0, CS_NO_MATCH = 1, CS_WINDOW_OOR = 2, CS_NEED_FP = 3, CS_CIG_OVERFLOW = 4, CS_PENDING = 5 };

// Rank block (64 B, 192 rows) = four 16-byte quarters, one per lane of the quad that reads it, all alike:
//   word 0   count of rows <= 192*blk whose L has code q   (q = quarter index = base code A,C,G,T; `$` excluded)
//   word 1   low  bit-plane of rows 48q .. 48q+31          (bit i = low bit of the 2-bit code of row 48q+i)
//   word 2   high bit-plane of rows 48q .. 48q+31
//   word 3   rows 48q+32 .. 48q+47: low plane in bits 0-15, high plane in bits 16-31
// so every lane of a quad does identical work: xor/and its two planes against the wanted code, mask the
// first t rows, popcount; the lane whose quarter index equals the code adds its count word.
__host__ __device__ inline void blk_set_code(uint32_t *w, uint32_t i, uint32_t code) {
    uint32_t *q = w + 4 * (i / 48);
    const uint32_t r = i % 48;
    if (r < 32) { q[1] |= (code & 1u) << r; q[2] |= (code >> 1) << r; }
    else { q[3] |= (code & 1u) << (r - 32); q[3] |= (code >> 1) << (r - 16); }
}
__host__ __device__ inline uint32_t blk_get_code(const uint32_t *w, uint32_t i) {
    const uint32_t *q = w + 4 * (i / 48);
    const uint32_t r = i % 48;
    if (r < 32) return ((q[1] >> r) & 1u) | (((q[2] >> r) & 1u) << 1);
    return ((q[3] >> (r - 32)) & 1u) | (((q[3] >> (r - 16)) & 1u) << 1);
}

struct DevIndex {
    const uint4 *blocks;   // nblk x 4 uint4, quarter q of block b at blocks[4b+q] (layout above)
    const uint32_t *sa;    // SA[row] at sa[row-1]
    const uint8_t *ref1;   // 200 x 'N' ++ .ref ++ 200 x 'N'
    uint64_t ref1_len;
    const uint32_t *rplanes;  // bit-planes over Ref1 (k_read_planes layout)
    uint64_t rpw;             // their size in words (all four planes)
    uint32_t Pc, Pg, Pt, Last, dollar, nblk;
    uint32_t c0[4];        // {1, Pc-1, Pg-1, Pt-1}: 0-based first row of each F block (src/bwt.erl:105-112)
    const uint64_t *ktab;  // suffix-interval table, levels 8..kt_k (kernels.hip.h: kt_build); null when kt_k == 0
    const uint4 *ktw;      // level kt_k as 64-byte entries that also carry the first 14 SA values, or null (then ktab holds it)
    int kt_k;
    // LOWER levels kept as 64-byte entries as well (behind the top level in ktw; round 4): ktw_off[m - 8] = where level m starts in
    // ktw, in entries, or KTW_NONE.  A read's short last sub-read (12 bases of a 100-bp read, 14 of a 150-bp one) ends at such a
    // level; on an index where that level's k-mers mostly have a seed list -- 4^12 k-mers over 2.3e8 positions: 14 occurrences
    // each -- its seeds come with the table line instead of by a dozen random SA reads.  ktw_off[7] (the top level) is 0.
    uint32_t ktw_off[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    // one bit per 64 entries of the 8-byte levels of ktab: clear = all 64 are plain "too many seeds" entries (the search ran its
    // full length and the interval is wider than MAX_RANGE), so a lookup that finds the bit clear knows its answer without the
    // table line.  ktsum_levels: bit m set = level m is sparse enough for the check to pay (kernels.hip.h: k_kt_summary).
    const uint32_t *ktsum = nullptr;
    uint32_t ktsum_levels = 0;
};

#define KTW_NONE 0xffffffffu
// where level m (8..15) starts in ktw, or KTW_NONE (chained selects: a per-lane index into the kernel-argument array goes through scratch)
__device__ __forceinline__ uint32_t ktw_level_off(const DevIndex &ix, int m) {
    uint32_t o = KTW_NONE;
    o = m == 9 ? ix.ktw_off[1] : o; o = m == 10 ? ix.ktw_off[2] : o; o = m == 11 ? ix.ktw_off[3] : o; o = m == 12 ? ix.ktw_off[4] : o;
    o = m == 13 ? ix.ktw_off[5] : o; o = m == 14 ? ix.ktw_off[6] : o; o = m == 15 ? ix.ktw_off[7] : o; o = m == 8 ? ix.ktw_off[0] : o;
    return ix.ktw ? o : KTW_NONE;
}

// byte-stream bit-planes (reads, Ref1) ---------------------------------------------------------------
__device__ __forceinline__ int base_code(uint32_t c) {  // A C G T -> 0..3, anything else -> -1
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// ------------------------------------------------------------------------------------------------
// k_read_planes: a byte stream of bases (the concatenated reads, Ref1) as four bit-planes (bit i of a plane =
// property of base i): low / high bit of the 2-bit code (A C G T = 0..3; anything else 0), is-N, is-other (neither
// ACGT nor N).  The planes are interleaved word by word -- planes[4*w + p] holds bits 32w..32w+31 of plane p -- so the
// four properties of a stretch of bases sit in the same 16 bytes.  Everything downstream (sub-read enumeration, seed
// keys, the extend prefilter) works on 16..32-bit windows of these planes instead of bytes.
// Lane = 16 bases (one 16-byte load); the plane bits come out of each 4-byte word by SWAR arithmetic:
//   code bits: hi = bit 2, lo = bit 1 ^ bit 2 of the ASCII code (A 41, C 43, G 47, T 54);
//   is-ACGTN: the byte equals the entry of an 8-byte table (v_perm_b32) picked by its bits 1..3, by an exact zero-byte test;
//   a dot product with the weights 1 2 4 8 (16 32 64 128) gathers the flag bits of a word's four bytes (v_dot4_u32_u8).
#define RP_LO 0
#define RP_HI 1
#define RP_N 2
#define RP_O 3
__device__ __forceinline__ uint32_t rp_zero_bytes(uint32_t d) {            // 1 in bit 0 of every byte of d that is zero
    return (~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
}
// One 4-byte word: v_perm_b32 as an 8-entry table -- sel = bits 1..3 of the ASCII code (A 0, C 1, T 2, G 3, N 7) picks the
// byte the code would have to be ('A' 'C' 'T' 'G' 0 0 0 'N'); one exact zero-byte test of x ^ expected then says "one of ACGTN",
// and among those only N has bit 2 of sel.  Code bits as before: hi = sel bit 1, lo = sel bit 0 ^ sel bit 1.
__device__ __forceinline__ void rp_word(uint32_t x, uint32_t &l, uint32_t &h, uint32_t &nb, uint32_t &ob) {
    const uint32_t sel = (x >> 1) & 0x07070707u;
    const uint32_t expect = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, sel);
    const uint32_t a = rp_zero_bytes(x ^ expect);
    const uint32_t s1 = sel >> 1;
    nb = a & (sel >> 2);
    const uint32_t acgt = a ^ nb;
    ob = a ^ 0x01010101u;
    h = s1 & acgt;
    l = (sel ^ s1) & acgt;
}
// the flag bits (bit 0 of every byte) of four words = 16 bases as 16 bits: one dot product per word (weights 1 2 4 8 / 16 32 64 128)
__device__ __forceinline__ uint32_t rp_gather16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    const uint32_t lo = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(f3, 0x80402010u, __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false), false);
    return lo | (hi << 8);
}
// the 16 bases of segment s of a stream of `total` bases as four words (past the end: 'A' = all planes 0)
__device__ __forceinline__ void rp_load16(const uint8_t *__restrict__ bases, uint64_t total, uint64_t s, uint32_t (&x)[4]) {
    const uint64_t p0 = s * 16;
    x[0] = x[1] = x[2] = x[3] = 0x41414141u;
    if (p0 + 16 <= total) {
        const uint4 v = *(const uint4 *)(bases + p0);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if (p0 < total) {
        for (uint32_t i = 0; p0 + i < total; i++)
            x[i >> 2] = (x[i >> 2] & ~(0xffu << (8 * (i & 3)))) | ((uint32_t)bases[p0 + i] << (8 * (i & 3)));
    }
}
// ... -> the four plane words of the 32 bases an (even, odd) lane pair holds; valid in the even lane.  Whole waves.
__device__ __forceinline__ uint4 rp_pair_word(const uint32_t (&x)[4]) {
    uint32_t fl[4], fh[4], fn[4], fo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rp_word(x[k], fl[k], fh[k], fn[k], fo[k]);
    uint32_t out[4];
    out[RP_LO] = rp_gather16(fl[0], fl[1], fl[2], fl[3]);
    out[RP_HI] = rp_gather16(fh[0], fh[1], fh[2], fh[3]);
    out[RP_N] = rp_gather16(fn[0], fn[1], fn[2], fn[3]);
    out[RP_O] = rp_gather16(fo[0], fo[1], fo[2], fo[3]);
    // the odd lane's 16 bits complete the even lane's word
    uint4 w;
    w.x = out[0] | ((uint32_t)__shfl_xor((int)out[0], 1, 64) << 16);
    w.y = out[1] | ((uint32_t)__shfl_xor((int)out[1], 1, 64) << 16);
    w.z = out[2] | ((uint32_t)__shfl_xor((int)out[2], 1, 64) << 16);
    w.w = out[3] | ((uint32_t)__shfl_xor((int)out[3], 1, 64) << 16);
    return w;
}
static __global__ __launch_bounds__(256) __attribute__((unused)) void k_read_planes(const uint8_t *__restrict__ bases, uint64_t total,
                                                      uint32_t *__restrict__ planes) {
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nseg = (((total + 15) >> 4) + 1) & ~1ull;      // 16-base segments, lane pairs finish whole words
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += nthreads) {
        uint32_t x[4];
        rp_load16(bases, total, s, x);
        const uint4 w = rp_pair_word(x);
        if (!(s & 1)) ((uint4 *)planes)[s >> 1] = w;
    }
}
// words the plane buffer of a stream of `total` bases needs (u32; 16-byte groups, padded so that windows may run on)
static inline uint64_t rp_words(uint64_t total) { return 4 * (((total + 63) / 64) * 2 + 8); }
// 32 bits of plane p starting at absolute bit position pos
__device__ __forceinline__ uint32_t plane_bits(const uint32_t *__restrict__ planes, int p, uint64_t pos) {
    const uint64_t w = pos >> 5;
    return __builtin_amdgcn_alignbit(planes[4 * (w + 1) + p], planes[4 * w + p], (uint32_t)pos & 31u);
}
// ... of all four planes (x lo, y hi, z is-N, w is-other): two 16-byte loads
__device__ __forceinline__ uint4 planes4(const uint32_t *__restrict__ planes, uint64_t pos) {
    const uint64_t w = pos >> 5;
    const uint32_t sh = (uint32_t)pos & 31u;
    const uint4 a = ((const uint4 *)planes)[w], b = ((const uint4 *)planes)[w + 1];
    uint4 r;
    r.x = __builtin_amdgcn_alignbit(b.x, a.x, sh); r.y = __builtin_amdgcn_alignbit(b.y, a.y, sh);
    r.z = __builtin_amdgcn_alignbit(b.z, a.z, sh); r.w = __builtin_amdgcn_alignbit(b.w, a.w, sh);
    return r;
}


struct Counters {          // device-side work counters (roofline accounting)
    unsigned long long lf_steps, sa_reads, n_tasks, sw_cells, window_bytes, n_float, err_flags, kt_lookups;
    unsigned int max_w, max_lq, n_big, fp_count, tq_count, dq_count, tq2_count, n_mid2, n_mid4, n_small;
    unsigned int n_gq, pad_;   // groups k_seed_fast handed back to k_seed_tasks
    unsigned int max_D, n_pend;  // (n_pend: wave passes of the seed kernels that left a task to k_search_pending; 0 = it has nothing to do) // largest cluster width D the seed kernels staged (window = Lq + D: sizes the extend stage without a round trip)
    unsigned int why[6];       // ... by reason: [0] base outside ACGT, [1] table overflow, [4] > 64 seeds in a read, [5] > CL_STAGE clusters
    unsigned int band_in, band_left, band_next, pf_ov;   // (pf_ov: candidates the fused prefilter pass listed for the quad-per-candidate pass)   // k_sw_band, first tier: DP candidates it was given, those it passed on, its work counter
    unsigned int band2_in, band2_left, band2_next, band_direct;   // ... second tier (wider band); band_direct: candidates the first tier sent straight to the third
    unsigned int band3_in, band3_left, band3_next, tqw_count;   // (tqw_count: what the wide pass of k_sw_pk left for the generic trace kernel)   // ... third tier (64 diagonals); what it passes on goes to the full-rectangle kernels
    unsigned long long cig_total;                            // bytes of all CIGAR texts of the batch's hits (k_hits_count)
    // round 4: a 12-diagonal tier in front of the banded tiers, for the candidates the prefilter found on a square window with at most
    // six mismatches (its own queue, dq_lo): candidates it was given, those it passed on to the 20-diagonal tier, its work counter, those
    // it sent straight to the 32-diagonal tier; the length of its queue
    unsigned int band0_in, band0_left, band0_next, band0_direct, dq_lo_count;
    // ... and in front of that tier the certificate kernel k_sw_cert: what it could not settle (the 12-diagonal tier's queue then), what it settled
    unsigned int cert_left, cert_done, cert2_in;      // (cert2_in: the queue of its wider instance, candidates with 7..10 mismatches)
    unsigned int n_huge, pad3_[3];   // reads of more seeds than k_cluster_big's LDS holds, listed by it for k_cluster_huge
};

