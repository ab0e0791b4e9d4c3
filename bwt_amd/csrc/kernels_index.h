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
    const uint32_t *rplanes;  // bit-planes over Ref1 (k_read_planes layout), rpw words per plane
    uint64_t rpw;
    uint32_t Pc, Pg, Pt, Last, dollar, nblk;
    uint32_t c0[4];        // {1, Pc-1, Pg-1, Pt-1}: 0-based first row of each F block (src/bwt.erl:105-112)
    const uint64_t *ktab;  // suffix-interval table, levels 8..kt_k (kernels.hip.h: kt_build); null when kt_k == 0
    int kt_k;
};

// byte-stream bit-planes (reads, Ref1) ---------------------------------------------------------------
__device__ __forceinline__ int base_code(uint32_t c) {  // A C G T -> 0..3, anything else -> -1
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// ------------------------------------------------------------------------------------------------
// k_read_planes: the concatenated read bases as four bit-planes over the byte stream (bit i of a plane
// = property of base i): low / high bit of the 2-bit code (A C G T = 0..3), is-N, is-other (neither
// ACGT nor N).  One coalesced 64-byte read and four ballots per wave iteration; everything downstream
// (sub-read enumeration, LF steps) works on 16..32-bit windows of these planes instead of bytes.
#define RP_LO 0
#define RP_HI 1
#define RP_N 2
#define RP_O 3
static __global__ __launch_bounds__(256) void k_read_planes(const uint8_t *__restrict__ bases, uint64_t total,
                                                      uint32_t *__restrict__ planes, uint64_t pw) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t nchunks = (total + 63) >> 6;
    // RP_U independent 64-byte loads in flight per wave before any of them is consumed: the kernel is a pure stream and
    // bytes in flight per wave set its rate (4 loads: 1.0 TB/s)
    constexpr int RP_U = 16;
    for (uint64_t c0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c0 < nchunks; c0 += RP_U * nwaves) {
        uint32_t ch[RP_U];
#pragma unroll
        for (int u = 0; u < RP_U; u++) {
            const uint64_t pos = (c0 + (uint64_t)u * nwaves) * 64 + lane;
            ch[u] = pos < total ? bases[pos] : 'A';
        }
#pragma unroll
        for (int u = 0; u < RP_U; u++) {
            const uint64_t ci = c0 + (uint64_t)u * nwaves;
            if (ci >= nchunks) break;
            const uint32_t c = ch[u];
            const int code = base_code(c);
            const unsigned long long lo = __ballot(code == 1 || code == 3), hi = __ballot(code >= 2),
                                     nn = __ballot(c == 'N'), oo = __ballot(code < 0 && c != 'N');
            if (lane < 8) {
                const unsigned long long m = (lane >> 1) == 0 ? lo : (lane >> 1) == 1 ? hi : (lane >> 1) == 2 ? nn : oo;
                planes[(uint64_t)(lane >> 1) * pw + 2 * ci + (lane & 1)] = (lane & 1) ? (uint32_t)(m >> 32) : (uint32_t)m;
            }
        }
    }
}
// 32 bits of a plane starting at absolute bit position pos
__device__ __forceinline__ uint32_t plane_bits(const uint32_t *__restrict__ plane, uint64_t pos) {
    const uint64_t w = pos >> 5;
    return __builtin_amdgcn_alignbit(plane[w + 1], plane[w], (uint32_t)pos & 31u);
}


struct Counters {          // device-side work counters (roofline accounting)
    unsigned long long lf_steps, sa_reads, n_tasks, sw_cells, window_bytes, n_float, err_flags, kt_lookups;
    unsigned int max_w, max_lq, n_big, fp_count, tq_count, dq_count, tq2_count, n_mid2, n_mid4, pad_;
};

