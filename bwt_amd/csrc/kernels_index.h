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
    uint32_t Pc, Pg, Pt, Last, dollar, nblk;
    uint32_t c0[4];        // {1, Pc-1, Pg-1, Pt-1}: 0-based first row of each F block (src/bwt.erl:105-112)
};

struct Counters {          // device-side work counters (roofline accounting)
    unsigned long long lf_steps, sa_reads, n_tasks, sw_cells, window_bytes, n_float, err_flags;
    unsigned int max_w, max_lq, n_big, fp_count, tq_count, pad_;
};

