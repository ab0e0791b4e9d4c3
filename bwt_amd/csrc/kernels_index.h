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

struct DevIndex {
    const uint4 *blocks;   // nblk x 4 uint4 : [cntA cntC cntG cntT][64 bases][64 bases][64 bases]
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

