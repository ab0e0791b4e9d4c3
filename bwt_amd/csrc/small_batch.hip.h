// libbwtgpu: bwt_small -- the hot path for batches of the size the reference's job API delivers.
//
// The master hands a worker 10 lists of 200 reads per message (src/master.erl:59; src/worker_bwt.erl:16,62-69 runs the
// `running` clause once per list).  Through the general path such a call costs 0.7-0.9 ms, nearly all of it fixed: about forty
// launches, six synchronisations, pageable copies in both directions, two-call sizing -- against 5.8 ms for 12.5 M reads.
// This path keeps the KERNELS (the same seed / cluster / prefilter / DP / hit kernels, so the answer is the general path's
// by construction: tests/test_gpu_small.py compares whole tables) and removes what surrounds them:
//   * the reads are copied once into PINNED host memory and the first kernel reads them from there (zero-copy: a DMA of 64 KB
//     from pinned memory costs 57 us on this part, a kernel reading it 4 us -- profiles/r4/launch_probe.txt); the same kernel
//     makes the device copy, the offsets, the bit-planes and clears the batch's counters and per-read words;
//   * no size ever comes back to the host in mid-run: every buffer has a fixed capacity (reads, seeds, candidates, window width),
//     the kernels that take a queue length read it on the device, and a capacity that does not hold raises a flag -- the call
//     is then redone through the general path (reported in bwt_small_result.path);
//   * the DP queue of a batch this size is always k_sw_wave's; the generic traced / fp64 kernels behind it start on a handful of
//     waves and leave at once when their queues are empty;
//   * the compact hit table is written straight into pinned host memory by the last kernel; ONE synchronisation ends the call.
// Included at the end of bwt_gpu.hip (it uses bwt_batch and the launch helpers there).
#pragma once

#define SMALL_MAX_READS 4096u
#define SMALL_MAX_BASES (SMALL_MAX_READS * 256ull)     // in all; a read of this path has at most ~300 bases (small_wave_wcap)
#define SMALL_SEED_CAP (1u << 20)            // seeds left to the cluster kernels (u32 positions)
#define SMALL_CAND_CAP 32768u                // candidates (= k_sw_wave's queue limit)
#define SMALL_W_TRACE 2048u                  // widest window the generic traced / fp64 kernels of this path are sized for
#define SMALL_SEQ_OFF (((size_t)SMALL_MAX_READS + 1) * 8)   // the call number's place in the pinned input block (behind the offsets)
#define SMALL_ONE_PASS 256u                  // reads up to which k_sw_wave runs once, with full LDS
#define SMALL_ERR_CAP 16ull                  // Counters::err_flags: a capacity of this path did not hold

struct SmallZero { uint32_t *p[8]; uint32_t words[8]; };

// pinned host block -> device: offsets, bases (for the kernels that read bytes), bit-planes (k_read_planes' arithmetic); clears
// the arrays the seed stage expects zeroed and the batch's counters.  h_bases is 16-byte aligned.
__global__ __launch_bounds__(256) void k_small_in(const uint64_t *__restrict__ h_off, const uint8_t *__restrict__ h_bases, uint32_t n,
                                                   uint64_t total, uint64_t *__restrict__ off, uint8_t *__restrict__ bases,
                                                   uint32_t *__restrict__ planes, uint64_t plane_words, SmallZero z,
                                                   const unsigned long long *__restrict__ h_seq, unsigned long long *__restrict__ d_seq) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    // the call's number, for the `done` word the last kernel writes: it travels in the pinned block, not as a launch argument, so
    // that a captured graph of these launches can be replayed as it is
    if (tid == 0) *d_seq = *h_seq;
    const uint64_t nseg = (((total + 15) >> 4) + 1) & ~1ull;
    for (uint64_t s = tid; s < nseg; s += nthreads) {
        uint32_t x[4];
        rp_load16(h_bases, total, s, x);
        *(uint4 *)(bases + s * 16) = make_uint4(x[0], x[1], x[2], x[3]);      // (the device buffer is padded to whole segments)
        const uint4 w = rp_pair_word(x);
        if (!(s & 1)) ((uint4 *)planes)[s >> 1] = w;
    }
    for (uint64_t w = (nseg >> 1) * 4 + tid; w < plane_words; w += nthreads) planes[w] = 0u;      // windows may run on past the last base
    for (uint64_t i = tid; i <= n; i += nthreads) off[i] = h_off[i];
#pragma unroll
    for (int k = 0; k < 8; k++)
        for (uint64_t i = tid; i < z.words[k]; i += nthreads) z.p[k][i] = 0u;
}

// exclusive scan of up to three u32 arrays of n <= 4096 entries (blockIdx.x picks the array) -> u64 offsets + total; ONE block per
// array.  cap[a] (0 = none): a total above it raises SMALL_ERR_CAP and the array is cleared (offsets all zero), so that whatever
// runs behind this launch finds nothing to do instead of writing past a buffer.
struct SmallScan { uint32_t *in[3]; uint64_t *out[3]; uint64_t cap[3]; };
__global__ __launch_bounds__(256) void k_small_scan(SmallScan a, uint32_t n, Counters *ctr) {
    uint32_t *in = a.in[blockIdx.x];
    uint64_t *out = a.out[blockIdx.x];
    const uint64_t cap = a.cap[blockIdx.x];
    constexpr uint32_t ITEMS = SMALL_MAX_READS / 256;
    const uint32_t base = threadIdx.x * ITEMS;
    uint32_t v[ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint64_t excl;
    const uint64_t total = block_scan_u64(s, excl);
    const bool over = cap && total > cap;
    uint64_t run = excl;
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        if (base + k < n) { out[base + k] = over ? 0ull : run; if (over) in[base + k] = 0u; }
        run += v[k];
    }
    if (threadIdx.x == 0) { out[n] = over ? 0ull : total; if (over) atomicOr(&ctr->err_flags, SMALL_ERR_CAP); }
}

// the compact hit table and the run's counters into the pinned result block
struct SmallHeader { unsigned long long n_hits, n_extra, xcig_bytes, n_cand; Counters ctr; unsigned long long done; };

// hits_count + the three scans + hits_write + the copy out as ONE launch of SMALL_HIT_BLOCKS co-resident blocks (one read per lane):
//   phase 1  a lane counts its read's hits / extras / text bytes (hit_count_one, as k_hits_count); block totals to bt[]
//   barrier  an atomic counter every block bumps once and spins on (Counters::pad_, cleared by k_small_in; 16 blocks: ~1 us,
//            profiles/r4/launch_probe.txt); the spin gives up after ~1 s and flags the batch, so every wave leaves
//   phase 2  a lane's offsets = the totals of the blocks before its own + its place in the block's scan
//   phase 3  hit_put (as k_hits_write) into a device copy of the result block
//   barrier  (the same counter, on to 2 x blocks)
//   phase 4  the used part of every section goes to the pinned block 16 bytes a lane, whole lines per wave; block 0 writes the header.
//            (hit_put's records and text pieces written to host memory one by one -- a status byte, a 16-byte record, 1..8 bytes of
//            text per store, each its own transaction on the link -- made this kernel 30 us for a list of 200 reads.)
// one hit into the small path's table.  As hit_put, but the text of a hit that is not plain is not packed against its neighbour's:
// it takes the whole 16-byte pieces of its slot that hold it (a text sits right-aligned in its slot, slots are 16-byte aligned), so
// that it moves as one or two uint4 instead of byte pieces at an odd address -- bwt_hit_extra names offset and length of every
// text, bwt_hits_expand asks for nothing more.  co1: a multiple of 16; returns the bytes taken.
__device__ __forceinline__ uint32_t small_text_bytes(uint32_t clf) { return (clf + 15u) & ~15u; }
__device__ __forceinline__ void hit_put_small(uint32_t rd, uint64_t k, uint64_t ho1, uint64_t xo1, uint64_t co1, uint32_t wl, uint32_t clf, int32_t shift,
                                          int32_t score, uint32_t S, const uint8_t *__restrict__ cig, uint32_t cig_stride,
                                          HitRec *__restrict__ h_rec, HitExtra *__restrict__ x_rec, uint8_t *__restrict__ x_cig) {
    HitRec hr; hr.read = rd; hr.score = score;
    hr.pos = (int64_t)S - (int64_t)wl + shift;                 // worker_bwt.erl:94
    h_rec[ho1] = hr;
    if (!(clf & CIG_DIAG)) {
        const uint32_t padded = small_text_bytes(clf);
        HitExtra x; x.hit = (uint32_t)ho1; x.win_len = wl; x.cig_off = (uint32_t)co1 + (padded - clf); x.cig_len = clf;
        x_rec[xo1] = x;
        const uint4 *src = (const uint4 *)(cig + (k + 1) * (uint64_t)cig_stride - padded);
        uint4 *dst = (uint4 *)(x_cig + co1);
        for (uint32_t i = 0; i < padded / 16; i++) dst[i] = src[i];
    }
}

#define SMALL_HIT_BLOCKS (SMALL_MAX_READS / 256u)
__global__ __launch_bounds__(256) void k_small_hits(uint32_t n, int32_t shift, const uint64_t *__restrict__ clus_off,
                                                     const int32_t *__restrict__ c_status, const int32_t *__restrict__ c_score,
                                                     const uint32_t *__restrict__ c_ciglen, const uint32_t *__restrict__ c_S,
                                                     const uint32_t *__restrict__ c_wlen, const uint8_t *__restrict__ cig, uint32_t cig_stride,
                                                     const uint32_t *__restrict__ r_qty, Counters *ctr, unsigned long long *bt /* blocks x 4 */,
                                                     uint8_t *__restrict__ h_out, uint8_t *__restrict__ d_out, uint32_t off_status, uint32_t off_hits,
                                                     uint32_t off_extra, uint32_t off_cig, uint64_t cap_text,
                                                     const unsigned long long *__restrict__ d_seq) {
    // h_out: the pinned result block (header at 0, then the four sections at off_*); d_out: its copy in device memory
    SmallHeader *const o_hdr = (SmallHeader *)h_out;
    int8_t *const o_status = (int8_t *)(d_out + off_status);
    HitRec *const o_hits = (HitRec *)(d_out + off_hits);
    HitExtra *const o_extra = (HitExtra *)(d_out + off_extra);
    uint8_t *const o_cig = d_out + off_cig;
    // *d_seq: the call's number (k_small_in left it there); the LAST block to finish writes it to the header's `done` word -- every block fences its writes to
    // host memory and bumps the counter once more, and the host, spinning on that word, knows the whole table has landed without
    // paying hipStreamSynchronize's wake-up (6 against 12 us, profiles/r4/launch_probe.txt)
#ifdef SMALL_HITS_TIMING
    unsigned long long TT[12]; int tti = 0;
#define TMARK() do { __syncthreads(); TT[tti++] = wall_clock64(); } while (0)
#else
#define TMARK() do { } while (0)
#endif
    TMARK();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < n;
    uint32_t nh = 0, nx = 0, xcb = 0;
    unsigned long long cbr = 0;
    int32_t st = 1;                                   // sga returned []
    if (valid && (r_qty[r] >> 31)) st = -1;
    const uint64_t a = valid ? clus_off[r] : 0ull, b = valid ? clus_off[r + 1] : 0ull;
    if (b > a) st = 0;
    // Counting and writing both take the wave's 64 reads' candidates -- one contiguous range -- 64 at a time, a LANE PER CANDIDATE;
    // a lane finds its candidate's read by bisection over the reads' range ends (sb[], relative to the wave's first candidate).
    // (A lane per READ looping over its candidates: a contig's repeats give a few reads per list thirty clusters each, and those
    // lanes' loops were 24 of this kernel's 30 us.)
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ uint32_t sb_all[4][64];
    __shared__ uint32_t acc_all[4][64][5];                // per read: hits, extras, text bytes (padded), CIGAR bytes, window out of range
    uint32_t *const sb = sb_all[threadIdx.x >> 6];
    uint32_t (*const acc)[5] = acc_all[threadIdx.x >> 6];
    const uint32_t r0 = r - lane;
    const bool wave_on = r0 < n;
    const uint32_t nvalid = wave_on ? (n - r0 < 64u ? n - r0 : 64u) : 1u;
    const uint64_t A = (uint64_t)__shfl((long long)a, 0, 64), B = (uint64_t)__shfl((long long)b, (int)nvalid - 1, 64);
    sb[lane] = valid ? (uint32_t)(b - A) : 0xffffffffu;                          // (at most SMALL_CAND_CAP candidates in all)
#pragma unroll
    for (int i = 0; i < 5; i++) acc[lane][i] = 0u;
    __builtin_amdgcn_wave_barrier();
    auto read_of = [&](uint32_t krel) {                                          // the first read whose range ends behind candidate A + krel
        uint32_t lo = 0, hi = nvalid - 1;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool gt = sb[mid] > krel;
            hi = gt ? mid : hi; lo = gt ? lo : (mid + 1 < hi ? mid + 1 : hi);
        }
        return lo;
    };
    if (wave_on) {
        for (uint64_t k0 = A; k0 < B; k0 += 64) {
            const uint64_t k = k0 + lane;
            if (k < B) {
                const int32_t cs = c_status[k];
                const uint32_t clf = c_ciglen[k];
                const uint32_t L = read_of((uint32_t)(k - A));
                if (cs == CS_OK) {
                    atomicAdd(&acc[L][0], 1u); atomicAdd(&acc[L][3], clf & ~CIG_DIAG);
                    if (!(clf & CIG_DIAG)) { atomicAdd(&acc[L][1], 1u); atomicAdd(&acc[L][2], small_text_bytes(clf)); }
                } else if (cs == CS_WINDOW_OOR) atomicOr(&acc[L][4], 1u);
                else if (cs != CS_NO_MATCH) atomicOr(&ctr->err_flags, cs == CS_CIG_OVERFLOW ? 2ull : 4ull);      // (as hit_count_one)
            }
        }
        __builtin_amdgcn_wave_barrier();
        nh = acc[lane][0]; nx = acc[lane][1]; xcb = acc[lane][2]; cbr = acc[lane][3];
        if (acc[lane][4]) st = -2;
    }
    if (st == -2) { nh = 0; nx = 0; xcb = 0; cbr = 0; }
    TMARK();
    uint64_t eh, ex, et, ec;
    const uint64_t th = block_scan_u64(nh, eh);
    __syncthreads();
    const uint64_t tx = block_scan_u64(nx, ex);
    __syncthreads();
    const uint64_t tt = block_scan_u64(xcb, et);
    __syncthreads();
    const uint64_t tc = block_scan_u64(cbr, ec);
    if (threadIdx.x == 0) {
        bt[blockIdx.x * 4 + 0] = th; bt[blockIdx.x * 4 + 1] = tx; bt[blockIdx.x * 4 + 2] = tt; bt[blockIdx.x * 4 + 3] = tc;
        __threadfence();
        atomicAdd(&ctr->pad_, 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(&ctr->pad_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { atomicOr(&ctr->err_flags, SMALL_ERR_CAP | 32ull); break; }
        }
    }
    __syncthreads();
    TMARK();
    uint64_t bh = 0, bx = 0, btx = 0, gh = 0, gx = 0, gt = 0, gc = 0;
    for (uint32_t k = 0; k < gridDim.x; k++) {
        const uint64_t v0 = __hip_atomic_load(&bt[k * 4 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v1 = __hip_atomic_load(&bt[k * 4 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       v2 = __hip_atomic_load(&bt[k * 4 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v3 = __hip_atomic_load(&bt[k * 4 + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < blockIdx.x) { bh += v0; bx += v1; btx += v2; }
        gh += v0; gx += v1; gt += v2; gc += v3;
    }
    const unsigned long long flags = __hip_atomic_load(&ctr->err_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool fits = gh <= SMALL_CAND_CAP && gx <= SMALL_CAND_CAP && gt <= cap_text && flags == 0;
    TMARK();
    if (blockIdx.x == 0 && threadIdx.x == 0) {                // (the header too is assembled in device memory and copied in lines)
        SmallHeader *const dh = (SmallHeader *)d_out;
        dh->n_hits = gh; dh->n_extra = gx; dh->xcig_bytes = gt; dh->n_cand = clus_off[n];
        Counters c = *ctr;
        c.err_flags = flags | (fits || flags ? 0ull : SMALL_ERR_CAP);
        c.cig_total = gc;
        dh->ctr = c;
    }
    TMARK();
    if (fits && valid) o_status[r] = (int8_t)st;
    const bool live = fits && valid && st == 0 && nh;
    // The hits, a lane per candidate again, from the top of the wave's range down (a read's hits go out in the reverse of its cluster
    // order: lane 0 has the round's highest candidate).  The read's place in the table comes by shuffles, the hit's place inside the
    // read from the lanes before it that belong to the same read (ballots; a scan of the text sizes) plus what the rounds before
    // carried over for the read that straddles the round's end.
    if (fits && wave_on) {
        const uint32_t ho32 = (uint32_t)(bh + eh), xo32 = (uint32_t)(bx + ex), co32 = (uint32_t)(btx + et);
        uint32_t carry_L = 0xffffffffu, ch = 0, cx = 0, ct = 0;
        for (uint64_t kend = B; kend > A; kend = kend - A > 64 ? kend - 64 : A) {
            const bool in = kend - A > lane;
            const uint64_t k = in ? kend - 1 - lane : A;
            const uint32_t krel = (uint32_t)(k - A);
            const uint32_t L = read_of(krel);
            const uint64_t bL = A + sb[L];
            const bool liveL = __shfl((int)live, (int)L, 64) != 0;
            const uint32_t hoL = (uint32_t)__shfl((int)ho32, (int)L, 64), xoL = (uint32_t)__shfl((int)xo32, (int)L, 64), coL = (uint32_t)__shfl((int)co32, (int)L, 64);
            const int32_t cs = c_status[k], sc = c_score[k];
            const uint32_t wl = c_wlen[k], clf = c_ciglen[k], cS = c_S[k];
            const bool ok = in && liveL && cs == CS_OK;
            const bool ex = ok && !(clf & CIG_DIAG);
            const unsigned long long mok = __ballot(ok), mex = __ballot(ex);
            const uint32_t first = kend > bL ? (uint32_t)(kend - bL) : 0u;          // the lane with the read's highest candidate of this round
            const unsigned long long seg = ((1ull << lane) - 1ull) & ~((1ull << first) - 1ull);
            const uint32_t tl = ex ? small_text_bytes(clf) : 0u, incl = dpp_scan_incl_u32(tl), excl = incl - tl;
            const uint32_t excl_first = (uint32_t)__shfl((int)excl, (int)first, 64);
            const bool mine = L == carry_L;
            if (ok) hit_put_small(r0 + L, k, (uint64_t)hoL + (mine ? ch : 0u) + (uint32_t)__popcll(mok & seg), (uint64_t)xoL + (mine ? cx : 0u) + (uint32_t)__popcll(mex & seg),
                                  (uint64_t)coL + (mine ? ct : 0u) + (excl - excl_first), wl, clf, shift, sc, cS, cig, cig_stride, o_hits, o_extra, o_cig);
            if (kend - A > 63) {                                                     // a full round: the read of its last lane may go on in the next
                const uint32_t L63 = (uint32_t)__builtin_amdgcn_readlane((int)L, 63), f63 = (uint32_t)__builtin_amdgcn_readlane((int)first, 63);
                const unsigned long long s63 = ~((1ull << f63) - 1ull);
                const uint32_t add_h = (uint32_t)__popcll(mok & s63), add_x = (uint32_t)__popcll(mex & s63);
                const uint32_t add_t = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) - (uint32_t)__shfl((int)excl, (int)f63, 64);
                if (L63 == carry_L) { ch += add_h; cx += add_x; ct += add_t; }
                else { carry_L = L63; ch = add_h; cx = add_x; ct = add_t; }
            }
        }
    }
    // ---- every block's part of the table is in d_out: across the link in whole lines
    TMARK();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(&ctr->pad_, 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(&ctr->pad_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 2u * gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) break;                     // (flagged at the first barrier already if blocks are missing)
        }
    }
    __syncthreads();
    TMARK();
    {
        const uint64_t hdr8 = offsetof(SmallHeader, done) / 8;                      // everything in front of the `done` word
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hdr8; i += (uint64_t)gridDim.x * blockDim.x)
            ((unsigned long long *)h_out)[i] = ((const unsigned long long *)d_out)[i];
    }
    if (fits) {
        const uint64_t u_st = ((uint64_t)n + 15) >> 4, u_h = gh, u_x = gx, u_t = gt >> 4;             // 16-byte units per section
        const uint64_t total_u = u_st + u_h + u_x + u_t;
        for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_u; u += (uint64_t)gridDim.x * blockDim.x) {
            uint64_t k = u;
            uint32_t sec = off_status;
            if (k >= u_st) { k -= u_st; sec = off_hits; if (k >= u_h) { k -= u_h; sec = off_extra; if (k >= u_x) { k -= u_x; sec = off_cig; } } }
            const uint4 v = ((const uint4 *)(d_out + sec))[k];
            ((uint4 *)(h_out + sec))[k] = v;
        }
    }
    __syncthreads();
    TMARK();
    if (threadIdx.x == 0) {
        __threadfence_system();
#ifdef SMALL_HITS_TIMING
        TT[tti++] = wall_clock64();
        if (blockIdx.x == 0) for (int i = 0; i < tti; i++) bt[64 + i] = TT[i];
        __threadfence();
#endif
        if (atomicAdd(&ctr->pad_, 1u) == 3u * gridDim.x - 1u) { __threadfence_system(); *(volatile unsigned long long *)&o_hdr->done = *d_seq; }
    }
#undef TMARK
}

// what the launches of a small batch depend on besides the contents of the pinned block: a graph captured for one key is replayed
// only for the same key.  sig: the device addresses the launches were given (small_ptr_sig: the handle's buffers have fixed
// capacities, so they only move when the general path ran on the handle -- which drops the graph anyway -- and the index's arrays
// when its table is rebuilt).
struct SmallKey { uint32_t n, max_lq, ulen, cig_stride; uint64_t total, sig; bwt_options opt; };

struct bwt_small {
    bwt_index *ix = nullptr;
    bwt_batch *b = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr;          // pinned
    size_t in_off_bytes = 0, in_bytes = 0, out_bytes = 0;
    size_t o_status = 0, o_hits = 0, o_extra = 0, o_cig = 0;     // section offsets inside h_out
    uint64_t cap_text = 0;
    bool sized = false;
    uint32_t sized_cig_stride = 0;
    // what the general path returned when the call was redone through it
    std::vector<int8_t> g_status; std::vector<bwt_hit_rec> g_hits; std::vector<bwt_hit_extra> g_extra; std::vector<char> g_cig;
    uint64_t calls = 0, fallbacks = 0;
    DevBuf dseq;                                        // the call number on the device (k_small_in -> k_small_hits)
    DevBuf dout;                                        // the result block as the hit kernel assembles it (same layout as h_out)
    // a batch of the same shape as the one before it replays the launches as ONE hipGraph (profiles/r4/launch_probe.txt: ~2.3 us per
    // kernel inside a graph against ~4.5 us for launches that follow each other on a stream)
    SmallKey last_key{}, graph_key{};
    bool have_last = false, graph_broken = false;
    hipGraphExec_t gexec = nullptr;
    uint64_t graph_launches = 0, graph_captures = 0;
};

extern "C" int bwt_small_create(bwt_index *ix, bwt_small **out) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_small_create: null argument");
    HIPCHK(hipSetDevice(ix->device));
    bwt_small *s = new bwt_small();
    s->ix = ix;
    int rc = bwt_batch_create(ix, &s->b);
    if (rc) { delete s; return rc; }
    // CIGAR slots of 128 bytes from the start: the general path begins at 32 and redoes a batch with wider slots when a text does
    // not fit -- here that would be a whole second run through the general path for the first read with two indels
    s->b->cig_stride = 128;
    s->in_off_bytes = (SMALL_SEQ_OFF + 8 + 15) & ~(size_t)15;
    s->in_bytes = s->in_off_bytes + SMALL_MAX_BASES + 64;
    s->cap_text = (uint64_t)SMALL_CAND_CAP * 32;
    s->o_status = (sizeof(SmallHeader) + 63) & ~(size_t)63;
    s->o_hits = s->o_status + SMALL_MAX_READS + 64;
    s->o_extra = s->o_hits + (size_t)SMALL_CAND_CAP * 16;
    s->o_cig = s->o_extra + (size_t)SMALL_CAND_CAP * 16;
    s->out_bytes = s->o_cig + s->cap_text + 64;
    if (hipHostMalloc((void **)&s->h_in, s->in_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&s->h_out, s->out_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        bwt_small_destroy(s);
        return fail(BWT_ENOMEM, "bwt_small_create: hipHostMalloc of the pinned staging blocks failed");
    }
    memset(s->h_out, 0, sizeof(SmallHeader));
    if ((rc = s->dseq.ensure(64)) || (rc = s->dout.ensure(s->out_bytes))) { bwt_small_destroy(s); return rc; }
    if (hipFuncSetAttribute((const void *)k_sw_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
    *out = s;
    return 0;
}

extern "C" void bwt_small_destroy(bwt_small *s) {
    if (!s) return;
    if (s->ix) (void)hipSetDevice(s->ix->device);
    if (s->b && s->b->stream) (void)hipStreamSynchronize(s->b->stream);
    if (s->gexec) (void)hipGraphExecDestroy(s->gexec);
    s->dseq.release(); s->dout.release();
    if (s->b) bwt_batch_destroy(s->b);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    delete s;
}

extern "C" int bwt_small_set_options(bwt_small *s, const bwt_options *opt) {
    if (!s) return fail(BWT_EINVAL, "bwt_small_set_options: null handle");
    return bwt_batch_set_options(s->b, opt);
}
extern "C" int bwt_small_profile(bwt_small *s, int enable) { return s ? bwt_batch_profile(s->b, enable) : fail(BWT_EINVAL, "bwt_small_profile: null handle"); }
extern "C" int bwt_small_kernel_times(bwt_small *s, char *names, uint32_t names_cap, double *ms, uint64_t *launches, uint32_t cap) {
    return s ? bwt_batch_kernel_times(s->b, names, names_cap, ms, launches, cap) : fail(BWT_EINVAL, "bwt_small_kernel_times: null handle");
}

// widest window k_sw_wave's LDS (up to 150 KB per block of four waves: bwt_small_create raises the kernel's limit) holds next to
// the directions of max_lq rows
static uint32_t small_wave_wcap(uint32_t max_lq) {
    uint32_t w = 0;
    while (w + 16 <= 4080 && ((uint64_t)(w + 16 + 1) + (uint64_t)max_lq * swv_dwords(w + 16)) * 16 <= 150 * 1024) w += 16;
    return w;
}

// the addresses a captured run holds: every buffer small_size_buffers sizes, the counters, the index's arrays and scalars
static uint64_t small_ptr_sig(const bwt_small *s) {
    const bwt_batch *b = s->b;
    const DevBuf *bufs[] = {&b->bases, &b->off, &b->rplanes, &b->t_def, &b->t_sp, &b->t_ep, &b->t_info, &b->t_cnt, &b->r_P, &b->r_qty, &b->r_pend,
                            &b->r_K, &b->r_cl, &b->bigq, &b->mid2q, &b->mid4q, &b->gq, &b->seed_off, &b->clus_off, &b->pos, &b->c_read, &b->c_S,
                            &b->c_D, &b->c_qoff, &b->c_qlen, &b->c_woff, &b->c_wlen, &b->c_status, &b->c_score, &b->c_ciglen, &b->fpq, &b->tq,
                            &b->dq, &b->tq2, &b->pfq, &b->cig, &b->scan_tmp, &b->slab_bnd_t, &b->slab_dir, &b->slab_bnd_fp, &b->slab_bndv,
                            &b->slab_dir_fp, &b->ctr, &s->dseq, &s->dout};
    uint64_t h = 0x9E3779B97F4A7C15ull;
    auto mix = [&](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); };
    for (const DevBuf *d : bufs) mix((uint64_t)(uintptr_t)d->p);
    const unsigned char *ib = (const unsigned char *)&s->ix->d;
    for (size_t i = 0; i + 8 <= sizeof(DevIndex); i += 8) { uint64_t v; memcpy(&v, ib + i, 8); mix(v); }
    mix((uint64_t)s->ix->shift);
    return h;
}

// every device buffer of the run at its fixed capacity (a compare each once they are there)
static int small_size_buffers(bwt_small *s, uint32_t maxT, uint32_t max_lq) {
    bwt_batch *b = s->b;
    const size_t n = SMALL_MAX_READS, T = (size_t)n * maxT, nc = (size_t)SMALL_CAND_CAP + 1;
    int rc;
    if ((rc = b->bases.ensure(SMALL_MAX_BASES + 64)) || (rc = b->off.ensure((n + 1) * 8)) || (rc = b->rplanes.ensure(rp_words(SMALL_MAX_BASES) * 4)) ||
        (rc = b->t_def.ensure(T * 4 + 4)) || (rc = b->t_sp.ensure(T * 4 + 4)) || (rc = b->t_ep.ensure(T * 4 + 4)) ||
        (rc = b->t_info.ensure(T * 4 + 4)) || (rc = b->t_cnt.ensure(T * 4 + 4)) ||
        (rc = b->r_P.ensure(n * 4 + 4)) || (rc = b->r_qty.ensure(n * 4 + 4)) || (rc = b->r_pend.ensure(n * 8 + 8)) ||
        (rc = b->r_K.ensure(n * 4 + 4)) || (rc = b->r_cl.ensure(n * 8 * CL_STAGE + 16)) || (rc = b->bigq.ensure(n * 4 + 4)) ||
        (rc = b->mid2q.ensure(n * 4 + 4)) || (rc = b->mid4q.ensure(n * 4 + 4)) || (rc = b->gq.ensure((n + 2) * 4 + 256)) ||
        (rc = b->seed_off.ensure((n + 1) * 8)) || (rc = b->clus_off.ensure((n + 1) * 8)) || (rc = b->pos.ensure((size_t)SMALL_SEED_CAP * 4 + 16)) ||
        (rc = b->c_read.ensure(nc * 4)) || (rc = b->c_S.ensure(nc * 4)) || (rc = b->c_D.ensure(nc * 4)) || (rc = b->c_qoff.ensure(nc * 8)) ||
        (rc = b->c_qlen.ensure(nc * 4)) || (rc = b->c_woff.ensure(nc * 8)) || (rc = b->c_wlen.ensure(nc * 4)) ||
        (rc = b->c_status.ensure(nc * 4 + 4)) || (rc = b->c_score.ensure(nc * 4 + 4)) || (rc = b->c_ciglen.ensure(nc * 4 + 4)) ||
        (rc = b->fpq.ensure(nc * 4 + 4)) || (rc = b->tq.ensure(nc * 4 + 4)) || (rc = b->dq.ensure(nc * 4 + 4)) || (rc = b->tq2.ensure(nc * 4 + 4)) ||
        (rc = b->pfq.ensure(nc * 4 + 256)) || (rc = b->cig.ensure(nc * b->cig_stride + 16)) ||
        (rc = b->scan_tmp.ensure(SMALL_HIT_BLOCKS * 4 * 8 + 64 + 128)))
        return rc;
    // the generic traced / fp64 kernels: 16 waves each, windows of up to SMALL_W_TRACE bases
    const uint64_t maxW1 = SMALL_W_TRACE + 1, strips = (max_lq + SW_R - 1) / SW_R, waves = 16;
    if ((rc = b->slab_bnd_t.ensure(waves * maxW1 * 64 * 4)) || (rc = b->slab_dir.ensure(waves * maxW1 * 64 * 4 * strips)) ||
        (rc = b->slab_bnd_fp.ensure(waves * maxW1 * 64 * 4)) || (rc = b->slab_bndv.ensure(waves * maxW1 * 64 * 8)) ||
        (rc = b->slab_dir_fp.ensure(waves * maxW1 * 64 * 4 * strips)))
        return rc;
    return 0;
}

// the kernels of one small batch, no host round trip in between; the reads are in s->h_in
static int small_launch(bwt_small *s, uint32_t n, uint32_t max_lq, uint32_t ulen, uint64_t total) {
    bwt_batch *b = s->b;
    bwt_index *ix = s->ix;
    const uint32_t maxT = std::max<uint32_t>(1, max_lq / BWT_MIN_LEN);
    int rc = small_size_buffers(s, maxT, max_lq);
    if (rc) return rc;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
    b->n = n; b->max_lq = max_lq; b->maxT = maxT; b->ulen = ulen; b->total_bases = total;
    b->planes_valid = true; b->ascii_valid = true; b->done = 0; b->n_cand = 0; b->n_hits = 0; b->cig_bytes = 0; b->n_extra = 0; b->xcig_bytes = 0;
    b->expanded = false; b->cand_pending = false; b->desc_partial = false; b->band_ran = false; b->wave_upto = SMALL_CAND_CAP;
    Counters *dc = b->ctr.as<Counters>();
    {
        SmallZero z{};
        z.p[0] = (uint32_t *)dc; z.words[0] = (uint32_t)(sizeof(Counters) / 4);
        z.p[1] = b->r_K.as<uint32_t>(); z.words[1] = n;
        z.p[2] = (uint32_t *)b->r_pend.p; z.words[2] = 2 * n;
        z.p[3] = b->r_P.as<uint32_t>(); z.words[3] = n;
        const uint64_t nseg = (((total + 15) >> 4) + 1) & ~1ull;
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nseg + 255) / 256, (uint64_t)cus * 2));
        KLAUNCH(b, "small_in", k_small_in, grid, 256, (const uint64_t *)s->h_in, (const uint8_t *)(s->h_in + s->in_off_bytes), n, total,
                b->off.as<uint64_t>(), b->bases.as<uint8_t>(), b->rplanes.as<uint32_t>(), rp_words(total), z,
                (const unsigned long long *)(s->h_in + SMALL_SEQ_OFF), s->dseq.as<unsigned long long>());
    }
    // ---- seed search (+ clustering where the table's top level carries the seeds): the general path's launches
    if ((rc = launch_search(b, n, maxT, false, true, true))) return rc;
    // ---- what the seed kernels left for the cluster kernels; queue lengths are read on the device
    if (!b->counts_done)
        KLAUNCH(b, "read_counts", k_read_counts, std::min<uint32_t>(cdiv(n, 256), 2048), 256, n, maxT, b->t_info.as<uint32_t>(),
                b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->bigq.as<uint32_t>(),
                b->mid2q.as<uint32_t>(), b->mid4q.as<uint32_t>(), dc);
    {
        SmallScan a{};
        a.in[0] = b->r_P.as<uint32_t>(); a.out[0] = b->seed_off.as<uint64_t>(); a.cap[0] = SMALL_SEED_CAP;
        KLAUNCH(b, "scan", k_small_scan, 1, 256, a, n, dc);
    }
    KLAUNCH(b, "cluster_small", k_cluster_small, cdiv((uint64_t)n * 64, 256), 256, ix->d, n, maxT,
            b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(),
            b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(), b->pos.as<uint32_t>(),
            b->r_K.as<uint32_t>(), (const uint32_t *)nullptr, (const unsigned int *)nullptr);
    KLAUNCH(b, "cluster_rest", k_cluster_rest, std::min<uint32_t>(64, std::max<uint32_t>(1, n / 4)), 256, ix->d, maxT, (const uint32_t *)b->mid2q.as<uint32_t>(),
            (const uint32_t *)b->mid4q.as<uint32_t>(), (const uint32_t *)b->bigq.as<uint32_t>(), (const Counters *)dc, b->t_def.as<uint32_t>(),
            b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(),
            b->seed_off.as<uint64_t>(), b->pos.as<uint32_t>(), b->r_K.as<uint32_t>());
    {
        SmallScan a{};
        a.in[0] = b->r_K.as<uint32_t>(); a.out[0] = b->clus_off.as<uint64_t>(); a.cap[0] = SMALL_CAND_CAP;
        KLAUNCH(b, "scan", k_small_scan, 1, 256, a, n, dc);
    }
    // ---- extend: the fused prefilter (window arithmetic on the way), then the DP kernels of a short queue
    SwIO io;
    io.qbuf = b->bases.as<uint8_t>(); io.rbuf = ix->d.ref1; io.rbuf_len = ix->d.ref1_len;
    io.c_qoff = b->c_qoff.as<uint64_t>(); io.c_qlen = b->c_qlen.as<uint32_t>();
    io.c_woff = b->c_woff.as<int64_t>(); io.c_wlen = b->c_wlen.as<uint32_t>();
    io.c_status = b->c_status.as<int32_t>(); io.c_score = b->c_score.as<int32_t>();
    io.c_ciglen = b->c_ciglen.as<uint32_t>(); io.cig = b->cig.as<uint8_t>(); io.cig_stride = b->cig_stride;
    io.fpq = b->fpq.as<uint32_t>(); io.tq = b->tq.as<uint32_t>(); io.ctr = dc;
    io.dq = b->dq.as<uint32_t>(); io.tq2 = b->tq2.as<uint32_t>(); io.qplanes = b->rplanes.as<uint32_t>(); io.wplanes = ix->d.rplanes;
    io.wave_upto = SMALL_CAND_CAP; io.tq_in = nullptr; io.tq_in_count = nullptr;
    {
        CandSrc src{};
        src.off = b->off.as<uint64_t>(); src.r_K = b->r_K.as<uint32_t>(); src.seed_off = b->seed_off.as<uint64_t>();
        src.clus_off = b->clus_off.as<uint64_t>(); src.pos = b->pos.as<uint32_t>(); src.r_qty = b->r_qty.as<uint32_t>();
        src.r_cl = b->r_cl.as<uint32_t>(); src.c_S = b->c_S.as<uint32_t>(); src.c_D = b->c_D.as<uint32_t>(); src.c_qoff = b->c_qoff.as<uint64_t>();
        src.c_qlen = b->c_qlen.as<uint32_t>(); src.c_woff = b->c_woff.as<int64_t>(); src.c_wlen = b->c_wlen.as<uint32_t>();
        src.ovq = b->pfq.as<uint32_t>(); src.ov_count = &dc->pf_ov;
        src.ulen = ulen;
        // The fused pass tests NO candidate itself here (kcap 0): it only writes every candidate's descriptor and lists it, and the
        // list pass tests them all, a quad per candidate -- 6.9 + 6.4 us for a list of 200 reads where the general path's split (two
        // per read in the fused pass) took 12.2 + 9.0 and everything in the fused pass 40: a wave of the fused pass holds sixteen
        // reads and loops to the largest cluster count among them, each turn a round trip to the window planes.
        src.kcap = 1;
        const uint32_t pf_grid = (uint32_t)std::min<uint64_t>(cdiv((uint64_t)n * 4, 256), (uint64_t)cus * 5);
        KLAUNCH(b, "sw_prefilter", k_sw_prefilter<true>, pf_grid, 256, io, n, src);
        CandSrc rest{};
        rest.list = b->pfq.as<uint32_t>(); rest.list_count = &dc->pf_ov;
        KLAUNCH(b, "sw_prefilter", k_sw_prefilter<false>, std::min<uint32_t>((uint32_t)cus * 4, cdiv((uint64_t)n * 12, 256)), 256, io, SMALL_CAND_CAP, rest);
    }
    {
        // k_sw_wave over the whole DP queue (never longer than its limit here), twice: first with LDS for windows up to 96 bases wider
        // than the longest read (at the most LDS a block can have a CU runs one block, and a batch of 2000 reads takes rounds of
        // alignments), then on what that pass found too wide with all the LDS there is -- only what is wider still goes to the
        // lane-per-alignment traced kernel, where a single window of a few hundred bases is a millisecond
        const uint32_t wfull = small_wave_wcap(max_lq);
        const uint32_t wcap = std::min<uint32_t>(wfull, ((max_lq + 96 + 15) / 16) * 16);
        const uint32_t grid = std::min<uint32_t>(1024, std::max<uint32_t>(64, n));
        ProfScope ps(b, "sw_wave");
        // (one list of the reference's 200 reads: a few hundred candidates, fewer than the 1024 waves that run side by side with all
        // the LDS a block can have -- one pass, one launch less)
        if (n <= SMALL_ONE_PASS)
            hipLaunchKernelGGL(k_sw_wave, dim3(256), dim3(256), (size_t)(((uint64_t)wfull + 1 + (uint64_t)max_lq * swv_dwords(wfull)) * 16), b->stream, io,
                               SMALL_CAND_CAP, wfull + 1, swv_dwords(wfull), max_lq, (const uint32_t *)nullptr, (const unsigned int *)nullptr,
                               (uint32_t *)nullptr, (unsigned int *)nullptr);
        else {
        hipLaunchKernelGGL(k_sw_wave, dim3(grid), dim3(256), (size_t)(((uint64_t)wcap + 1 + (uint64_t)max_lq * swv_dwords(wcap)) * 16), b->stream, io,
                           SMALL_CAND_CAP, wcap + 1, swv_dwords(wcap), max_lq, (const uint32_t *)nullptr, (const unsigned int *)nullptr,
                           b->tq2.as<uint32_t>(), &dc->tq2_count);
        hipLaunchKernelGGL(k_sw_wave, dim3(std::min<uint32_t>(grid, 256)), dim3(256), (size_t)(((uint64_t)wfull + 1 + (uint64_t)max_lq * swv_dwords(wfull)) * 16),
                           b->stream, io, SMALL_CAND_CAP, wfull + 1, swv_dwords(wfull), max_lq, (const uint32_t *)b->tq2.as<uint32_t>(),
                           (const unsigned int *)&dc->tq2_count, (uint32_t *)nullptr, (unsigned int *)nullptr);
        }
    }
    {
        // what is left: windows wider than any LDS (traced integer kernel) and candidates with a base outside ACGT (fp64), one launch
        const uint32_t maxW1 = SMALL_W_TRACE + 1, strips = (max_lq + SW_R - 1) / SW_R;
        ProfScope ps(b, "sw_tail");
        hipLaunchKernelGGL(k_sw_tail, dim3(4, 2), dim3(256), 0, b->stream, io, SMALL_CAND_CAP, maxW1, strips, b->slab_bnd_t.as<uint32_t>(),
                           b->slab_dir.as<uint32_t>(), b->slab_bnd_fp.as<uint32_t>(), b->slab_bndv.as<double>(), b->slab_dir_fp.as<uint32_t>());
    }
    // ---- the hit table, written straight into the pinned result block
    KLAUNCH(b, "small_hits", k_small_hits, cdiv(n, 256), 256, n, ix->shift, (const uint64_t *)b->clus_off.as<uint64_t>(), (const int32_t *)b->c_status.as<int32_t>(),
            (const int32_t *)b->c_score.as<int32_t>(), (const uint32_t *)b->c_ciglen.as<uint32_t>(), (const uint32_t *)b->c_S.as<uint32_t>(),
            (const uint32_t *)b->c_wlen.as<uint32_t>(), (const uint8_t *)b->cig.as<uint8_t>(), b->cig_stride, (const uint32_t *)b->r_qty.as<uint32_t>(), dc,
            b->scan_tmp.as<unsigned long long>(), s->h_out, s->dout.as<uint8_t>(), (uint32_t)s->o_status, (uint32_t)s->o_hits, (uint32_t)s->o_extra,
            (uint32_t)s->o_cig, s->cap_text, (const unsigned long long *)s->dseq.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int bwt_small_align(bwt_small *s, const uint8_t *bases, const uint64_t *off, uint32_t n, bwt_small_result *res) {
    if (!s || !res || !off || (!bases && n)) return fail(BWT_EINVAL, "bwt_small_align: null argument");
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto us_since = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
    memset(res, 0, sizeof *res);
    res->n_reads = n;
    s->calls++;
    bwt_batch *b = s->b;
    // read lengths on the host (a few thousand compares): what the general path asks a kernel and a round trip for
    uint32_t maxlq = 0, minlq = 0xffffffffu;
    bool bad = false, longr = false;
    for (uint32_t i = 0; i < n; i++) {
        if (off[i + 1] < off[i]) { bad = true; break; }
        const uint64_t l = off[i + 1] - off[i];
        if (l > 65535) { longr = true; break; }
        maxlq = l > maxlq ? (uint32_t)l : maxlq; minlq = l < minlq ? (uint32_t)l : minlq;
    }
    if (bad) return fail(BWT_EINVAL, "bwt_small_align: offsets must be non-decreasing");
    if (longr) return fail(BWT_EINVAL, "bwt_small_align: read longer than 65535 bases");
    const uint64_t total = n ? off[n] - off[0] : 0;
    const uint32_t ulen = (n && minlq == maxlq) ? maxlq : 0;
    const bool eligible = n >= 1 && n <= SMALL_MAX_READS && total <= SMALL_MAX_BASES && maxlq >= 1 &&
                          small_wave_wcap(maxlq) >= maxlq + 64 && b->opt.sw_prefilter != 0 && b->opt.sw_wave != 0 && b->opt.sw_fuse_cand != 0;
    bool small_ok = false;
    HIPCHK(hipSetDevice(b->device));
    if (eligible) {
        std::shared_lock<std::shared_mutex> lk(s->ix->mu);
        uint64_t *ho = (uint64_t *)s->h_in;
        const uint64_t o0 = off[0];
        for (uint32_t i = 0; i <= n; i++) ho[i] = off[i] - o0;
        memcpy(s->h_in + s->in_off_bytes, bases + o0, total);
        *(unsigned long long *)(s->h_in + SMALL_SEQ_OFF) = (unsigned long long)s->calls;
        res->us_copy_in = us_since(t0);
        int rc = small_size_buffers(s, std::max<uint32_t>(1, maxlq / BWT_MIN_LEN), maxlq);
        if (rc) return rc;
        // the second batch in a row of one shape captures its launches into a hipGraph, the ones after it replay that graph; anything
        // that changes what a launch was given (another shape, options, a device buffer of the library moved) drops it
        SmallKey key;
        memset(&key, 0, sizeof key);
        key.n = n; key.max_lq = maxlq; key.ulen = ulen; key.cig_stride = b->cig_stride; key.total = total;
        key.sig = small_ptr_sig(s); key.opt = b->opt;
        const bool graph_on = b->opt.small_graph != 0 && !b->prof && !s->graph_broken;
        const bool same_as_last = s->have_last && !memcmp(&key, &s->last_key, sizeof key);
        s->last_key = key; s->have_last = true;
        if (s->gexec && (!graph_on || memcmp(&key, &s->graph_key, sizeof key))) { (void)hipGraphExecDestroy(s->gexec); s->gexec = nullptr; }
        bool launched = false;
        if (s->gexec) {
            if (hipGraphLaunch(s->gexec, b->stream) == hipSuccess) { launched = true; s->graph_launches++; res->graph = 1; }
            else { (void)hipGetLastError(); (void)hipGraphExecDestroy(s->gexec); s->gexec = nullptr; s->graph_broken = true; }
        } else if (graph_on && same_as_last) {
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                rc = small_launch(s, n, maxlq, ulen, total);
                const hipError_t e = hipStreamEndCapture(b->stream, &g);
                if (!rc && e == hipSuccess && g && hipGraphInstantiate(&s->gexec, g, nullptr, nullptr, 0) == hipSuccess) {
                    s->graph_key = key; s->graph_captures++;
                    if (hipGraphLaunch(s->gexec, b->stream) == hipSuccess) { launched = true; s->graph_launches++; res->graph = 2; }
                }
                if (g) (void)hipGraphDestroy(g);
                if (!launched) {
                    (void)hipGetLastError();
                    if (s->gexec) { (void)hipGraphExecDestroy(s->gexec); s->gexec = nullptr; }
                    s->graph_broken = true; rc = 0;
                }
                // shapes that keep changing after two batches each: capturing costs more than it saves
                if (s->graph_captures >= 16 && s->graph_launches < 4 * s->graph_captures) s->graph_broken = true;
            } else { (void)hipGetLastError(); s->graph_broken = true; }
        }
        if (!launched) rc = small_launch(s, n, maxlq, ulen, total);
        res->us_enqueue = us_since(t0);
        if (rc) { (void)hipStreamSynchronize(b->stream); return rc; }
        const SmallHeader *h = (const SmallHeader *)s->h_out;
        if (b->prof) HIPCHK(hipStreamSynchronize(b->stream));       // (profiling runs: the events want a synchronised stream anyway)
        else {
            // spin on the `done` word the last block of the last kernel writes (bounded: then the stream is asked)
            const volatile unsigned long long *done = &h->done;
            const auto tw = clk::now();
            uint32_t polls = 0;
            while (*done != (unsigned long long)s->calls) {
                __builtin_ia32_pause();
                if ((++polls & 0xffffu) == 0 && us_since(tw) > 2e6) { HIPCHK(hipStreamSynchronize(b->stream)); break; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        res->us_wait = us_since(t0);
        b->hctr = h->ctr;
#ifdef SMALL_HITS_TIMING
        if ((s->calls % 10) == 0) {
            unsigned long long tt[12];
            (void)hipStreamSynchronize(b->stream);
            (void)hipMemcpy(tt, b->scan_tmp.as<unsigned long long>() + 64, sizeof tt, hipMemcpyDeviceToHost);
            fprintf(stderr, "k_small_hits block 0, ticks of 10 ns: count %llu, scans+barrier %llu, totals %llu, header %llu, write %llu, barrier %llu, copy %llu, fence %llu\n", tt[1] - tt[0],
                    tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tt[6] - tt[5], tt[7] - tt[6], tt[8] - tt[7]);
        }
#endif
#ifdef SWV_TIMING
        if ((s->calls % 10) == 0) fprintf(stderr, "k_sw_wave slowest wave: prologue %u, DP %u, traceback %u ticks of 10 ns; widest window %u\n", h->ctr.pad2_[0], h->ctr.pad2_[1], h->ctr.pad2_[2], h->ctr.band0_next);
#endif
        res->overflow_flags = h->ctr.err_flags;
        if (h->ctr.err_flags == 0) {
            small_ok = true;
            b->n_cand = h->n_cand; b->n_hits = h->n_hits; b->n_extra = h->n_extra; b->xcig_bytes = h->xcig_bytes; b->cig_bytes = h->ctr.cig_total;
            res->path = 1;
            res->n_hits = h->n_hits; res->n_extra = h->n_extra; res->cigar_bytes = h->xcig_bytes;
            res->read_status = (const int8_t *)(s->h_out + s->o_status);
            res->hits = (const bwt_hit_rec *)(s->h_out + s->o_hits);
            res->extra = (const bwt_hit_extra *)(s->h_out + s->o_extra);
            res->cigar = (const char *)(s->h_out + s->o_cig);
            res->n_candidates = h->n_cand; res->n_dp = h->ctr.dq_count; res->n_float = h->ctr.fp_count;
        }
    }
    if (!small_ok) {
        // the general path (a capacity of this one did not hold, or the batch is not one it takes): same handle, same results
        if (eligible) s->fallbacks++;
        if (s->gexec) { (void)hipGraphExecDestroy(s->gexec); s->gexec = nullptr; }      // (the general path may move the handle's buffers)
        s->have_last = false;
        int rc;
        bwt_batch_counts c{};
        uint64_t need[3] = {0, 0, 0};
        if ((rc = bwt_batch_set_reads(b, bases, off, n)) || (rc = bwt_batch_run(b, BWT_STAGE_EXTEND, &c))) return rc;
        rc = bwt_batch_fetch_hits_compact(b, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, need);
        if (rc && rc != BWT_ENOSPC) return rc;
        s->g_status.resize((size_t)n + 1); s->g_hits.resize(need[0] + 1); s->g_extra.resize(need[1] + 1); s->g_cig.resize(need[2] + 1);
        if ((rc = bwt_batch_fetch_hits_compact(b, s->g_status.data(), s->g_hits.data(), s->g_extra.data(), s->g_cig.data(), need[0], need[1], need[2], need)))
            return rc;
        res->path = eligible ? 2 : 0;
        res->n_hits = need[0]; res->n_extra = need[1]; res->cigar_bytes = need[2];
        res->read_status = s->g_status.data(); res->hits = s->g_hits.data(); res->extra = s->g_extra.data(); res->cigar = s->g_cig.data();
        res->n_candidates = c.n_clusters; res->n_dp = c.n_dp; res->n_float = c.n_float;
    }
    res->us_total = us_since(t0);
    return 0;
}
