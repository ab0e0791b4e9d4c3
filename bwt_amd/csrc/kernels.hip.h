// Device kernels of libbwtgpu (gfx950 / CDNA4, wave64).  Included once by bwt_gpu.hip.
//
// Reference functions restated here (paths relative to /root/reference):
//   k_subreads        sga:skip_Ns/1,4 + the sub-read chain of sga:sga/9      src/sga.erl:18-34,69-76
//   lf_step, k_search, k_search_pending
//                     bwa:find_seeds/6,/5 + fmi:se_down/se_up in Occ form     src/bwa.erl:12-45, src/fmi.erl:25-39
//   kt_build          the same steps, once per k-mer (suffix-interval table)
//   k_seed_tasks      sub-reads + table lookups (+ clustering) per read       src/sga.erl:17-64, src/bwa.erl:12-45
//   k_seed_list       fmi:get_seed_ends/4,5                                   src/fmi.erl:89-100
//   k_cluster_*       sga:add_seeds/3 + sga:get_similar/2,7                   src/sga.erl:37-64
//   k_candidates      window slice of worker_loop(running,..)                 src/worker_bwt.erl:86-90
//   k_sw_prefilter, k_sw_pk<LQ,TRACE>, k_sw_score, k_sw<MODE,LDSB>
//                     sw:sw/3, build_tab, add_row2tab, sigma, get_CIGAR       src/sw.erl:47-202
//   k_hits_*          the Cigars fold (order, Pos) of worker_loop             src/worker_bwt.erl:84-97
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels_index.h"
#include "sw_cert.h"

// ------------------------------------------------------------------------------------------------
// wave helpers
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
// A wave-private LDS staging buffer (128 u32) in front of a global queue: pushes gather in LDS and take queue space 64
// entries at a time -- one same-address atomic per 64 entries instead of one per wave-iteration (they run at ~88 per us,
// which a kernel that queues most of its items would spend tens of ms on).  push/drain are reached by whole waves.
struct WaveQ { uint32_t *buf; uint32_t have; };
// FL: entries a flush takes queue space for (a multiple of 64; the buffer holds FL + 64).  A kernel that queues most of what it looks
// at -- the prefilter on a repeat-rich batch: 15 M DP candidates and 13 M list entries -- is bound by that one atomic per 64 entries
// (240 k same-address atomics = 2.7 ms): its queues flush 256 at a time.
template <int FL = 64>
__device__ __forceinline__ void waveq_out(WaveQ &q, uint32_t nout, uint32_t *__restrict__ gq, unsigned int *counter) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, nout);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)FL; k += 64)
        if (k + lane < nout) gq[base + k + lane] = q.buf[k + lane];
    const uint32_t rest = lane + nout < q.have ? q.buf[lane + nout] : 0u;          // (fewer than 64 entries are ever left behind a flush)
    __builtin_amdgcn_wave_barrier();
    if (lane + nout < q.have) q.buf[lane] = rest;
    q.have -= nout;
    __builtin_amdgcn_wave_barrier();
}
template <int FL = 64>
__device__ __forceinline__ void waveq_push(WaveQ &q, bool pred, uint32_t v, uint32_t *__restrict__ gq, unsigned int *counter) {
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63;
    if (pred) q.buf[q.have + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
    q.have += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (q.have >= (uint32_t)FL) waveq_out<FL>(q, (uint32_t)FL, gq, counter);
}
// (what is left at the end of the kernel: up to FL + 63 entries)
template <int FL = 64>
__device__ __forceinline__ void waveq_drain(WaveQ &q, uint32_t *__restrict__ gq, unsigned int *counter) {
    while (q.have) waveq_out<FL>(q, q.have < (uint32_t)FL ? q.have : (uint32_t)FL, gq, counter);
}
// Work counters at the end of a kernel: a wave's (wave-uniform) sums go into block-shared accumulators and ONE thread of the
// block adds them to the global counters.  Same-address global atomics run at ~88 per us on this part: a launch of a few hundred
// short waves -- a batch of the reference's message size -- that ends with four atomics per wave spent more time on its counters
// than on its work (k_seed_tasks 11 -> 23 us, k_search_pending 18 -> 58 us from 200 to 2000 reads).  Reached by EVERY thread of
// the block, once per kernel.  sums[k] -> 64-bit counter g64[k] (or the 32-bit one g32[k] where g64[k] is null); maxes likewise.
template <int NS, int NM>
__device__ __forceinline__ void block_counters(const unsigned long long (&sums)[NS], unsigned long long *const (&g64)[NS],
                                               unsigned int *const (&g32)[NS], const uint32_t (&maxes)[NM], unsigned int *const (&gmx)[NM]) {
    __shared__ unsigned long long s_[NS];
    __shared__ unsigned int m_[NM];
    if (threadIdx.x < (uint32_t)NS) s_[threadIdx.x] = 0;
    if (threadIdx.x < (uint32_t)NM) m_[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) if (sums[k]) atomicAdd(&s_[k], sums[k]);
#pragma unroll
        for (int k = 0; k < NM; k++) if (maxes[k]) atomicMax(&m_[k], maxes[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) if (s_[k]) { if (g64[k]) atomicAdd(g64[k], s_[k]); else atomicAdd(g32[k], (unsigned int)s_[k]); }
#pragma unroll
        for (int k = 0; k < NM; k++) if (m_[k]) atomicMax(gmx[k], m_[k]);
    }
}
__device__ __forceinline__ bool is_acgt(uint32_t c) {
    uint32_t i = c - 'A';   // A=0 C=2 G=6 T=19
    return i < 20u && ((0x80045u >> i) & 1u);
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort of one u32 per lane across the wave, ascending.  Stage (K, J): lane l meets lane l ^ J and keeps the
// smaller value iff ((l & J) == 0) == ((l & K) == 0) -- a compile-time lane mask, fed to v_cndmask as an SGPR pair.
// The partner comes by DPP quad_perm (J = 1, 2), ds_swizzle xor mode (J = 4, 8, 16) or ds_bpermute (J = 32).
template <int K, int J>
__device__ __forceinline__ constexpr unsigned long long bitonic_keep_min() {
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++)
        if (((l & J) == 0) == ((l & K) == 0)) m |= 1ull << l;
    return m;
}
template <int K, int J>
__device__ __forceinline__ uint32_t bitonic_stage(uint32_t p) {
    uint32_t o;
    if constexpr (J == 1) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J <= 16) o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)p, (J << 10) | 0x1f);         // lane ^ J inside 32
    else o = (uint32_t)__shfl_xor((int)p, J, 64);
    // keep the smaller of (p, o) where the stage's mask says so, else the larger: take the partner's value iff
    // (p < o) differs from the mask bit -- one compare into a lane mask, a scalar xor, one select (called by whole waves)
    constexpr unsigned long long M = bitonic_keep_min<K, J>();
    // (the select takes the lane mask through inverse_ballot, not inline asm: the compiler has to see the VALU write to
    // place the wait states a following DPP read of the result needs)
    const unsigned long long take = __ballot(p < o) ^ M;
    return __builtin_amdgcn_inverse_ballot_w64(take) ? o : p;
}
template <int K, int J>
__device__ __forceinline__ uint32_t bitonic_merge(uint32_t p) {
    p = bitonic_stage<K, J>(p);
    if constexpr (J > 1) p = bitonic_merge<K, J / 2>(p);
    return p;
}
template <int K>
__device__ __forceinline__ uint32_t bitonic_sort_upto(uint32_t p) {
    if constexpr (K > 2) p = bitonic_sort_upto<K / 2>(p);
    return bitonic_merge<K, K / 2>(p);
}
// two independent sorts side by side: each is a chain of 21 dependent stages (10 of them LDS round trips), interleaved
// the swizzle latency of one hides behind the VALU work of the other
template <int K, int J>
__device__ __forceinline__ void bitonic_merge2(uint32_t &a, uint32_t &b) {
    a = bitonic_stage<K, J>(a); b = bitonic_stage<K, J>(b);
    if constexpr (J > 1) bitonic_merge2<K, J / 2>(a, b);
}
template <int K>
__device__ __forceinline__ void bitonic_sort_upto2(uint32_t &a, uint32_t &b) {
    if constexpr (K > 2) bitonic_sort_upto2<K / 2>(a, b);
    bitonic_merge2<K, K / 2>(a, b);
}
// Sort of N registers (one u32 per lane each, N independent sorts side by side), ascending.  A bitonic
// network in its "mirror" form: the merge of two sorted runs of K/2 first pairs lane l with lane l ^ (K - 1), then halves the
// distance (l ^ K/4, l ^ K/8, ...), and in EVERY stage the lower lane of a pair keeps the smaller value.  The select then
// needs no compare and no lane mask in SGPRs: r = bfi(lower ? ~0 : 0, min(p, o), max(p, o)) with six per-lane constants
// (is bit b of my lane number clear?), and the partner comes folded into v_min_u32_dpp / v_max_u32_dpp for 15 of the 21
// stages (quad_perm, row_half_mirror, row_mirror, row_ror:8); the other six go through the LDS crossbar (ds_swizzle,
// one ds_bpermute).  3 VALU per DPP stage, 1 DS + 3 VALU otherwise, no SALU -- the compare / ballot / xor / select form of
// bitonic_stage costs 6 issue slots per stage, and this kernel is bound by instruction issue.
struct SortMasks { uint32_t m1, m2, m4, m8, m16, m32; int mirror63, x32; };
__device__ __forceinline__ SortMasks sort_masks() {
    const uint32_t l = threadIdx.x & 63u;
    SortMasks M;
    M.m1 = (l & 1u) ? 0u : ~0u; M.m2 = (l & 2u) ? 0u : ~0u; M.m4 = (l & 4u) ? 0u : ~0u; M.m8 = (l & 8u) ? 0u : ~0u;
    M.m16 = (l & 16u) ? 0u : ~0u; M.m32 = (l & 32u) ? 0u : ~0u;
    M.mirror63 = (int)((l ^ 63u) << 2); M.x32 = (int)((l ^ 32u) << 2);
    return M;
}
// (as ONE instruction: the median of {p, o, K} with K = 0 in the lanes that keep the smaller value -- nothing is below 0 --
// and K = ~0 in those that keep the larger; `lower` holds ~K.  v_med3_u32 comes out of the min / max form below.)
__device__ __forceinline__ uint32_t ms_keep(uint32_t p, uint32_t o, uint32_t lower) {
    const uint32_t K = ~lower;
    const uint32_t mn = p < o ? p : o, mx = p < o ? o : p;
    const uint32_t c = mx < K ? mx : K;
    return mn > c ? mn : c;
}
template <int CTRL, int N>
__device__ __forceinline__ void ms_dpp(uint32_t (&p)[N], uint32_t lower) {
#pragma unroll
    for (int t = 0; t < N; t++) p[t] = ms_keep(p[t], (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p[t], CTRL, 0xF, 0xF, true), lower);
}
template <int XOR, int N>
__device__ __forceinline__ void ms_swz(uint32_t (&p)[N], uint32_t lower) {          // partner lane ^ XOR inside 32 lanes
    uint32_t o[N];
#pragma unroll
    for (int t = 0; t < N; t++) o[t] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)p[t], (XOR << 10) | 0x1f);
#pragma unroll
    for (int t = 0; t < N; t++) p[t] = ms_keep(p[t], o[t], lower);
}
// a bitonic sequence of 64 (per register) into ascending order: the half-cleaners l ^ 32, 16, 8, 4, 2, 1
template <int N>
__device__ __forceinline__ void bitonic_merge64(uint32_t (&p)[N], const SortMasks &M) {
    enum { X1 = 0xB1, X2 = 0x4E, ROR8 = 0x128 };
    {
        uint32_t o[N];
#pragma unroll
        for (int t = 0; t < N; t++) o[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(M.x32, (int)p[t]);
#pragma unroll
        for (int t = 0; t < N; t++) p[t] = ms_keep(p[t], o[t], M.m32);
    }
    ms_swz<16>(p, M.m16); ms_dpp<ROR8>(p, M.m8); ms_swz<4>(p, M.m4); ms_dpp<X2>(p, M.m2); ms_dpp<X1>(p, M.m1);
}
template <int N>
__device__ __forceinline__ void mirror_sort64(uint32_t (&p)[N], const SortMasks &M) {
    enum { X1 = 0xB1, X2 = 0x4E, X3 = 0x1B, HALF_MIRROR = 0x141, ROW_MIRROR = 0x140, ROR8 = 0x128 };
    ms_dpp<X1>(p, M.m1);                                                                                  // runs of 2
    ms_dpp<X3>(p, M.m2); ms_dpp<X1>(p, M.m1);                                                             // 4
    ms_dpp<HALF_MIRROR>(p, M.m4); ms_dpp<X2>(p, M.m2); ms_dpp<X1>(p, M.m1);                               // 8
    ms_dpp<ROW_MIRROR>(p, M.m8); ms_swz<4>(p, M.m4); ms_dpp<X2>(p, M.m2); ms_dpp<X1>(p, M.m1);            // 16
    ms_swz<31>(p, M.m16); ms_dpp<ROR8>(p, M.m8); ms_swz<4>(p, M.m4); ms_dpp<X2>(p, M.m2); ms_dpp<X1>(p, M.m1);   // 32
    {
        uint32_t o[N];
#pragma unroll
        for (int t = 0; t < N; t++) o[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(M.mirror63, (int)p[t]);
#pragma unroll
        for (int t = 0; t < N; t++) p[t] = ms_keep(p[t], o[t], M.m32);
    }
    ms_swz<16>(p, M.m16); ms_dpp<ROR8>(p, M.m8); ms_swz<4>(p, M.m4); ms_dpp<X2>(p, M.m2); ms_dpp<X1>(p, M.m1);   // 64
}
// The general kernels keep the compare / ballot / select form: with the mirror form k_seed_tasks needs 127 VGPRs and spills
// SGPRs, and is slower (6.9 against 6.7 ms at GRCh38 scale, 12.6 against 12.0 ms on 150-bp reads).
#ifndef BWT_GENERIC_MIRROR_SORT
#define BWT_GENERIC_MIRROR_SORT 1
#endif
// (one register / two independent registers side by side)
__device__ __forceinline__ uint32_t wave_sort_u32(uint32_t p) {
#if BWT_GENERIC_MIRROR_SORT
    uint32_t a[1] = {p};
    mirror_sort64(a, sort_masks());
    return a[0];
#else
    return bitonic_sort_upto<64>(p);
#endif
}
__device__ __forceinline__ void wave_sort2_u32(uint32_t &a, uint32_t &b) {
#if BWT_GENERIC_MIRROR_SORT
    uint32_t v[2] = {a, b};
    mirror_sort64(v, sort_masks());
    a = v[0]; b = v[1];
#else
    bitonic_sort_upto2<64>(a, b);
#endif
}

// ------------------------------------------------------------------------------------------------
// exclusive scan  u32 -> u64   (3 kernels: tile sums, scan of the sums, tile scans + offset: the input is read
// twice, the output written once)
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

// sum of this block's 256 per-thread values in every thread's return; `excl` = sum of the threads before this one
__device__ __forceinline__ uint64_t block_scan_u64(uint64_t s, uint64_t &excl) {
    __shared__ uint64_t wsum[SCAN_BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t up = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint64_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; w++) { const uint64_t t = wsum[w]; if ((uint32_t)w < wv) before += t; total += t; }
    excl = before + incl - s;
    return total;
}
__device__ __forceinline__ void scan1_body(const uint32_t *__restrict__ in, uint64_t *__restrict__ bsum, uint64_t n) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
    if ((uint64_t)(blockIdx.x + 1) * SCAN_TILE <= n) {             // a full tile: two 16-byte loads per thread
        const uint4 a = *(const uint4 *)(in + base), b = *(const uint4 *)(in + base + 4);
        s = (uint64_t)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) s += base + k < n ? in[base + k] : 0u;
    }
    uint64_t excl;
    const uint64_t total = block_scan_u64(s, excl);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
// single block: exclusive scan of bsum[0..nb) in place, total to bsum[nb]
__device__ __forceinline__ void scan2_body(uint64_t *bsum, uint32_t nb) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += SCAN_BLOCK) {
        uint32_t i = c0 + threadIdx.x;
        uint64_t v = i < nb ? bsum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
            uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}
__device__ __forceinline__ void scan3_body(const uint32_t *__restrict__ in, uint64_t *__restrict__ out,
                                           const uint64_t *__restrict__ bsum, uint64_t n, uint32_t nb) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    const bool full = (uint64_t)(blockIdx.x + 1) * SCAN_TILE <= n;  // a full tile: 16-byte loads and stores
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
    if (full) {
        const uint4 a = *(const uint4 *)(in + base), b = *(const uint4 *)(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) s += v[k];
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    }
    uint64_t excl;
    (void)block_scan_u64(s, excl);
    uint64_t run = bsum[blockIdx.x] + excl;
    if (full) {
        uint64_t o[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = run; run += v[k]; }
        ulonglong2 *dst = (ulonglong2 *)(out + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS / 2; k++) dst[k] = make_ulonglong2(o[2 * k], o[2 * k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan1(const uint32_t *__restrict__ in, uint64_t *__restrict__ bsum, uint64_t n) { scan1_body(in, bsum, n); }
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan2(uint64_t *bsum, uint32_t nb) { scan2_body(bsum, nb); }
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan3(const uint32_t *__restrict__ in, uint64_t *__restrict__ out,
                                                       const uint64_t *__restrict__ bsum, uint64_t n, uint32_t nb) { scan3_body(in, out, bsum, n, nb); }
// two arrays of the same length in one set of launches (blockIdx.y picks the array): half the launches of two scans
struct ScanPair { const uint32_t *in[4]; uint64_t *out[4]; uint64_t *bsum[4]; };
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan1_pair(ScanPair p, uint64_t n) { scan1_body(p.in[blockIdx.y], p.bsum[blockIdx.y], n); }
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan2_pair(ScanPair p, uint32_t nb) { scan2_body(p.bsum[blockIdx.y], nb); }
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan3_pair(ScanPair p, uint64_t n, uint32_t nb) {
    scan3_body(p.in[blockIdx.y], p.out[blockIdx.y], p.bsum[blockIdx.y], n, nb);
}

// ... and when everything fits one tile (batches of up to 2048 reads: the size a worker's message has): one launch
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_one(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n) {
    const uint64_t base = (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint64_t excl;
    const uint64_t total = block_scan_u64(s, excl);
    uint64_t run = excl;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) out[n] = total;
}

// ------------------------------------------------------------------------------------------------
// k_subreads: one lane per read.  Enumerates the sub-reads sga:sga/9 will search:
//   {Skip,Qseq1} = skip_Ns(Qseq); search Qseq1; Qseq <- first (len(Qseq1)-8) bases; Shift += 8+Skip.
// skip_Ns (sga.erl:69-76) scans from the right with J counting since (and including) the last N and
// stops at J = 8: the sub-read ends at e = len, or right after an N, such that the 7 bases left of its
// last base hold no N.  On the is-N plane that is: while the 7-bit window [e-8, e-2] is non-zero, move e to
// just past its highest set bit.
// task word = end (exclusive, = len(Qseq1)) | (Shift+Skip) << 16 ; 0 = no task.
__global__ __launch_bounds__(256) void k_subreads(const uint32_t *__restrict__ planes,
                                                   const uint64_t *__restrict__ off, uint32_t n, uint32_t maxT,
                                                   uint32_t *__restrict__ t_def) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t P = off[r];
    int len = (int)(off[r + 1] - P);
    uint32_t shift = 0, t = 0;
    uint32_t *td = t_def + (uint64_t)r * maxT;
    for (;;) {
        if (len < BWT_MIN_LEN) break;                       // sga.erl:69
        int e = len;
        bool ok = true;
        for (;;) {
            if (e < BWT_MIN_LEN) { ok = false; break; }     // list exhausted before J reached 8 (sga.erl:76)
            const uint32_t w7 = plane_bits(planes, RP_N, P + (uint64_t)(e - 8)) & 0x7fu;
            if (!w7) break;
            e = e - 7 + (31 - __clz((int)w7));              // the sub-read now ends in that N
        }
        if (!ok) break;
        const int skipped = len - e;
        td[t++] = (uint32_t)e | ((shift + (uint32_t)skipped) << 16);
        len = e - BWT_MIN_LEN;
        shift += BWT_MIN_LEN + (uint32_t)skipped;
    }
    for (; t < maxT; t++) td[t] = 0;
}

// ------------------------------------------------------------------------------------------------
// Occ lookups on the rank-sampled BWT (block layout: kernels_index.h).  A quad (4 lanes) owns one task;
// lane q of the quad loads quarter q (16 B) of the 64-byte block, so one wave instruction fetches 16
// whole blocks with every 64-byte line requested exactly once, and all four lanes do the same work.
// sum over the 4 lanes of a quad (DPP quad_perm, no LDS traffic)
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    return v;
}
// lane T of the quad's value in all four lanes (DPP quad_perm [T,T,T,T])
template <int T>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, T * 0x55, 0xF, 0xF, true);
}
// match bits of this quarter's 48 rows against a code given as plane masks (nl = ~(-(c&1)), nh = ~(-(c>>1)))
__device__ __forceinline__ void quarter_match(uint4 v, uint32_t nl, uint32_t nh, uint32_t n16, uint32_t &y0, uint32_t &y1) {
    y0 = (v.y ^ nl) & (v.z ^ nh);
    const uint32_t x = v.w ^ n16;
    y1 = x & (x >> 16) & 0xffffu;
}
// matches among the first t (clamped to 0..48) rows of the quarter
__device__ __forceinline__ uint32_t quarter_prefix(uint32_t y0, uint32_t y1, int t) {
    const int tc = t < 0 ? 0 : (t > 48 ? 48 : t);
    const uint32_t mlo = tc >= 32 ? 0xffffffffu : ((1u << tc) - 1u);
    const int th = tc > 32 ? tc - 32 : 0;
    const uint32_t mhi = (1u << th) - 1u;
    return __popc(y0 & mlo) + __popc(y1 & mhi);
}

// k_search: quad per task, grid-stride.  Closed form of se_down / se_up:
//   found  <=>  occ(c,Ep-1) > occ(c,Sp-1);  Sp' = C0[c] + occ(c,Sp-1);  Ep' = C0[c] + occ(c,Ep) - 1
// (I is the 0-based LF target but is used as a 1-based row by the reference -- kept as is).
// occ(c,k) = cnt[k/192][c] + #{c among the first k%192 rows of block k/192}; the `$` row is stored
// as code 0 and taken out again.  Per LF step a quad issues two 16-byte-per-lane loads (blocks of
// Sp-1 and Ep-1) and three DPP quad reductions (occ(c,Sp-1), occ(c,Ep-1), the L[Ep]==c bit).
// U independent tasks per quad are advanced in lock step (their block loads are issued together), which
// doubles the bytes each wave keeps in flight: the step is a dependent chain of random 64-byte reads.
// The clause order of bwa:find_seeds/5 (bwa.erl:31-45) is folded into a per-task step limit:
//   n stops at MAX_LEN, at the start of the read, or at the first N (all three: range check applies),
//   or at the first base that is neither ACGT nor N (never equals any L: not_found, no range check).
struct LfTask {
    uint64_t task;
    uint32_t sp, ep, lo16, hi16, status;
    int n, nlim;
    bool check_at_limit, check, active;
};

// What bwa:find_seeds/6 does before the first LF step, for the sub-read that ends (exclusive) at read base e:
// the first range (bwa.erl:15-24), the plane windows of the (at most) 15 bases a search consumes -- [e-15, e),
// bit 15 of each 16-bit window = base e-1, base of step n at bit 15-n -- and the step limit.
__device__ __forceinline__ void task_init(LfTask &t, const DevIndex &ix, const uint32_t *__restrict__ planes,
                                          uint64_t endp, int e) {
    uint32_t n16b, o16;
    if (endp >= 16) {
        const uint4 w = planes4(planes, endp - 16);
        t.lo16 = w.x; t.hi16 = w.y; n16b = w.z; o16 = w.w;
    } else {                                             // only the very first read of a batch
        const uint32_t sh = 16 - (uint32_t)endp;
        const uint4 w = ((const uint4 *)planes)[0];
        t.lo16 = w.x << sh; t.hi16 = w.y << sh; n16b = w.z << sh; o16 = w.w << sh;
    }
    t.lo16 &= 0xffffu; t.hi16 &= 0xffffu;
    t.status = TS_SEEDS;
    t.n = 1;
    const uint32_t hc = ((t.lo16 >> 15) & 1u) | (((t.hi16 >> 15) & 1u) << 1);       // bwa.erl:15-24
    if ((o16 >> 15) & 1u) { t.status = TS_CRASH; t.active = false; return; }
    else if (((n16b >> 15) & 1u) || hc == 0) { t.sp = 1; t.ep = ix.Pc - 1; }        // N is searched as A
    else if (hc == 1) { t.sp = ix.Pc; t.ep = ix.Pg - 1; }
    else if (hc == 2) { t.sp = ix.Pg; t.ep = ix.Pt - 1; }
    else { t.sp = ix.Pt; t.ep = ix.Last; }
    t.nlim = e < BWT_MAX_LEN ? e : BWT_MAX_LEN;          // n == MAX_LEN (bwa.erl:31-32) or read exhausted (:44-45)
    const uint32_t sm = (n16b | o16) & 0x7fffu;           // base k of step n sits at plane bit 15-n
    if (sm) {
        const int hb = 31 - __clz((int)sm);
        const int nstop = 15 - hb;
        if (nstop < t.nlim) { t.nlim = nstop; t.check_at_limit = ((o16 >> hb) & 1u) == 0; }   // N: :33-34
    }
    t.active = t.n < t.nlim;
}
// the clause a finished search leaves through (bwa.erl:31-45) and what get_seed_ends makes of it (fmi.erl:89)
__device__ __forceinline__ void task_finish(const LfTask &t_in, bool write, uint32_t *__restrict__ t_sp,
                                            uint32_t *__restrict__ t_ep, uint32_t *__restrict__ t_info,
                                            uint32_t *__restrict__ t_cnt) {
    LfTask t = t_in;
    uint32_t status = t.status;
    if (status == TS_SEEDS) {
        if (t.n >= t.nlim && t.check) t.check = t.check_at_limit;          // tasks that never stepped
        if (t.check && t.ep - t.sp > BWT_MAX_RANGE) status = TS_TOO_MANY;
        else if (t.n < BWT_MIN_LEN) status = TS_NO_SEEDS;                  // fmi.erl:89
    }
    if (write) {
        t_sp[t.task] = t.sp; t_ep[t.task] = t.ep;
        t_info[t.task] = status | ((uint32_t)t.n << 8);
        t_cnt[t.task] = status == TS_SEEDS ? t.ep - t.sp + 1 : 0;
    }
}

// One LF step of a quad on the range [sp, ep] with the base whose code bits are (cl, ch); v1/v2 = this lane's quarter
// of the blocks of rows sp-1 / ep-1.  Returns found; on found sp/ep are the new range.
__device__ __forceinline__ bool lf_step(const DevIndex &ix, int ql, uint32_t dblk, uint4 v1, uint4 v2, uint32_t b1, uint32_t b2,
                                        uint32_t cl, uint32_t ch, uint32_t &sp, uint32_t &ep) {
    const int row0 = 48 * ql;                                    // first row of this lane's quarter
    const int code = (int)(cl | (ch << 1));
    const uint32_t nl = cl - 1u, nh = ch - 1u;        // ~(-bit): all ones when the wanted bit is 0
    const uint32_t n16 = (nl & 0xffffu) | (nh << 16);
    const uint32_t k1 = sp - 1, k2 = ep - 1;
    const int m1 = (int)(k1 - b1 * BWT_BLOCK_ROWS), m2 = (int)(k2 - b2 * BWT_BLOCK_ROWS);
    uint32_t ya0, ya1, yb0, yb1;
    quarter_match(v1, nl, nh, n16, ya0, ya1);
    quarter_match(v2, nl, nh, n16, yb0, yb1);
    const bool mine = ql == code;                     // this lane's count word is the wanted one
    uint32_t o1 = quad_sum(quarter_prefix(ya0, ya1, m1 - row0) + (mine ? v1.x : 0u));   // occ(c,Sp-1)
    uint32_t o2 = quad_sum(quarter_prefix(yb0, yb1, m2 - row0) + (mine ? v2.x : 0u));   // occ(c,Ep-1)
    const int pos = m2 - row0;                        // L[Ep] is row m2 of block b2
    const uint32_t sel = pos < 32 ? yb0 >> (pos & 31) : yb1 >> ((pos - 32) & 31);
    uint32_t bit = quad_sum((pos >= 0 && pos < 48) ? sel & 1u : 0u);                     // L[Ep] == c
    if (b1 == dblk || b2 == dblk) {             // (rare) the block that holds the `$` row: take it out again
        if (code == 0) {
            const uint32_t f = dblk * BWT_BLOCK_ROWS;
            if (b1 == dblk && ix.dollar > f && ix.dollar <= k1) o1 -= 1;
            if (b2 == dblk && ix.dollar > f && ix.dollar <= k2) o2 -= 1;
            if (ix.dollar == ep) bit = 0;
        }
    }
    if (!(o2 > o1)) return false;                                          // se_down: not_found (bwa.erl:38)
    // C0[code] as bitwise selects on the code's plane masks (a per-lane index into the kernel-argument array
    // goes through scratch; chained ?: becomes divergent branches): nl/nh are all ones when that code bit is 0
    const uint32_t c_lo = (nl & 1u) | (~nl & (ix.Pc - 1)), c_hi = (nl & (ix.Pg - 1)) | (~nl & (ix.Pt - 1));
    const uint32_t c0 = (nh & c_lo) | (~nh & c_hi);
    sp = c0 + o1;
    ep = c0 + o2 + bit - 1;                         // C0 + occ(c,Ep) - 1
    return true;
}

template <int U>
__global__ __launch_bounds__(256) void k_search(DevIndex ix, const uint32_t *__restrict__ planes,
                                                 const uint64_t *__restrict__ off, uint64_t n_tasks, uint32_t maxT,
                                                 const uint32_t *__restrict__ t_def, uint32_t *__restrict__ t_sp,
                                                 uint32_t *__restrict__ t_ep, uint32_t *__restrict__ t_info,
                                                 uint32_t *__restrict__ t_cnt, Counters *ctr) {
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t dblk = (ix.dollar - 1) / BWT_BLOCK_ROWS;      // the one block whose planes hold the `$` row
    const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    uint32_t steps = 0, live = 0;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; base < n_tasks; base += (uint64_t)U * nquads) {
        LfTask T[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            LfTask &t = T[u];
            t.task = base + (uint64_t)u * nquads;
            t.sp = 0; t.ep = 0; t.lo16 = 0; t.hi16 = 0; t.status = TS_NONE; t.n = 0; t.nlim = 0;
            t.check_at_limit = true; t.check = true; t.active = false;
            if (t.task >= n_tasks) continue;
            const int e = (int)(t_def[t.task] & 0xffffu);
            if (e <= 0) continue;
            live++;
            const uint32_t r = (uint32_t)(t.task / maxT);
            task_init(t, ix, planes, off[r] + (uint64_t)e, e);
        }
        for (;;) {
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; u++) any |= T[u].active;
            if (!any) break;
            uint4 v1[U], v2[U];
            uint32_t b1[U], b2[U];
#pragma unroll
            for (int u = 0; u < U; u++) {                        // issue every task's two block loads first
                if (!T[u].active) continue;
                b1[u] = (T[u].sp - 1) / BWT_BLOCK_ROWS; b2[u] = (T[u].ep - 1) / BWT_BLOCK_ROWS;
                // (non-temporal loads for the deep, touched-once steps measured slower: 38.6-42 ms vs 36.2 ms)
                v1[u] = ix.blocks[(uint64_t)b1[u] * 4 + ql];
                v2[u] = ix.blocks[(uint64_t)b2[u] * 4 + ql];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                LfTask &t = T[u];
                if (!t.active) continue;
                const uint32_t bi = 15u - (uint32_t)t.n;          // plane bit of the base this step consumes
                steps++;
                if (!lf_step(ix, ql, dblk, v1[u], v2[u], b1[u], b2[u], (t.lo16 >> bi) & 1u, (t.hi16 >> bi) & 1u, t.sp, t.ep)) {
                    t.check = false; t.active = false; continue;
                }
                t.n++;
                if (t.n >= t.nlim) { t.check = t.check_at_limit; t.active = false; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (T[u].task < n_tasks) task_finish(T[u], ql == 0, t_sp, t_ep, t_info, t_cnt);
    }
    // one atomic pair per wave (the grid is capped, so a few thousand per launch) for the work counters
    unsigned long long s = wave_sum_u64(ql == 0 ? steps : 0), l = wave_sum_u64(ql == 0 ? live : 0);
    if ((threadIdx.x & 63) == 0 && (s | l)) {
        atomicAdd(&ctr->lf_steps, s);
        atomicAdd(&ctr->n_tasks, l);
    }
}

// ------------------------------------------------------------------------------------------------
// Suffix-interval table ("ktab"): for every m in 8..K and every m-mer, what the first m-1 LF steps of
// bwa:find_seeds/5 arrive at -- the same closed-form steps, taken once per k-mer when the index is opened
// instead of once per task.  Level m, key = lo_m | hi_m << m where lo_m / hi_m are the low / high code bits
// of the m bases in search order, first searched base in bit m-1 (= the top m bits of a task's plane windows).
// Entry (u64): Sp | (Ep - Sp) << 32 (28 bits; 0xfffffff = does not fit, search it step by step) | n << 60 with
// n = bases matched: n == m: the whole m-mer occurs; n < m: step n+1 came back not_found from [Sp, Ep] (bwa.erl:38).
// Levels below 8 are never asked for: skip_Ns leaves >= 7 N-free bases left of a sub-read's last base and a
// sub-read has >= 8 bases, so a shorter search only happens in front of a base outside ACGTN.
#define KT_MIN 8
#define KT_OVF 0xfffffffu
__host__ __device__ __forceinline__ uint64_t kt_level_off(int m) { return ((1ull << (2 * m)) - (1ull << (2 * KT_MIN))) / 3ull; }
__host__ __device__ __forceinline__ uint64_t kt_entries(int K) { return kt_level_off(K + 1); }
__device__ __forceinline__ uint64_t kt_pack(uint32_t sp, uint32_t ep, int n) {
    const uint32_t w = ep - sp;
    return (uint64_t)sp | ((uint64_t)(w >= KT_OVF ? KT_OVF : w) << 32) | ((uint64_t)n << 60);
}

// quad per key of level m.  m == KT_MIN (or a parent that did not fit): all m-1 steps; else one step from the parent.
// wide != null: the level is written as 64-byte entries {Sp, width | n << 28, SA[Sp], SA[Sp+1], ... 14 values} instead
// (lane q of the quad writes 16-byte quarter q): a task that ends at this level gets its seeds with the same line.
#define KT_INLINE 14
__global__ __launch_bounds__(256) void kt_build(DevIndex ix, uint64_t *__restrict__ tab, int m, uint4 *__restrict__ wide) {
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t dblk = (ix.dollar - 1) / BWT_BLOCK_ROWS;
    const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    const uint64_t nkeys = 1ull << (2 * m);
    const uint32_t mask = (1u << m) - 1u;
    for (uint64_t key = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; key < nkeys; key += nquads) {
        const uint32_t lo = (uint32_t)key & mask, hi = (uint32_t)(key >> m) & mask;
        uint32_t sp = 0, ep = 0;
        int n = 0, todo_from;                 // steps todo_from .. m-1 (step j consumes the base at bit m-1-j) remain
        bool dead = false;
        uint64_t par = 0;
        if (m > KT_MIN) par = tab[kt_level_off(m - 1) + ((uint64_t)(lo >> 1) | ((uint64_t)(hi >> 1) << (m - 1)))];
        if (m > KT_MIN && ((par >> 32) & KT_OVF) != KT_OVF) {
            sp = (uint32_t)par; ep = sp + (uint32_t)((par >> 32) & KT_OVF); n = (int)(par >> 60);
            dead = n < m - 1;
            todo_from = m - 1;
        } else {
            const uint32_t hc = ((lo >> (m - 1)) & 1u) | (((hi >> (m - 1)) & 1u) << 1);
            if (hc == 0) { sp = 1; ep = ix.Pc - 1; }
            else if (hc == 1) { sp = ix.Pc; ep = ix.Pg - 1; }
            else if (hc == 2) { sp = ix.Pg; ep = ix.Pt - 1; }
            else { sp = ix.Pt; ep = ix.Last; }
            n = 1;
            todo_from = 1;
        }
        for (int j = todo_from; j < m && !dead; j++) {
            const uint32_t b1 = (sp - 1) / BWT_BLOCK_ROWS, b2 = (ep - 1) / BWT_BLOCK_ROWS;
            const uint4 v1 = ix.blocks[(uint64_t)b1 * 4 + ql], v2 = ix.blocks[(uint64_t)b2 * 4 + ql];
            const int bi = m - 1 - j;
            if (lf_step(ix, ql, dblk, v1, v2, b1, b2, (lo >> bi) & 1u, (hi >> bi) & 1u, sp, ep)) n++;
            else dead = true;
        }
        const uint64_t ent = kt_pack(sp, ep, n);
        if (!wide) { if (ql == 0) tab[kt_level_off(m) + key] = ent; }
        else {
            const uint32_t wd = (uint32_t)(ent >> 32) & KT_OVF;
            const uint32_t cnt = wd == KT_OVF ? 0u : wd + 1u;
            uint32_t w[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int wi = 4 * ql + c;
                uint32_t v = 0;
                if (wi == 0) v = (uint32_t)ent;
                else if (wi == 1) v = (uint32_t)(ent >> 32);
                else if ((uint32_t)(wi - 2) < cnt) v = ix.sa[sp - 1 + (uint32_t)(wi - 2)];
                w[c] = v;
            }
            wide[key * 4 + (uint64_t)ql] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// kt_widen: level m (below the top) once more as 64-byte entries {Sp, width | n << 28, 14 SA values}, from its 8-byte entries
// (quad per key, lane q writes quarter q).  kt_count_seeded: how many keys of level m give a task that ENDS there a seed list
// (the whole m-mer occurs at most MAX_RANGE + 1 times, or the search stopped short of it with >= 8 bases matched): the levels
// where that is common are the ones worth widening (bwt_index_set_ktable_ex).
__global__ __launch_bounds__(256) void kt_widen(DevIndex ix, const uint64_t *__restrict__ tab, int m, uint4 *__restrict__ wide) {
    const int ql = (int)(threadIdx.x & 3);
    const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2, nkeys = 1ull << (2 * m);
    for (uint64_t key = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; key < nkeys; key += nquads) {
        const uint64_t ent = tab[kt_level_off(m) + key];
        const uint32_t sp = (uint32_t)ent, wd = (uint32_t)(ent >> 32) & KT_OVF;
        const uint32_t cnt = wd == KT_OVF ? 0u : wd + 1u;
        uint32_t w[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int wi = 4 * ql + c;
            uint32_t v = 0;
            if (wi == 0) v = (uint32_t)ent;
            else if (wi == 1) v = (uint32_t)(ent >> 32);
            else if ((uint32_t)(wi - 2) < cnt) v = ix.sa[sp - 1 + (uint32_t)(wi - 2)];
            w[c] = v;
        }
        wide[key * 4 + (uint64_t)ql] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
__global__ __launch_bounds__(256) void kt_count_seeded(const uint64_t *__restrict__ tab, int m, unsigned long long *__restrict__ out) {
    const uint64_t nkeys = 1ull << (2 * m), base = kt_level_off(m);
    unsigned long long mine = 0;
    for (uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; key < nkeys; key += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t ent = tab[base + key];
        const uint32_t wd = (uint32_t)(ent >> 32) & KT_OVF, nn = (uint32_t)(ent >> 60);
        mine += (wd != KT_OVF && nn >= BWT_MIN_LEN && ((int)nn < m || wd <= BWT_MAX_RANGE)) ? 1u : 0u;
    }
    mine = wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out + m, mine);
}

// k_kt_summary: level m of the 8-byte table, one bit per 64 entries (DevIndex::ktsum): set unless all 64 entries are plain
// "too many seeds" entries (n = m steps done, MAX_RANGE < width, not the overflow mark).  A wave per 64 entries, grid-stride;
// nset[m] counts the set bits (how sparse the level is decides whether checking the bit first pays).
__global__ __launch_bounds__(256) void k_kt_summary(const uint64_t *__restrict__ tab, int m, uint32_t *__restrict__ sum,
                                                     unsigned int *__restrict__ nset) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t groups = (1ull << (2 * m)) >> 6, base = kt_level_off(m);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t mine = 0;
    for (uint64_t g = wave; g < groups; g += nwaves) {
        const uint64_t ent = tab[base + g * 64 + lane];
        const uint32_t wd = (uint32_t)(ent >> 32) & KT_OVF, nn = (uint32_t)(ent >> 60);
        const bool plain = (int)nn >= m && wd > BWT_MAX_RANGE && wd != KT_OVF;
        if (__ballot(!plain)) {
            mine++;
            const uint64_t bit = (base >> 6) + g;
            if (lane == 0) atomicOr(&sum[bit >> 5], 1u << (bit & 31));
        }
    }
    if (lane == 0 && mine) atomicAdd(&nset[m], mine);
}

// What a finished read hands to the cluster stage: P = seeds in all, Qty = sub-reads that returned a seed list
// (sga.erl:32); reads with Qty < 2 cluster to [] (sga.erl:42-43) and get P = 0; reads with more than 64 seeds go to
// the queues of the wider cluster kernels.  Reached by whole waves (wave_append); `active` lanes carry a read.
#define RQ_CRASH 0x80000000u
#define RQ_PENDING 0x40000000u
struct ReadQ { WaveQ big, m4, m2; };          // staging for the three queues of reads with more than 64 seeds
#define READQ_LDS_WORDS (3 * 128)
__device__ __forceinline__ void readq_init(ReadQ &q, uint32_t *lds) {
    q.big.buf = lds; q.m4.buf = lds + 128; q.m2.buf = lds + 256; q.big.have = 0; q.m4.have = 0; q.m2.have = 0;
}
__device__ __forceinline__ void readq_drain(ReadQ &q, uint32_t *__restrict__ bigq, uint32_t *__restrict__ mid2q,
                                            uint32_t *__restrict__ mid4q, Counters *ctr) {
    waveq_drain(q.big, bigq, &ctr->n_big); waveq_drain(q.m4, mid4q, &ctr->n_mid4); waveq_drain(q.m2, mid2q, &ctr->n_mid2);
}
__device__ __forceinline__ void read_finish(bool active, uint32_t r, unsigned long long P, uint32_t qty, uint32_t crash,
                                            uint32_t maxT, uint32_t *__restrict__ r_P, uint32_t *__restrict__ r_qty,
                                            uint32_t *__restrict__ bigq, uint32_t *__restrict__ mid2q,
                                            uint32_t *__restrict__ mid4q, Counters *ctr, unsigned long long &Psum, uint32_t &nsmall,
                                            ReadQ &rq) {
    if (qty < 2 || crash) P = 0;
    if (active && P > 0x7fffffffull) { atomicOr(&ctr->err_flags, 1ull); P = 0; }
    if (active) { r_P[r] = (uint32_t)P; r_qty[r] = qty | (crash ? RQ_CRASH : 0u); }
    const bool big = active && (P > 256 || (P > 0 && maxT > 64)), m4 = active && !big && P > 128, m2 = active && !big && !m4 && P > 64;
    waveq_push(rq.big, big, r, bigq, &ctr->n_big);
    waveq_push(rq.m4, m4, r, mid4q, &ctr->n_mid4);
    waveq_push(rq.m2, m2, r, mid2q, &ctr->n_mid2);
    if (active) { Psum += P; nsmall += (P > 0 && P <= 64 && maxT <= 64) ? 1u : 0u; }   // nsmall: reads k_cluster_small has to take
}

#define RQ_DONE 0x20000000u
#define CL_STAGE 4
#define CL_ROW 128                                            // most seeds of a read the seed kernel clusters itself
// get_similar on one sorted lane-per-seed register (P <= 64) and the read's clusters into its staging row (descending
// position order); more than CL_STAGE clusters: the read goes back to k_cluster_small (returns true).  Whole waves.
__device__ __forceinline__ bool cl_emit64(uint32_t p, uint32_t Pj, uint32_t qj, uint32_t rj, uint32_t *__restrict__ r_cl,
                                          uint32_t *__restrict__ r_K, uint32_t *__restrict__ r_P, uint32_t *__restrict__ r_qty,
                                          uint32_t &nsmall, uint32_t &dmax) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nthr = 2 + qj / 7;                     // sga.erl:46
    const uint32_t prev = __shfl_up(p, 1, 64), next = __shfl_down(p, 1, 64);
    const bool vl = lane < Pj;
    const bool start = vl && (lane == 0 || p - prev > BWT_TOLERANCE);
    const bool end = vl && (lane == Pj - 1 || next - p > BWT_TOLERANCE);
    const unsigned long long smask = __ballot(start);
    const unsigned long long below = smask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int s_i = 63 - __clzll(below | 1ull);          // lane of this run's first element
    const uint32_t pfirst = __shfl(p, s_i, 64);
    const bool emit = end && (lane - (uint32_t)s_i + 1 >= nthr);
    const unsigned long long emask = __ballot(emit);
    const uint32_t Kc = (uint32_t)__popcll(emask);
    if (Kc <= CL_STAGE) {
        if (emit) {
            const uint32_t kk = __popcll(lane == 63 ? 0ull : (emask >> (lane + 1)));   // emitters above me
            r_cl[(uint64_t)rj * (2 * CL_STAGE) + 2 * kk] = p;
            r_cl[(uint64_t)rj * (2 * CL_STAGE) + 2 * kk + 1] = p - pfirst;
            dmax = p - pfirst > dmax ? p - pfirst : dmax;
        }
        if (lane == 0) r_K[rj] = Kc;
    } else if (lane == 0) {                               // more clusters than the staging row holds: k_cluster_small redoes it
        r_P[rj] = Pj; r_qty[rj] = qj; nsmall++;
    }
    return Kc > CL_STAGE;
}

// k_seed_tasks: the seed stage with the suffix-interval table.  A wave takes 64 / maxT reads at a time, lane = one
// task slot of one read (so the five task-table columns are written as contiguous runs):
//   * the slot's sub-read (sga:sga/9 + skip_Ns, as k_subreads): a read without N has its k-th sub-read end at
//     len - 8k; only reads that hold an N walk the skip_Ns chain;
//   * one table read answers the task; what the table cannot finish -- sub-reads longer than K (they resume at n = K),
//     searches that stop before 8 bases in front of a base outside ACGTN, entries whose range did not fit -- becomes a
//     bit of r_pend[r] for k_search_pending;
//   * ballots and one wave prefix sum total each read for the cluster stage (as k_read_counts) when nothing is pending;
//   * FUSE (the table's top level carries SA values, 8 <= maxT): complete reads that hold at most 64 seeds are clustered
//     on the spot (seeds of tasks that ended below the top level, or more than 14 of them, are read from the SA) -- positions through a wave-private LDS row per read, then
//     sort and get_similar exactly as k_cluster_small -- into r_cl[r] (up to CL_STAGE {S, D} pairs) and marked
//     RQ_DONE: no SA gather, no second pass over the task table.  Everything else takes the cluster kernels.
//     lean: the task-table rows of such reads are not written at all (nothing reads them; bwt_batch_fetch_tasks refuses).
// RAW: one task per "read", already in t_def (bwt_find_seeds_batch).  1 <= maxT <= 64.
template <bool RAW, bool FUSE>
__global__ __launch_bounds__(256) void k_seed_tasks(DevIndex ix, const uint32_t *__restrict__ planes,
                                                     const uint64_t *__restrict__ off, uint32_t n, uint32_t maxT,
                                                     uint32_t *__restrict__ t_def, uint32_t *__restrict__ t_sp,
                                                     uint32_t *__restrict__ t_ep, uint32_t *__restrict__ t_info,
                                                     uint32_t *__restrict__ t_cnt, unsigned long long *__restrict__ r_pend,
                                                     uint32_t *__restrict__ r_P, uint32_t *__restrict__ r_qty,
                                                     uint32_t *__restrict__ bigq, uint32_t *__restrict__ mid2q,
                                                     uint32_t *__restrict__ mid4q, uint32_t *__restrict__ r_K,
                                                     uint32_t *__restrict__ r_cl, Counters *ctr, int lean, uint32_t ulen,
                                                     const uint32_t *__restrict__ glist, const unsigned int *__restrict__ gcount) {
    __shared__ uint32_t lds_[FUSE ? 4 * 8 * CL_ROW : 1];
    __shared__ uint32_t rq_[4 * READQ_LDS_WORDS];
    ReadQ rq;
    readq_init(rq, rq_ + (threadIdx.x >> 6) * READQ_LDS_WORDS);
    const int K = ix.kt_k;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *lp = lds_ + (FUSE ? (threadIdx.x >> 6) * 8 * CL_ROW : 0);  // [read of the pass][seed]
    const uint32_t rw = 64u / maxT;                               // reads per wave pass (FUSE: <= 8)
    const uint32_t lread = lane / maxT, slot = lane - lread * maxT;
    const uint32_t seg0 = lread * maxT;                           // first lane of my read
    const unsigned long long segmask = maxT == 64 ? ~0ull : ((1ull << maxT) - 1ull);
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    // glist != null: only the READS k_seed_fast handed back (it wrote nothing for them), rw of them per wave pass
    const uint32_t nlisted = glist ? *gcount : 0u;
    const uint32_t groups = glist ? (nlisted + rw - 1) / rw : (n + rw - 1) / rw;
    uint32_t live = 0, looked = 0, nsmall = 0, dmax = 0, had_pend = 0;
    unsigned long long Psum = 0;
    for (uint32_t gi = wave; gi < groups; gi += nwaves) {
        const uint32_t li = gi * rw + lread;
        const uint32_t r = !glist ? li : (lread < rw && li < nlisted) ? glist[li] : 0xffffffffu;
        const bool valid = lread < rw && r < n;
        auto rid = [&](uint32_t j) { return (uint32_t)__shfl((int)r, (int)(j * maxT), 64); };      // the pass's j-th read (j wave-uniform)
        LfTask tk;
        tk.task = (uint64_t)r * maxT + slot;
        tk.sp = 0; tk.ep = 0; tk.lo16 = 0; tk.hi16 = 0; tk.status = TS_NONE; tk.n = 0; tk.nlim = 0;
        tk.check_at_limit = true; tk.check = true; tk.active = false;
        bool pending = false, inl = false;                        // inl: my seeds came with the table entry
        uint32_t status = TS_NONE, c = 0, nadd = 0, shsk = 0;
        int m = 0;                                                // table level asked (0: none)
        uint64_t key = 0;
        // does the read hold an N at all?  lane k of the read looks at word k of its is-N plane (len / 32 < maxT words)
        uint64_t P0 = 0;
        int len = 0;
        bool nword = false;
        if (valid) {
            if (ulen) P0 = (uint64_t)r * ulen;                    // reads of one length: no offset load in front of the plane loads
            else P0 = off[r];
            if (!RAW) {
                len = ulen ? (int)ulen : (int)(off[r + 1] - P0);
                const int i = 32 * (int)slot;
                if (i < len) {
                    uint32_t w = plane_bits(planes, RP_N, P0 + (uint64_t)i);
                    if (len - i < 32) w &= (1u << (len - i)) - 1u;
                    nword = w != 0;
                }
            }
        }
        const bool anyN = ((__ballot(nword) >> seg0) & segmask) != 0;
        if (valid) {
            int e = 0;
            if (RAW) e = (int)(t_def[tk.task] & 0xffffu);
            else {
                if (!anyN) {                                      // sub-read k ends at len - 8k, nothing skipped
                    e = len - (int)(BWT_MIN_LEN * slot);
                    shsk = BWT_MIN_LEN * slot;
                    if (e < BWT_MIN_LEN) e = 0;
                } else {                                          // the chain of k_subreads up to this slot
                    int l2 = len;
                    uint32_t shift = 0;
                    for (uint32_t t = 0;; t++) {
                        e = 0;
                        if (l2 < BWT_MIN_LEN) break;               // sga.erl:69
                        int ee = l2;
                        bool ok = true;
                        for (;;) {
                            if (ee < BWT_MIN_LEN) { ok = false; break; }     // list exhausted before J reached 8 (sga.erl:76)
                            const uint32_t w7 = plane_bits(planes, RP_N, P0 + (uint64_t)(ee - 8)) & 0x7fu;
                            if (!w7) break;
                            ee = ee - 7 + (31 - __clz((int)w7));   // the sub-read now ends in that N
                        }
                        if (!ok) break;
                        const uint32_t skipped = (uint32_t)(l2 - ee);
                        if (t == slot) { e = ee; shsk = shift + skipped; break; }
                        l2 = ee - BWT_MIN_LEN;
                        shift += BWT_MIN_LEN + skipped;
                    }
                }
                t_def[tk.task] = e > 0 ? ((uint32_t)e | (shsk << 16)) : 0u;
            }
            if (e > 0) {
                live++;
                task_init(tk, ix, planes, P0 + (uint64_t)e, e);
                pending = tk.active;
                if (tk.active && tk.nlim >= KT_MIN) {
                    m = tk.nlim < K ? tk.nlim : K;
                    const uint32_t mask = (1u << m) - 1u;
                    key = (uint64_t)((tk.lo16 >> (16 - m)) & mask) | ((uint64_t)((tk.hi16 >> (16 - m)) & mask) << m);
                    looked++;
                }
            }
        }
        // the table read.  Wide top level under FUSE: the four lanes of a quad fetch the quad's four 64-byte entries one
        // after the other, lane q taking quarter q -- one L2 request per entry instead of four (the L2 request rate, not
        // HBM, is what a lane-per-entry fetch of all four quarters runs into); Q[t] = my quarter of the entry of lane t.
        const uint32_t woff = m > 0 ? ktw_level_off(ix, m) : KTW_NONE;     // the level asked is kept as 64-byte entries (the top one, or a widened lower one)
        const bool askw = woff != KTW_NONE;
        const uint64_t wkey = key + woff;
        uint64_t ent = 0;
        uint4 Q[4];
        if (FUSE) {
            const int ql = (int)(lane & 3);
            const uint32_t klo = (uint32_t)wkey, khi = (uint32_t)(wkey >> 32), aw = askw ? 1u : 0u;
#define KT_QLOAD(T)                                                                                                    \
            {                                                                                                          \
                const uint64_t kt = (uint64_t)quad_bcast<T>(klo) | ((uint64_t)quad_bcast<T>(khi) << 32);                \
                Q[T] = quad_bcast<T>(aw) ? ix.ktw[kt * 4 + (uint64_t)ql] : make_uint4(0, 0, 0, 0);                      \
            }
            KT_QLOAD(0) KT_QLOAD(1) KT_QLOAD(2) KT_QLOAD(3)
#undef KT_QLOAD
            // {Sp, width | n} of my own entry: quarter 0 lives in lane 0 of the quad, register [my lane]
            const uint32_t x0 = quad_bcast<0>(Q[0].x), x1 = quad_bcast<0>(Q[1].x), x2 = quad_bcast<0>(Q[2].x), x3 = quad_bcast<0>(Q[3].x);
            const uint32_t y0 = quad_bcast<0>(Q[0].y), y1 = quad_bcast<0>(Q[1].y), y2 = quad_bcast<0>(Q[2].y), y3 = quad_bcast<0>(Q[3].y);
            const uint32_t ex = ql == 0 ? x0 : ql == 1 ? x1 : ql == 2 ? x2 : x3, ey = ql == 0 ? y0 : ql == 1 ? y1 : ql == 2 ? y2 : y3;
            if (askw) { ent = (uint64_t)ex | ((uint64_t)ey << 32); inl = true; }
        } else if (askw) {
            const uint4 q0 = ix.ktw[wkey * 4];
            ent = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
        }
        if (m > 0 && !askw) ent = ix.ktab[kt_level_off(m) + key];
        if (valid) {
            if (m > 0) {
                const uint32_t wd = (uint32_t)(ent >> 32) & KT_OVF;
                if (wd != KT_OVF) {
                    tk.sp = (uint32_t)ent; tk.ep = tk.sp + wd; tk.n = (int)(ent >> 60);
                    if (tk.n < m) { tk.check = false; pending = false; }                          // not_found inside the table
                    else if (m == tk.nlim) { tk.check = tk.check_at_limit; pending = false; }
                }
            }
            status = tk.status;
            if (pending) status = TS_NONE;
            else if (status == TS_SEEDS) {
                if (tk.n >= tk.nlim && tk.check) tk.check = tk.check_at_limit;          // tasks that never stepped
                if (tk.check && tk.ep - tk.sp > BWT_MAX_RANGE) status = TS_TOO_MANY;
                else if (tk.n < BWT_MIN_LEN) status = TS_NO_SEEDS;                      // fmi.erl:89
            }
            c = (!pending && status == TS_SEEDS) ? tk.ep - tk.sp + 1 : 0;
            nadd = (uint32_t)tk.n + shsk;                                               // N + Shift + Skip (sga.erl:32,37-39)
        }
        // per-read totals: the read's lanes are [seg0, seg0 + maxT)
        const unsigned long long bp = __ballot(pending), bs = __ballot(valid && !pending && status == TS_SEEDS),
                                 bc = __ballot(valid && tk.status == TS_CRASH);
        unsigned long long incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long up = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
        const unsigned long long hi = __shfl(incl, (int)((seg0 + maxT - 1) & 63u), 64), lo = __shfl(incl, (int)((seg0 - 1) & 63u), 64);
        const unsigned long long before = seg0 ? lo : 0ull;
        unsigned long long P = hi - before;
        const unsigned long long pend = (bp >> seg0) & segmask;
        const uint32_t qty = (uint32_t)__popcll((bs >> seg0) & segmask), crash = ((bc >> seg0) & segmask) ? 1u : 0u;
        const bool leader = valid && slot == 0;
        if (leader && pend != 0) had_pend = 1u;                  // (never with a full-depth table and clean reads)
        if (leader) {
            r_pend[r] = pend;
            if (pend) {                                           // k_search_pending adds its share and finishes the read
                r_P[r] = (uint32_t)(P > 0x7fffffffull ? 0x7fffffffull : P);
                r_qty[r] = qty | (crash ? RQ_CRASH : 0u) | RQ_PENDING;
            }
        }
        bool fuse_read = false;
        if (FUSE) {
            // a read is clustered here when it is complete, clusters at all (Qty >= 2, no crash) and holds <= 128 seeds
            fuse_read = valid && !pend && qty >= 2 && !crash && P > 0 && P <= CL_ROW;
            const bool inline_ok = inl && c <= KT_INLINE;                 // my seeds came with the table entry
            // tasks that ended below the top level, or hold more than 14 seeds: their seeds are read from the SA, the whole
            // wave on one task's range at a time (lane i takes SA[Sp + i], SA[Sp + 64 + i]), four tasks' loads in flight together
            {
                unsigned long long need = __ballot(fuse_read && c && !inline_ok);
                const uint32_t mydst = lread * CL_ROW + (uint32_t)(incl - c - before);
                while (need) {
                    uint32_t gsp[4], gc[4], gadd[4], gdst[4], gv[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        gc[t] = 0; gsp[t] = 1; gadd[t] = 0; gdst[t] = 0;
                        if (need) {
                            const int src = __ffsll((long long)need) - 1;
                            need &= need - 1ull;
                            gsp[t] = (uint32_t)__shfl((int)tk.sp, src, 64); gc[t] = (uint32_t)__shfl((int)c, src, 64);
                            gadd[t] = (uint32_t)__shfl((int)nadd, src, 64); gdst[t] = (uint32_t)__shfl((int)mydst, src, 64);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) gv[t] = lane < gc[t] ? ix.sa[gsp[t] - 1 + lane] : 0u;
#pragma unroll
                    for (int t = 0; t < 4; t++) if (lane < gc[t]) lp[gdst[t] + lane] = gv[t] + gadd[t];
#pragma unroll
                    for (int t = 0; t < 4; t++)                                   // (a range of 65..128 rows)
                        if (lane + 64 < gc[t]) lp[gdst[t] + 64 + lane] = ix.sa[gsp[t] - 1 + 64 + lane] + gadd[t];
                }
            }
            // positions into the read's LDS row: lane q of the quad holds quarter q of the quad's four entries, i.e. SA values
            // [4q-2, 4q+2) of each (quarter 0: values 0, 1 in its .z, .w)
            {
                const int ql = (int)(lane & 3);
                const uint32_t cw = fuse_read && inline_ok ? c : 0u;          // seeds of mine the quad writes
                const uint32_t dsto = cw ? lread * CL_ROW + (uint32_t)(incl - c - before) : 0u;
#define KT_QWRITE(T)                                                                                                   \
                {                                                                                                      \
                    const uint32_t ct = quad_bcast<T>(cw), dt = quad_bcast<T>(dsto), at = quad_bcast<T>(nadd);         \
                    const uint32_t i0 = ql ? 4u * (uint32_t)ql - 2u : 0u;                                              \
                    const uint32_t v0 = ql ? Q[T].x : Q[T].z, v1 = ql ? Q[T].y : Q[T].w;                               \
                    if (i0 < ct) lp[dt + i0] = v0 + at;                                                                \
                    if (i0 + 1 < ct) lp[dt + i0 + 1] = v1 + at;                                                        \
                    if (ql && i0 + 2 < ct) lp[dt + i0 + 2] = Q[T].z + at;                                              \
                    if (ql && i0 + 3 < ct) lp[dt + i0 + 3] = Q[T].w + at;                                              \
                }
                KT_QWRITE(0) KT_QWRITE(1) KT_QWRITE(2) KT_QWRITE(3)
#undef KT_QWRITE
            }
        }
        // the task-table row (lean: only where a later kernel reads it -- the reads not clustered here)
        if (valid && !(lean && fuse_read)) {
            t_sp[tk.task] = tk.sp; t_ep[tk.task] = tk.ep; t_info[tk.task] = status | ((uint32_t)tk.n << 8); t_cnt[tk.task] = c;
        }
        read_finish(leader && !pend && !fuse_read, r, P, qty, crash, maxT, r_P, r_qty, bigq, mid2q, mid4q, ctr, Psum, nsmall, rq);
        if (FUSE) {
            if (leader && fuse_read) { r_P[r] = 0; r_qty[r] = qty | RQ_DONE; Psum += P; }   // r_P: seeds still to be clustered
            __builtin_amdgcn_wave_barrier();
            const unsigned long long bf = __ballot(leader && fuse_read);                   // bit seg0 of every read to cluster
            // reads of up to 64 seeds are sorted two at a time (two independent chains of stages), the rest one by one
            uint32_t held = 0xffffffffu;                              // a read of <= 64 seeds waiting for its partner
            for (uint32_t j = 0; j <= rw; j++) {
                const bool last = j == rw;
                if (!last && !((bf >> (j * maxT)) & 1ull)) continue;                          // wave-uniform
                uint32_t Pj = 0, qj = 0;
                if (!last) { Pj = (uint32_t)__shfl((int)(uint32_t)P, (int)(j * maxT), 64); qj = (uint32_t)__shfl((int)qty, (int)(j * maxT), 64); }
                if (last || Pj <= 64) {
                    if (!last && held == 0xffffffffu) { held = j; continue; }
                    if (held == 0xffffffffu) continue;                // (last, nothing waiting)
                    const uint32_t ja = held;
                    const uint32_t Pa = (uint32_t)__shfl((int)(uint32_t)P, (int)(ja * maxT), 64), qa = (uint32_t)__shfl((int)qty, (int)(ja * maxT), 64);
                    uint32_t pa = lane < Pa ? lp[ja * CL_ROW + lane] : 0xffffffffu;
                    // a read that goes back to the cluster kernels needs its task-table rows after all (lean mode left them
                    // out): its lanes still hold them
                    bool fba, fbb = false;
                    if (last) {
                        pa = wave_sort_u32(pa);
                        fba = cl_emit64(pa, Pa, qa, rid(ja), r_cl, r_K, r_P, r_qty, nsmall, dmax);
                    } else {
                        uint32_t pb = lane < Pj ? lp[j * CL_ROW + lane] : 0xffffffffu;
                        wave_sort2_u32(pa, pb);
                        fba = cl_emit64(pa, Pa, qa, rid(ja), r_cl, r_K, r_P, r_qty, nsmall, dmax);
                        fbb = cl_emit64(pb, Pj, qj, rid(j), r_cl, r_K, r_P, r_qty, nsmall, dmax);
                    }
                    if (lean && valid && ((fba && lread == ja) || (fbb && lread == j))) {
                        t_sp[tk.task] = tk.sp; t_ep[tk.task] = tk.ep; t_info[tk.task] = status | ((uint32_t)tk.n << 8); t_cnt[tk.task] = c;
                    }
                    held = 0xffffffffu;
                    continue;
                }
                const uint32_t rj = rid(j);
                const uint32_t nthr = 2 + qj / 7;                     // sga.erl:46
                uint32_t *row = lp + j * CL_ROW;
                {
                    // 65..128 seeds: two per lane (element e = 64 q + lane).  Both halves sorted, then element i meets element
                    // 127 - i (the smaller 64 stay in the low half), then each half -- bitonic now -- is merged.
                    uint32_t p0 = row[lane], p1 = lane + 64 < Pj ? row[64 + lane] : 0xffffffffu;
                    wave_sort2_u32(p0, p1);
                    const uint32_t rv = (uint32_t)__shfl((int)p1, 63 - (int)lane, 64);
                    p1 = p0 < rv ? rv : p0; p0 = p0 < rv ? p0 : rv;
#if BWT_GENERIC_MIRROR_SORT
                    { uint32_t v[2] = {p0, p1}; bitonic_merge64(v, sort_masks()); p0 = v[0]; p1 = v[1]; }
#else
                    bitonic_merge2<64, 32>(p0, p1);
#endif
                    __builtin_amdgcn_wave_barrier();
                    row[lane] = p0; row[64 + lane] = p1;
                    __builtin_amdgcn_wave_barrier();
                    // run starts and emit flags per half (ascending), the last run start carried across (as k_cluster_wave<2>)
                    unsigned long long em[2];
                    uint32_t s_of[2];
                    uint32_t carry_s = 0;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const uint32_t e = (uint32_t)(q * 64) + lane, pe = q ? p1 : p0;
                        const uint32_t prev = e ? row[e - 1] : 0u, next = e + 1 < 128u ? row[e + 1] : 0xffffffffu;
                        const bool vl = e < Pj;
                        const bool start = vl && (e == 0 || pe - prev > BWT_TOLERANCE);
                        const bool end = vl && (e == Pj - 1 || next - pe > BWT_TOLERANCE);
                        const unsigned long long smask = __ballot(start);
                        const unsigned long long below = smask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                        uint32_t s_e = carry_s;
                        if (below) s_e = (uint32_t)(q * 64 + (63 - __clzll(below)));
                        if (smask) carry_s = (uint32_t)(q * 64 + (63 - __clzll(smask)));
                        s_of[q] = s_e;
                        em[q] = __ballot(end && (e - s_e + 1 >= nthr));
                    }
                    const uint32_t Kc = (uint32_t)(__popcll(em[0]) + __popcll(em[1]));
                    if (Kc <= CL_STAGE) {
                        uint32_t above = 0;                           // output in descending position order
#pragma unroll
                        for (int q = 1; q >= 0; q--) {
                            if ((em[q] >> lane) & 1ull) {
                                const uint32_t pe = q ? p1 : p0;
                                const uint32_t kk = above + (uint32_t)__popcll(lane == 63 ? 0ull : (em[q] >> (lane + 1)));
                                r_cl[(uint64_t)rj * (2 * CL_STAGE) + 2 * kk] = pe;
                                r_cl[(uint64_t)rj * (2 * CL_STAGE) + 2 * kk + 1] = pe - row[s_of[q]];
                                dmax = pe - row[s_of[q]] > dmax ? pe - row[s_of[q]] : dmax;
                            }
                            above += (uint32_t)__popcll(em[q]);
                        }
                        if (lane == 0) r_K[rj] = Kc;
                    } else if (lane == 0) { r_P[rj] = Pj; r_qty[rj] = qj; }          // ... k_cluster_wave<2> redoes it
                    if (Kc > CL_STAGE && lean && valid && lread == j) {                 // (its rows: see above)
                        t_sp[tk.task] = tk.sp; t_ep[tk.task] = tk.ep; t_info[tk.task] = status | ((uint32_t)tk.n << 8); t_cnt[tk.task] = c;
                    }
                    waveq_push(rq.m2, lane == 0 && Kc > CL_STAGE, rj, mid2q, &ctr->n_mid2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    readq_drain(rq, bigq, mid2q, mid4q, ctr);
    if (__ballot(had_pend != 0) && lane == 0) ctr->n_pend = 1u;        // (a plain store: every writer writes the same)
    {
        const unsigned long long sums[4] = {wave_sum_u64(live), wave_sum_u64(looked), wave_sum_u64(Psum), wave_sum_u64(nsmall)};
        unsigned long long *const g64[4] = {&ctr->n_tasks, &ctr->kt_lookups, &ctr->sa_reads, nullptr};
        unsigned int *const g32[4] = {nullptr, nullptr, nullptr, &ctr->n_small};
        const uint32_t mx[1] = {wave_max_u32(dmax)};
        unsigned int *const gmx[1] = {&ctr->max_D};
        block_counters(sums, g64, g32, mx, gmx);
    }
}

// ------------------------------------------------------------------------------------------------
// k_seed_fast: k_seed_tasks<false, true> in lean mode, cut down to the ordinary read -- every read of the batch has the
// same length (ulen), the table is K = 15 with SA values in its top level, and in the group of 64 / maxT reads a wave
// takes per pass
//   * no read holds a base outside ACGT (the windows [e-16, e) of a read's sub-reads cover the whole read),
//   * every table entry fits (no KT_OVF),
//   * every read has at most 128 seeds and at most CL_STAGE clusters.
// A group that fails any of these is handed back untouched -- its index goes to gq and k_seed_tasks runs on that list
// afterwards -- so the results are k_seed_tasks' by construction; this kernel only has to be right on what it keeps.
// What it saves over the general kernel (which sits at the VALU issue limit): the sub-read chain and the N-plane loads
// (e = ulen - 8 slot is a lane constant), the u64 prefix sum (one packed u32 DPP scan gives seeds-before-me, P and Qty),
// the quad broadcasts that fetch a lane's own entry (the entries go through an LDS transpose: four 16-byte writes, then
// every lane reads its own slot), the predicated word-by-word seed stores (each lane copies its own seeds), get_similar
// on lane masks in SGPRs (run starts by one ballot; run lengths, the emit test and the cluster count are scalar ops; only
// the one or two emitted clusters touch the VALU again), no task-table rows, one store of the group's results at the end.
#define SF_ROW 128            // seeds of a read the fast kernel sorts: one register (<= 64) or two
#define SF_SLOT 5
__device__ __forceinline__ uint32_t dpp_scan_incl_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);    // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);    // row_bcast:31 into rows 2, 3
    return v;
}
// the (wave-uniform) value v into lane `idx` of a register
__device__ __forceinline__ uint32_t set_lane(uint32_t old, uint32_t idx, uint32_t v) {
    return (threadIdx.x & 63u) == idx ? v : old;
}
// dynamic LDS of a k_seed_fast block (4 waves)
__host__ __device__ inline uint32_t sf_lds_bytes(uint32_t maxT, bool big) {
    const uint32_t rw = 64u / maxT, nslot = (rw * maxT + 3u) & ~3u;
    return 4u * 4u * (nslot * SF_SLOT * 4 + rw * (big ? SF_ROW : 64u) + 128);
}
// BIG: reads of up to 128 seeds (two registers per read; 106 VGPRs, 4 waves per SIMD); else up to 64 (95 VGPRs, 5 waves -- worth
// 5 % on batches whose reads rarely pass 64 seeds, e.g. 100-bp reads)
// SUM: the batch's sub-read below the top level looks at the summary bit first (DevIndex::ktsum covers its level)
// (the two-register variant is held to the five waves per SIMD its LDS allows: left alone it takes 100 VGPRs)
template <bool BIG, bool SUM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BIG ? 5 : 4, 8)))
void k_seed_fast(DevIndex ix, const uint32_t *__restrict__ planes, uint32_t n, uint32_t maxT,
                                                    uint32_t ulen, unsigned long long *__restrict__ r_pend,
                                                    uint32_t *__restrict__ r_P, uint32_t *__restrict__ r_qty,
                                                    uint32_t *__restrict__ r_K, uint32_t *__restrict__ r_cl,
                                                    uint32_t *__restrict__ gq, Counters *ctr) {
    // LDS per wave (dynamic, sized by the host: sf_lds_bytes): task slots x four 16-byte quarters (+ 16 bytes: slots 80 bytes
    // apart, no bank conflicts) | a row of SF_ROW seeds per read of the pass | 128 words of queue staging
    extern __shared__ uint4 sf_lds_[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t rw = 64u / maxT;
    constexpr uint32_t ROW = BIG ? SF_ROW : 64u;                     // seeds a read's row holds
    const uint32_t nslot = (rw * maxT + 3u) & ~3u, wave_words = nslot * SF_SLOT * 4 + rw * ROW + 128;
    uint32_t *wbase = (uint32_t *)sf_lds_ + wv * wave_words;
    uint4 *stag = (uint4 *)wbase;
    uint32_t *rows = wbase + nslot * SF_SLOT * 4;
    WaveQ q;
    q.buf = rows + rw * ROW; q.have = 0;
    const uint32_t lread = lane / maxT, slot = lane - lread * maxT, seg0 = lread * maxT;
    const uint32_t ql = lane & 3u;
    // lane constants of the slot: sub-read end, step limit = table level, bits of the 16-base window that belong to the read
    const int e = (int)ulen - (int)(BWT_MIN_LEN * slot) >= BWT_MIN_LEN ? (int)ulen - (int)(BWT_MIN_LEN * slot) : 0;
    const int m = e < BWT_MAX_LEN ? e : BWT_MAX_LEN;
    const uint32_t kmask = (1u << m) - 1u, ksh = 16u - (uint32_t)m;
    const uint32_t inread = e >= 16 ? 0xffffu : ((0xffffu << (16 - e)) & 0xffffu);
    const uint32_t woff = m >= KT_MIN ? ktw_level_off(ix, m) : KTW_NONE;    // my level as 64-byte entries: the top one (always, here) or a widened lower one
    const bool wide = woff != KTW_NONE;
    const uint64_t loff = kt_level_off(m > KT_MIN ? m : KT_MIN);
    const bool usesum = SUM && !wide && m >= KT_MIN && ix.ktsum != nullptr && ((ix.ktsum_levels >> m) & 1u);   // check the summary bit before the 8-byte entry
    const uint32_t shsk = BWT_MIN_LEN * slot;
    const int hi_l = (int)(((seg0 + maxT - 1) & 63u) << 2), lo_l = (int)(((seg0 - 1) & 63u) << 2);
    const SortMasks SM = sort_masks();
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t groups = (n + rw - 1) / rw;
    uint32_t live = 0, why = 0, why2 = 0;                          // why: hand-backs by reason (packed counts, flushed when a byte fills)
    unsigned long long Pacc = 0;                                   // seeds clustered here, per leader lane
    uint32_t dmaxl = 0;                                            // widest cluster staged here
    // Software pipeline: when a pass starts on group g, g's table entries are already in flight (their loads were issued a
    // pass earlier, before the sorts of the group before it) and so are the plane windows of the group after g.  The pass
    // moves g's entries into the LDS staging area, issues the next group's table loads into the same registers and the
    // plane loads of the group after that, and only then works on g -- from LDS, with no global load to wait for.
    // (Entries of lanes that ask for none are left undefined on purpose -- a zero default would make the compiler merge it
    // with the loaded value right behind the load, i.e. wait for it; nothing reads them: such lanes have no seeds.  The plane
    // words travel raw, two 16-byte loads per lane, and are aligned to the window when the next pass picks them up.)
    uint4 Q0, Q1, Q2, Q3;
    unsigned long long nent;
    bool abn0_c = false, nplain_c = false;
    auto planes_raw = [&](uint32_t g2, uint4 &pa, uint4 &pb) {     // plane words around the window [e - 16, e) of my sub-read in group g2
        const uint32_t r2 = g2 * rw + lread;
        const uint64_t endp2 = (uint64_t)r2 * ulen + (uint64_t)e;
        const bool has = g2 < groups && lread < rw && r2 < n && e > 0 && endp2 >= 16;
        const uint64_t wi = has ? (endp2 - 16) >> 5 : 0ull;        // no such task: word 0 (cached, never looked at)
        pa = ((const uint4 *)planes)[wi]; pb = ((const uint4 *)planes)[wi + 1];
    };
    auto issue_table = [&](uint32_t g2, const uint4 &pa, const uint4 &pb) {   // keys of group g2 from its plane words, table loads issued
        const uint32_t r2 = g2 * rw + lread;
        const bool act2 = g2 < groups && lread < rw && r2 < n && e > 0;
        const uint64_t endp2 = (uint64_t)r2 * ulen + (uint64_t)e;
        const uint32_t sh = (uint32_t)(endp2 - 16) & 31u;
        const uint32_t wx = __builtin_amdgcn_alignbit(pb.x, pa.x, sh), wy = __builtin_amdgcn_alignbit(pb.y, pa.y, sh);
        const uint32_t wz = __builtin_amdgcn_alignbit(pb.z, pa.z, sh), ww = __builtin_amdgcn_alignbit(pb.w, pa.w, sh);
        const uint32_t key = ((wx >> ksh) & kmask) | (((wy >> ksh) & kmask) << m);
        abn0_c = act2 && (endp2 < 16 || ((wz | ww) & inread) != 0);
        // (the summary word of a lane below the top level: asked for first, so that it can be waited for with the table loads
        // behind it still in flight)
        const uint64_t nidx = loff + key;
        uint32_t sword = 0xffffffffu;
        if (SUM && act2 && usesum) sword = ix.ktsum[nidx >> 11];
        // lane q of a quad takes quarter q of the quad's four entries (one L2 request per entry)
        const uint32_t aw = act2 && wide ? 1u : 0u;
        const uint32_t wk = key + woff;                                 // (all widened levels together hold fewer than 2^32 entries)
        { const uint32_t kt = quad_bcast<0>(wk); if (quad_bcast<0>(aw)) Q0 = ix.ktw[(uint64_t)kt * 4 + ql]; }
        { const uint32_t kt = quad_bcast<1>(wk); if (quad_bcast<1>(aw)) Q1 = ix.ktw[(uint64_t)kt * 4 + ql]; }
        { const uint32_t kt = quad_bcast<2>(wk); if (quad_bcast<2>(aw)) Q2 = ix.ktw[(uint64_t)kt * 4 + ql]; }
        { const uint32_t kt = quad_bcast<3>(wk); if (quad_bcast<3>(aw)) Q3 = ix.ktw[(uint64_t)kt * 4 + ql]; }
        // summary bit clear: the entry is a plain "too many seeds" one -- known without its table line.  (The load stays a bare
        // conditional load: giving nent a value on the other path would merge right behind the load, i.e. wait for it here.)
        if (SUM) {
            nplain_c = act2 && !wide && !((sword >> ((nidx >> 6) & 31u)) & 1u);
            if (act2 && !wide && !nplain_c) nent = ix.ktab[nidx];
        } else if (act2 && !wide) nent = ix.ktab[nidx];
    };
    uint4 wa, wb;
    planes_raw(wave, wa, wb);
    issue_table(wave, wa, wb);
    planes_raw(wave + nwaves, wa, wb);
    for (uint32_t g = wave; g < groups; g += nwaves) {
        if (((why | why2) & 0x80808080u) && lane == 0) {
            atomicAdd(&ctr->why[0], why & 255u); atomicAdd(&ctr->why[1], (why >> 8) & 255u); atomicAdd(&ctr->why[2], (why >> 16) & 255u);
            atomicAdd(&ctr->why[3], why >> 24); atomicAdd(&ctr->why[4], why2 & 0xffffu); atomicAdd(&ctr->why[5], why2 >> 16);
        }
        if ((why | why2) & 0x80808080u) { why = 0; why2 = 0; }
        const uint32_t r = g * rw + lread;
        bool valid = lread < rw && r < n;                              // (cleared for a read that is handed back)
        const bool act = valid && e > 0;
        const bool abn0 = abn0_c;
        live += (uint32_t)__popcll(__ballot(act));
        // transpose through LDS: afterwards slot t of the staging area holds the whole entry of lane t
        const uint32_t qb = (lane & ~3u) * SF_SLOT + ql;
        stag[qb] = Q0; stag[qb + SF_SLOT] = Q1; stag[qb + 2 * SF_SLOT] = Q2; stag[qb + 3 * SF_SLOT] = Q3;
        __builtin_amdgcn_wave_barrier();
        uint4 own = stag[lane * SF_SLOT];
        if (!wide) {
            // (a plain too-many-seeds entry of level m: n = m steps, width MAX_RANGE + 1)
            const unsigned long long ne = SUM && nplain_c ? (((unsigned long long)(uint32_t)m << 60) | ((unsigned long long)(BWT_MAX_RANGE + 1u) << 32) | 1ull) : nent;
            own.x = act ? (uint32_t)ne : 0u; own.y = act ? (uint32_t)(ne >> 32) : 0u;
        }
        if (!act) own.y = 0;
        // the next group's table loads, the plane loads of the group after it
        issue_table(g + nwaves, wa, wb);
        planes_raw(g + 2 * nwaves, wa, wb);
        const uint32_t wd = own.y & KT_OVF, nn = own.y >> 28;
        // bwa.erl:31-45 at the end of the search: the range check applies unless a step came back not_found (n < m)
        const bool seeds0 = act && !((int)nn >= m && wd > BWT_MAX_RANGE) && nn >= BWT_MIN_LEN;
        // A read that is not ordinary -- a base outside ACGT, a table entry that did not fit, a task with more seeds than a row
        // holds (its read is over the limit anyway, and the packed scan below must not overflow) -- is handed back ALONE: its
        // lanes take no further part in the pass (round 2 handed back the whole group of 64 / maxT reads: on a repeat-rich genome
        // three quarters of all groups hold such a read and went through the general kernel for it).
        const bool ovf = act && wd == KT_OVF, over = seeds0 && wd + 1u > ROW;
        // handing a read back: its index into the queue of the general kernel (which writes everything for it), its tasks off this
        // kernel's count, its lanes out of the rest of the pass (`valid` cleared: no flag of its own to keep in a register)
        auto hand_back = [&](bool d) {
            live -= (uint32_t)__popcll(__ballot(act && d));
            __builtin_amdgcn_wave_barrier();
            waveq_push(q, d && slot == 0, r, gq, &ctr->n_gq);
            valid = valid && !d;
        };
        // (the two-register variant leaves the pass when it hands a read back and sends the other reads of the pass -- two at most --
        // along: carrying on costs it registers it does not have, 104 VGPRs = a wave per SIMD less, or spills; its batches are long
        // reads, where a pass holds three reads and hand-backs are rare)
#define SF_HAND_BACK(d)                                                                                       \
        if (BIG) {                                                                                            \
            if (__ballot(d)) {                                                                                \
                live -= (uint32_t)__popcll(__ballot(act && valid));                                           \
                __builtin_amdgcn_wave_barrier();                                                              \
                waveq_push(q, valid && slot == 0, r, gq, &ctr->n_gq);                                         \
                continue;                                                                                     \
            }                                                                                                 \
        } else hand_back(d)
        if (BIG) {
            if (__ballot(abn0 || ovf || over)) {                       // (leaves the pass: see SF_HAND_BACK)
                why += __ballot(abn0) ? 1u : __ballot(ovf) ? 1u << 8 : 0u;
                if (!__ballot(abn0) && !__ballot(ovf)) why2 += 1u;
                live -= (uint32_t)__popcll(__ballot(act));
                __builtin_amdgcn_wave_barrier();
                waveq_push(q, valid && slot == 0, r, gq, &ctr->n_gq);
                continue;
            }
        } else if (__ballot(abn0 || ovf || over)) {
            const unsigned long long b_abn0 = __ballot(abn0), b_ovf = __ballot(ovf), b_over = __ballot(over);
            const unsigned long long segm = (((1ull << maxT) - 1ull) << seg0);
            const bool d0 = (b_abn0 & segm) != 0, d1 = !d0 && (b_ovf & segm) != 0, d2 = !d0 && !d1 && (b_over & segm) != 0;
            why += (uint32_t)__popcll(__ballot(valid && slot == 0 && d0)) + ((uint32_t)__popcll(__ballot(valid && slot == 0 && d1)) << 8);
            why2 += (uint32_t)__popcll(__ballot(valid && slot == 0 && d2));
            hand_back(valid && (d0 || d1 || d2));
        }
        const bool seeds = seeds0 && valid;
        const uint32_t c = seeds ? wd + 1u : 0u;
        const uint32_t nadd = nn + shsk;                               // N + Shift (sga.erl:32,37-39; nothing skipped)
        // one packed scan: low half = seeds, high half = sub-reads that returned a seed list (Qty)
        const uint32_t v = c | (seeds ? 0x10000u : 0u);
        const uint32_t incl = dpp_scan_incl_u32(v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(hi_l, (int)incl);
        // (unconditional: a bpermute under a divergent branch reads 0 from the lanes that are switched off)
        const uint32_t lo_any = (uint32_t)__builtin_amdgcn_ds_bpermute(lo_l, (int)incl);
        const uint32_t lo = seg0 ? lo_any : 0u;
        const uint32_t Pq = hi - lo, P = Pq & 0xffffu, qty = Pq >> 16;
        const uint32_t dst = (incl - v - lo) & 0xffffu;
        bool fuse_read = valid && qty >= 2 && P > 0;                   // sga.erl:42-43
        if (__ballot(fuse_read && P > ROW)) {                          // a read with more seeds than a row holds: handed back (P is the same in all its lanes)
            const bool d = fuse_read && P > ROW;
            why2 += (uint32_t)__popcll(__ballot(d && slot == 0));
            fuse_read = fuse_read && !d;
            SF_HAND_BACK(d);
        }
        // every lane copies its own seeds (+ N + Shift) into its read's row
        {
            uint32_t *row = rows + lread * ROW + dst;
            // seeds that did not come with the entry -- everything of a task that ended below the top level, the seeds past the
            // first 14 of a top-level task -- are read from the SA, the whole wave on one task's range at a time (rare)
            const uint32_t inl = wide ? KT_INLINE : 0u;
            // (up to four ranges per round: their loads are all issued before the first one is waited for)
            unsigned long long more = __ballot(fuse_read && c > inl);
            while (more) {
                uint32_t gv[4], gv2[4], gdst[4], gadd[4], gc[4], gfrom[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    gc[t] = 0; gfrom[t] = 0; gdst[t] = 0; gadd[t] = 0; gv[t] = 0; gv2[t] = 0;
                    if (more) {
                        const int src = __ffsll((long long)more) - 1;
                        more &= more - 1ull;
                        const uint32_t sp_s = (uint32_t)__builtin_amdgcn_readlane((int)own.x, src);
                        gc[t] = (uint32_t)__builtin_amdgcn_readlane((int)c, src);
                        gadd[t] = (uint32_t)__builtin_amdgcn_readlane((int)nadd, src); gfrom[t] = (uint32_t)__builtin_amdgcn_readlane((int)inl, src);
                        gdst[t] = (uint32_t)__builtin_amdgcn_readlane((int)(lread * ROW + dst), src);
                        if (lane >= gfrom[t] && lane < gc[t]) gv[t] = ix.sa[sp_s - 1 + lane];
                        if (BIG && lane + 64 < gc[t]) gv2[t] = ix.sa[sp_s - 1 + 64 + lane];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (lane >= gfrom[t] && lane < gc[t]) rows[gdst[t] + lane] = gv[t] + gadd[t];
                    if (BIG && lane + 64 < gc[t]) rows[gdst[t] + 64 + lane] = gv2[t] + gadd[t];
                }
            }
            const uint32_t cw = fuse_read ? (c < inl ? c : inl) : 0u;
            if (cw > 0) row[0] = own.z + nadd;
            if (cw > 1) row[1] = own.w + nadd;
            if (__ballot(cw > 2)) {
                const uint4 t = stag[lane * SF_SLOT + 1];
                if (cw > 2) row[2] = t.x + nadd;
                if (cw > 3) row[3] = t.y + nadd;
                if (cw > 4) row[4] = t.z + nadd;
                if (cw > 5) row[5] = t.w + nadd;
                if (__ballot(cw > 6)) {
                    const uint4 t2 = stag[lane * SF_SLOT + 2];
                    if (cw > 6) row[6] = t2.x + nadd;
                    if (cw > 7) row[7] = t2.y + nadd;
                    if (cw > 8) row[8] = t2.z + nadd;
                    if (cw > 9) row[9] = t2.w + nadd;
                    if (__ballot(cw > 10)) {
                        const uint4 t3 = stag[lane * SF_SLOT + 3];
                        if (cw > 10) row[10] = t3.x + nadd;
                        if (cw > 11) row[11] = t3.y + nadd;
                        if (cw > 12) row[12] = t3.z + nadd;
                        if (cw > 13) row[13] = t3.w + nadd;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // Sort: reads of up to 64 seeds up to three side by side in one register each; a read of 65..128 seeds in two registers
        // (element e = 64 q + lane: both halves sorted, element i meets element 127 - i, each half -- bitonic now -- merged).
        // Each sorted read goes back into its row, and its run starts (a ballot per register: lane i starts a run iff
        // p[i] - p[i-1] > TOLERANCE) into the read's leader lane; get_similar itself then runs ONCE for all reads of the pass,
        // on lane masks held as integers in the leader lanes (sga.erl:42-64).
        const unsigned long long bfall = __ballot(fuse_read && slot == 0);   // bit seg0 of every read to cluster
        const unsigned long long bfbig = BIG ? __ballot(fuse_read && slot == 0 && P > 64) : 0ull;
        unsigned long long bf = bfall & ~bfbig;
        uint32_t smlo = 0, smhi = 0, sm2lo = 0, sm2hi = 0;                   // run starts of elements 0..63 / 64..127
        uint32_t left = (uint32_t)__popcll(bf);
        while (left) {
            const uint32_t take = left == 1 ? 1u : (left == 2 || left == 4) ? 2u : 3u;
            uint32_t jj[3], pp[3];
            int bb[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                jj[t] = 0; bb[t] = 0; pp[t] = 0xffffffffu;
                if ((uint32_t)t < take) {
                    bb[t] = __ffsll((long long)bf) - 1;
                    bf &= bf - 1ull;
                    jj[t] = (uint32_t)bb[t] / maxT;
                    const uint32_t Pj = (uint32_t)__builtin_amdgcn_readlane((int)P, bb[t]);
                    pp[t] = lane < Pj ? rows[jj[t] * ROW + lane] : 0xffffffffu;
                }
            }
            if (take == 1) { uint32_t a[1] = {pp[0]}; mirror_sort64(a, SM); pp[0] = a[0]; }
            else if (take == 2) { uint32_t a[2] = {pp[0], pp[1]}; mirror_sort64(a, SM); pp[0] = a[0]; pp[1] = a[1]; }
            else mirror_sort64(pp, SM);
#pragma unroll
            for (int t = 0; t < 3; t++)
                if ((uint32_t)t < take) {
                    rows[jj[t] * ROW + lane] = pp[t];
                    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pp[t], 0x138, 0xF, 0xF, false);   // wave_shr:1
                    // Keep the shift a plain v_mov_b32_dpp: folded into the subtraction (v_subrev_u32_dpp d, p, p wave_shr:1) this
                    // part returns prev - p instead of p - prev (measured, scripts/dpp_probe.hip).
                    asm volatile("" : "+v"(prev));
                    const unsigned long long sm = __ballot(pp[t] - prev > BWT_TOLERANCE) | 1ull;     // (lanes >= P: masked below)
                    smlo = set_lane(smlo, (uint32_t)bb[t], (uint32_t)sm);
                    smhi = set_lane(smhi, (uint32_t)bb[t], (uint32_t)(sm >> 32));
                }
            left -= take;
        }
        for (unsigned long long bg = bfbig; bg; bg &= bg - 1ull) {
            const int b1 = __ffsll((long long)bg) - 1;
            const uint32_t j1 = (uint32_t)b1 / maxT;
            const uint32_t Pj = (uint32_t)__builtin_amdgcn_readlane((int)P, b1);
            uint32_t *row = rows + j1 * ROW;
            uint32_t a[2] = {row[lane], lane + 64 < Pj ? row[64 + lane] : 0xffffffffu};
            mirror_sort64(a, SM);
            const uint32_t rv = (uint32_t)__builtin_amdgcn_ds_bpermute(SM.mirror63, (int)a[1]);
            a[1] = a[0] < rv ? rv : a[0]; a[0] = a[0] < rv ? a[0] : rv;
            bitonic_merge64(a, SM);
            row[lane] = a[0]; row[64 + lane] = a[1];
            uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[0], 0x138, 0xF, 0xF, false);
            uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[1], 0x138, 0xF, 0xF, false);
            asm volatile("" : "+v"(p0), "+v"(p1));
            const uint32_t last0 = (uint32_t)__builtin_amdgcn_readlane((int)a[0], 63);
            if (lane == 0) p1 = last0;                                         // element 64 follows element 63
            const unsigned long long s0 = __ballot(a[0] - p0 > BWT_TOLERANCE) | 1ull, s1 = __ballot(a[1] - p1 > BWT_TOLERANCE);
            smlo = set_lane(smlo, (uint32_t)b1, (uint32_t)s0); smhi = set_lane(smhi, (uint32_t)b1, (uint32_t)(s0 >> 32));
            sm2lo = set_lane(sm2lo, (uint32_t)b1, (uint32_t)s1); sm2hi = set_lane(sm2hi, (uint32_t)b1, (uint32_t)(s1 >> 32));
        }
        __builtin_amdgcn_wave_barrier();
        // get_similar in the leader lanes: run ends = the lane before a start + the last seed; a run of >= 2 + Qty div 7 (:46)
        // seeds ending in element e <=> no start in (e - nthr + 1, e]; clusters in descending position order.  Elements
        // 64..127 (mask words *1) only exist in passes that hold a read of more than 64 seeds.
        const bool lead = fuse_read && slot == 0;
        const uint32_t tq = qty / 7;
        const unsigned long long vm0 = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
        const unsigned long long sm0 = (((unsigned long long)smhi << 32) | smlo) & vm0;
        const unsigned long long A0 = ~sm0;
        unsigned long long ok0 = A0, em0, sm1 = 0, em1 = 0;
        if (!bfbig) {
            for (uint32_t k = 1; __ballot(lead && k <= tq); k++) if (k <= tq) ok0 &= A0 << k;
            em0 = lead ? ((sm0 >> 1) | (1ull << ((P - 1) & 63u))) & vm0 & ok0 : 0ull;
        } else {
            const unsigned long long vm1 = P <= 64 ? 0ull : P >= 128 ? ~0ull : ((1ull << (P - 64)) - 1ull);
            sm1 = (((unsigned long long)sm2hi << 32) | sm2lo) & vm1;
            const unsigned long long A1 = ~sm1;
            unsigned long long ok1 = A1;
            for (uint32_t k = 1; __ballot(lead && k <= tq); k++)
                if (k <= tq) { ok0 &= A0 << k; ok1 &= (A1 << k) | (A0 >> (64 - k)); }       // (k <= 18 < 64)
            const unsigned long long last0 = P <= 64 ? 1ull << ((P - 1) & 63u) : 0ull, last1 = P > 64 ? 1ull << ((P - 65) & 63u) : 0ull;
            em0 = lead ? ((sm0 >> 1) | (sm1 << 63) | last0) & vm0 & ok0 : 0ull;
            em1 = lead ? ((sm1 >> 1) | last1) & vm1 & ok1 : 0ull;
        }
        const uint32_t Kc = (uint32_t)(__popcll(em0) + __popcll(em1));
        if (__ballot(Kc > CL_STAGE)) {                                 // a read with more clusters than its staging row holds: handed back
            const unsigned long long bk = __ballot(lead && Kc > CL_STAGE);      // (bit seg0 of every such read)
            why2 += (uint32_t)__popcll(bk) << 16;
            const bool d = valid && ((bk >> seg0) & 1ull) != 0;
            if (d) { fuse_read = false; em0 = 0; em1 = 0; }
            SF_HAND_BACK(d);
        }
        // the group's results
        {
            uint32_t *cl = r_cl + (uint64_t)r * (2 * CL_STAGE);
            const uint32_t *row = rows + lread * ROW;
            while (__ballot((em0 | em1) != 0)) {
                if (em0 | em1) {
                    int e1, s1;
                    if (em1) {
                        const int h = 63 - __clzll((long long)em1);
                        em1 &= ~(1ull << h);
                        e1 = 64 + h;
                        const unsigned long long below = sm1 & ((2ull << h) - 1ull);
                        s1 = below ? 64 + 63 - __clzll((long long)below) : 63 - __clzll((long long)sm0);
                    } else {
                        e1 = 63 - __clzll((long long)em0);
                        em0 &= ~(1ull << e1);
                        s1 = 63 - __clzll((long long)(sm0 & ((2ull << e1) - 1ull)));
                    }
                    const uint32_t pe = row[e1], pf = row[s1];
                    cl[0] = pe; cl[1] = pe - pf;
                    cl += 2;
                    dmaxl = pe - pf > dmaxl ? pe - pf : dmaxl;
                }
            }
        }
        if (valid && slot == 0) {
            // (r_pend and r_P of the reads this kernel keeps are zero: the host cleared both arrays before the launch)
            r_qty[r] = qty | (fuse_read ? RQ_DONE : 0u);
            if (fuse_read) { r_K[r] = Kc; Pacc += P; }
        }
    }
#undef SF_HAND_BACK
    waveq_drain(q, gq, &ctr->n_gq);
    if (lane == 0 && (why | why2)) {
        atomicAdd(&ctr->why[0], why & 255u); atomicAdd(&ctr->why[1], (why >> 8) & 255u); atomicAdd(&ctr->why[2], (why >> 16) & 255u);
        atomicAdd(&ctr->why[3], why >> 24); atomicAdd(&ctr->why[4], why2 & 0xffffu); atomicAdd(&ctr->why[5], why2 >> 16);
    }
    {
        const unsigned long long sums[3] = {(unsigned long long)live, (unsigned long long)live, wave_sum_u64(Pacc)};
        unsigned long long *const g64[3] = {&ctr->n_tasks, &ctr->kt_lookups, &ctr->sa_reads};
        unsigned int *const g32[3] = {nullptr, nullptr, nullptr};
        const uint32_t mx[1] = {wave_max_u32(dmaxl)};
        unsigned int *const gmx[1] = {&ctr->max_D};
        block_counters(sums, g64, g32, mx, gmx);
    }
}

// k_search_pending: the LF steps the table left over.  A wave scans rpc (64 for large batches) reads' r_pend words at a time and lists their
// pending tasks in LDS ((read of the chunk) << 6 | slot); its 16 quads take the tasks 16 at a time -- QUAD PER TASK, from
// the (Sp, Ep, n) k_seed_tasks stored, two 16-byte-per-lane block loads per step as in k_search -- and add what each
// task found to its read's LDS accumulators; then one lane per read totals the read.  No global queue, no global atomics.
__global__ __launch_bounds__(256) void k_search_pending(DevIndex ix, const uint32_t *__restrict__ planes,
                                                         const uint64_t *__restrict__ off, uint32_t n, uint32_t maxT,
                                                         const uint32_t *__restrict__ t_def, uint32_t *__restrict__ t_sp,
                                                         uint32_t *__restrict__ t_ep, uint32_t *__restrict__ t_info,
                                                         uint32_t *__restrict__ t_cnt,
                                                         const unsigned long long *__restrict__ r_pend,
                                                         uint32_t *__restrict__ r_P, uint32_t *__restrict__ r_qty,
                                                         uint32_t *__restrict__ bigq, uint32_t *__restrict__ mid2q,
                                                         uint32_t *__restrict__ mid4q, Counters *ctr, uint32_t rpc) {
    // rpc: reads per wave chunk (a power of two, 1..64): lanes 0 .. rpc-1 carry a read each
    if (ctr->n_pend == 0) return;                                 // no seed-kernel pass left a task behind: no need to look through r_pend
    extern __shared__ uint16_t pend_list_[];                      // 4 waves x 64 reads x maxT pending tasks (dynamic)
    __shared__ unsigned long long accP_[4][64];
    __shared__ uint32_t accQ_[4][64];                             // qty | crash << 31
    __shared__ uint32_t rq_[4 * READQ_LDS_WORDS];
    ReadQ rq;
    readq_init(rq, rq_ + (threadIdx.x >> 6) * READQ_LDS_WORDS);
    uint16_t *list = pend_list_ + (size_t)(threadIdx.x >> 6) * 64 * maxT;
    unsigned long long *accP = accP_[threadIdx.x >> 6];
    uint32_t *accQ = accQ_[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63;
    const int ql = (int)(lane & 3);
    const uint32_t qi = lane >> 2;
    const uint32_t dblk = (ix.dollar - 1) / BWT_BLOCK_ROWS;
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t steps = 0, nsmall = 0;
    unsigned long long Psum = 0;
    for (uint32_t chunk = wave; (uint64_t)chunk * rpc < n; chunk += nwaves) {
        const uint32_t rl = chunk * rpc + lane;
        const unsigned long long mine = (lane < rpc && rl < n) ? r_pend[rl] : 0ull;
        if (!__ballot(mine != 0)) continue;
        // task list: lane l's tasks at [excl_l, excl_l + popc(mine_l))
        uint32_t cntl = (uint32_t)__popcll(mine), incl = cntl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        {
            uint32_t w = incl - cntl;
            for (unsigned long long mm = mine; mm; mm &= mm - 1ull) list[w++] = (uint16_t)((lane << 6) | (uint32_t)(__ffsll((long long)mm) - 1));
        }
        accP[lane] = 0; accQ[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t b0 = 0; b0 < total; b0 += 16) {             // wave-uniform; quads step their tasks side by side
            const bool work = b0 + qi < total;
            const uint32_t ent = work ? list[b0 + qi] : 0u;
            const uint32_t rloc = ent >> 6, k = ent & 63u;
            const uint32_t r = chunk * rpc + rloc;
            LfTask t;
            t.task = (uint64_t)r * maxT + k; t.sp = 0; t.ep = 0; t.lo16 = 0; t.hi16 = 0; t.status = TS_NONE; t.n = 0; t.nlim = 0;
            t.check_at_limit = true; t.check = true; t.active = false;
            if (work) {
                const int e = (int)(t_def[t.task] & 0xffffu);
                task_init(t, ix, planes, off[r] + (uint64_t)e, e);
                const uint32_t n0 = (t_info[t.task] >> 8) & 0xffu;
                if (n0 > 1) { t.sp = t_sp[t.task]; t.ep = t_ep[t.task]; t.n = (int)n0; t.active = t.n < t.nlim; }
            }
            while (__ballot(t.active)) {
                if (t.active) {
                    const uint32_t b1 = (t.sp - 1) / BWT_BLOCK_ROWS, b2 = (t.ep - 1) / BWT_BLOCK_ROWS;
                    const uint4 v1 = ix.blocks[(uint64_t)b1 * 4 + ql], v2 = ix.blocks[(uint64_t)b2 * 4 + ql];
                    const uint32_t bi = 15u - (uint32_t)t.n;
                    steps++;
                    if (!lf_step(ix, ql, dblk, v1, v2, b1, b2, (t.lo16 >> bi) & 1u, (t.hi16 >> bi) & 1u, t.sp, t.ep)) {
                        t.check = false; t.active = false;
                    } else {
                        t.n++;
                        if (t.n >= t.nlim) { t.check = t.check_at_limit; t.active = false; }
                    }
                }
            }
            if (work && ql == 0) {
                uint32_t status = t.status;
                if (status == TS_SEEDS) {
                    if (t.n >= t.nlim && t.check) t.check = t.check_at_limit;
                    if (t.check && t.ep - t.sp > BWT_MAX_RANGE) status = TS_TOO_MANY;
                    else if (t.n < BWT_MIN_LEN) status = TS_NO_SEEDS;
                }
                const uint32_t c = status == TS_SEEDS ? t.ep - t.sp + 1 : 0;
                t_sp[t.task] = t.sp; t_ep[t.task] = t.ep; t_info[t.task] = status | ((uint32_t)t.n << 8); t_cnt[t.task] = c;
                if (status == TS_SEEDS) { atomicAdd(&accP[rloc], (unsigned long long)c); atomicAdd(&accQ[rloc], 1u); }
                else if (status == TS_CRASH) atomicOr(&accQ[rloc], RQ_CRASH);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // one lane per read: what k_seed_tasks had already counted + what the steps added
        {
            const bool have = mine != 0;
            unsigned long long P = 0;
            uint32_t qty = 0, crash = 0;
            if (have) {
                const uint32_t q0 = r_qty[rl], a = accQ[lane];
                P = (unsigned long long)r_P[rl] + accP[lane];
                qty = (q0 & 0x1fffffffu) + (a & 0x7fffffffu);
                crash = (q0 >> 31) | (a >> 31);
            }
            read_finish(have, rl, P, qty, crash, maxT, r_P, r_qty, bigq, mid2q, mid4q, ctr, Psum, nsmall, rq);
        }
        __builtin_amdgcn_wave_barrier();
    }
    readq_drain(rq, bigq, mid2q, mid4q, ctr);
    {
        const unsigned long long sums[3] = {wave_sum_u64(ql == 0 ? steps : 0), wave_sum_u64(Psum), wave_sum_u64(nsmall)};
        unsigned long long *const g64[3] = {&ctr->lf_steps, &ctr->sa_reads, nullptr};
        unsigned int *const g32[3] = {nullptr, nullptr, &ctr->n_small};
        const uint32_t mx[1] = {0u};
        unsigned int *const gmx[1] = {&ctr->max_D};
        block_counters(sums, g64, g32, mx, gmx);
    }
}

// ------------------------------------------------------------------------------------------------
// k_seed_list: for bwt_find_seeds_batch -- SA[Ep]+N, SA[Ep-1]+N, ..., SA[Sp]+N per task (fmi.erl:92-100)
__global__ __launch_bounds__(256) void k_seed_list(DevIndex ix, uint64_t n_tasks, const uint32_t *__restrict__ t_sp,
                                                    const uint32_t *__restrict__ t_info, const uint32_t *__restrict__ t_cnt,
                                                    const uint64_t *__restrict__ seed_off, int64_t *__restrict__ seeds) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (w >= n_tasks) return;
    const uint32_t cnt = t_cnt[w];
    if (!cnt) return;
    const uint32_t sp = t_sp[w], nm = (t_info[w] >> 8) & 0xff;
    const uint64_t o = seed_off[w];
    for (uint32_t i = lane; i < cnt; i += 64)
        seeds[o + i] = (int64_t)ix.sa[sp - 1 + (cnt - 1 - i)] + nm;
}

// ------------------------------------------------------------------------------------------------
// k_read_counts: lane per read.  P = total seeds, Qty = sub-reads that returned a seed list
// (sga.erl:32).  Reads with Qty < 2 cluster to [] (sga.erl:42-43) and get P = 0.
__global__ __launch_bounds__(256) void k_read_counts(uint32_t n, uint32_t maxT, const uint32_t *__restrict__ t_info,
                                                      const uint32_t *__restrict__ t_cnt, uint32_t *__restrict__ r_P,
                                                      uint32_t *__restrict__ r_qty, uint32_t *__restrict__ bigq,
                                                      uint32_t *__restrict__ mid2q, uint32_t *__restrict__ mid4q, Counters *ctr) {
    // grid-stride (capped grid): one counter atomic per wave for the whole launch, not per 64 reads
    __shared__ uint32_t rq_[4 * READQ_LDS_WORDS];
    ReadQ rq;
    readq_init(rq, rq_ + (threadIdx.x >> 6) * READQ_LDS_WORDS);
    unsigned long long Psum = 0;
    uint32_t nsmall = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t r = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        unsigned long long P = 0;
        uint32_t qty = 0, crash = 0;
        if (r < n)
            for (uint32_t k = 0; k < maxT; k++) {
                uint32_t st = t_info[(uint64_t)r * maxT + k] & 0xff;
                if (st == TS_SEEDS) { qty++; P += t_cnt[(uint64_t)r * maxT + k]; }
                else if (st == TS_CRASH) crash = 1;
            }
        read_finish(r < n, r, P, qty, crash, maxT, r_P, r_qty, bigq, mid2q, mid4q, ctr, Psum, nsmall, rq);
    }
    readq_drain(rq, bigq, mid2q, mid4q, ctr);
    unsigned long long s = wave_sum_u64(Psum), ns = wave_sum_u64(nsmall);
    if ((threadIdx.x & 63) == 0 && (s | ns)) { atomicAdd(&ctr->sa_reads, s); if (ns) atomicAdd(&ctr->n_small, (unsigned int)ns); }
}

// k_collect_small: after a fused seed stage only a few reads are left for k_cluster_small (more clusters than the
// staging row, tasks whose seeds were not inline, reads finished by k_search_pending): list them instead of launching
// a wave per read of the batch.  Lane per read; a wave gathers its finds in LDS and takes queue space 64 at a time
// (same-address atomics run at ~88 per us: one per wave-iteration would cost more than the clustering).
__global__ __launch_bounds__(256) void k_collect_small(uint32_t n, const uint32_t *__restrict__ r_P,
                                                        const uint32_t *__restrict__ r_qty, uint32_t *__restrict__ queue,
                                                        unsigned int *count, uint32_t cap, Counters *ctr) {
    __shared__ uint32_t buf_[4][128];
    uint32_t *buf = buf_[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    uint32_t have = 0;                                            // wave-uniform: entries waiting in buf
    for (uint32_t it = 0; it <= rounds; it++) {
        bool take = false;
        const uint32_t r = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        if (it < rounds && r < n) { const uint32_t P = r_P[r]; take = P > 0 && P <= 64 && !(r_qty[r] & RQ_DONE); }
        const unsigned long long m = __ballot(take);
        if (take) buf[have + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
        have += (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        const bool last = it == rounds;
        if (have >= 64 || (last && have)) {
            const uint32_t nout = have >= 64 ? 64u : have;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(count, nout);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (lane < nout && base + lane < cap) queue[base + lane] = buf[lane];
            if (lane == 0 && base + nout > cap) atomicOr(&ctr->err_flags, 8ull);      // more reads than were counted: never expected
            const uint32_t rest = lane + 64 < have ? buf[lane + 64] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (lane + 64 < have) buf[lane] = rest;
            have -= nout;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_cluster_small: wave per read with 2 <= P <= 64 seeds: gather SA entries (one per lane), bitonic
// sort across the wave, cluster with ballots (get_similar/7): a run of consecutive gaps <= 20 with
// Count >= 2 + Qty div 7 emits {LastPos, LastPos-FirstPos}; output in descending position order.
// Clusters are staged as (S,D) u32 pairs in the read's own slice of `pos` (K <= P/2 pairs fit).
__global__ __launch_bounds__(256) void k_cluster_small(DevIndex ix, uint32_t n, uint32_t maxT,
                                                        const uint32_t *__restrict__ t_def,
                                                        const uint32_t *__restrict__ t_sp,
                                                        const uint32_t *__restrict__ t_info,
                                                        const uint32_t *__restrict__ t_cnt,
                                                        const uint32_t *__restrict__ r_P,
                                                        const uint32_t *__restrict__ r_qty,
                                                        const uint64_t *__restrict__ seed_off,
                                                        uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K,
                                                        const uint32_t *__restrict__ queue, const unsigned int *qcount) {
    const uint32_t w = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (w >= n || (queue && w >= *qcount)) return;
    const uint32_t r = queue ? queue[w] : w;              // queue: the reads k_collect_small listed
    const uint32_t P = r_P[r], q0 = r_qty[r];
    if (q0 & RQ_DONE) return;                             // clustered by k_seed_tasks already
    if (P == 0) { if (lane == 0) r_K[r] = 0; return; }
    if (P > 64 || maxT > 64) return;                      // k_cluster_big owns this read
    const uint32_t qty = q0 & 0x1fffffffu;
    uint32_t cnt = 0, sp = 0, nadd = 0;
    if (lane < maxT) {
        const uint64_t t = (uint64_t)r * maxT + lane;
        cnt = t_cnt[t]; sp = t_sp[t];
        nadd = ((t_info[t] >> 8) & 0xff) + (t_def[t] >> 16);   // N + Shift + Skip (sga.erl:32,37-39)
    }
    uint32_t incl = cnt;
    for (uint32_t o = 1; o < maxT; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const uint32_t excl = incl - cnt;
    // seed i of the read belongs to the sub-read whose slice [excl, excl+cnt) holds i: the sub-read lanes drop their
    // id at their slice start in a wave-private LDS row, a ballot + clz finds every lane's slice start, and three
    // ds_bpermutes fetch that sub-read's {Sp, N+Shift+Skip}
    __shared__ uint32_t mark[4][64];
    uint32_t *mk = mark[threadIdx.x >> 6];
    mk[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < maxT && cnt) mk[excl] = lane + 1;           // excl < P <= 64
    __builtin_amdgcn_wave_barrier();
    const uint32_t tk = mk[lane];
    uint32_t p = 0xffffffffu;
    {
        const unsigned long long tmask = __ballot(tk != 0);
        const unsigned long long tb = tmask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int s0 = 63 - __clzll(tb | 1ull);              // first seed slot of my sub-read
        const uint32_t tid = __shfl(tk, s0, 64);             // its lane + 1
        const uint32_t msp = __shfl(sp, (int)tid - 1, 64), madd = __shfl(nadd, (int)tid - 1, 64);
        if (lane < P) p = ix.sa[msp - 1 + (lane - (uint32_t)s0)] + madd;
    }
    p = wave_sort_u32(p);                                 // ascending over the 64 lanes
    const uint32_t prev = __shfl_up(p, 1, 64), next = __shfl_down(p, 1, 64);
    const bool valid = lane < P;
    const bool start = valid && (lane == 0 || p - prev > BWT_TOLERANCE);
    const bool end = valid && (lane == P - 1 || next - p > BWT_TOLERANCE);
    const unsigned long long smask = __ballot(start);
    const unsigned long long below = smask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int s_i = 63 - __clzll(below | 1ull);          // lane of this run's first element
    const uint32_t pfirst = __shfl(p, s_i, 64);
    const uint32_t nthr = 2 + qty / 7;                    // sga.erl:46
    const bool emit = end && (lane - (uint32_t)s_i + 1 >= nthr);
    const unsigned long long emask = __ballot(emit);
    if (emit) {
        const uint32_t kk = __popcll(lane == 63 ? 0ull : (emask >> (lane + 1)));   // emitters above me
        const uint64_t o = seed_off[r] + 2ull * kk;
        pos[o] = p;
        pos[o + 1] = p - pfirst;
    }
    if (lane == 0) r_K[r] = (uint32_t)__popcll(emask);
}

// k_cluster_wave<R>: wave per read with 64 < P <= 64*R seeds (R = 2, 4; 150 bp reads on a large index sit
// here).  Same algorithm as k_cluster_small with R elements per lane, element e = r*64 + lane: bitonic
// strides >= 64 are register-to-register exchanges, smaller strides are lane shuffles; run starts and
// emitters carry across the R chunks as wave-uniform scalars.
// (the read's work as a device function: k_cluster_wave runs it on a queue entry, k_cluster_rest -- the small-batch path -- on all
// three queues of reads with more than 64 seeds in one launch; mk = 2 * 64 R words of wave-private LDS)
template <int R>
__device__ __forceinline__ void cluster_wave_read(const DevIndex &ix, uint32_t r, uint32_t maxT, const uint32_t *__restrict__ t_def,
                                                  const uint32_t *__restrict__ t_sp, const uint32_t *__restrict__ t_info,
                                                  const uint32_t *__restrict__ t_cnt, const uint32_t *__restrict__ r_P,
                                                  const uint32_t *__restrict__ r_qty, const uint64_t *__restrict__ seed_off,
                                                  uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K, uint32_t *mk) {
    constexpr int N = 64 * R;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t P = r_P[r];
    const uint32_t qty = r_qty[r] & 0x7fffffffu;
    uint32_t *srt = mk + N;
    uint32_t cnt = 0, sp = 0, nadd = 0;
    if (lane < maxT) {
        const uint64_t t = (uint64_t)r * maxT + lane;
        cnt = t_cnt[t]; sp = t_sp[t];
        nadd = ((t_info[t] >> 8) & 0xff) + (t_def[t] >> 16);   // N + Shift + Skip (sga.erl:32,37-39)
    }
    uint32_t incl = cnt;
    for (uint32_t o = 1; o < maxT; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const uint32_t excl = incl - cnt;
#pragma unroll
    for (int q = 0; q < R; q++) mk[q * 64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < maxT && cnt) mk[excl] = lane + 1;           // excl < P <= 64*R
    __builtin_amdgcn_wave_barrier();
    uint32_t p[R];
    {
        uint32_t carry_tid = 0, carry_start = 0;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const uint32_t tk = mk[q * 64 + lane];
            const unsigned long long tmask = __ballot(tk != 0);
            const unsigned long long tb = tmask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
            uint32_t tid = carry_tid, start = carry_start;
            const int s0 = 63 - __clzll(tb | 1ull);
            const uint32_t tid_here = __shfl(tk, s0, 64);
            if (tb) { tid = tid_here; start = (uint32_t)(q * 64 + s0); }
            if (tmask) {
                const int top = 63 - __clzll(tmask);
                carry_tid = (uint32_t)__builtin_amdgcn_readlane((int)tk, top);
                carry_start = (uint32_t)(q * 64 + top);
            }
            const uint32_t i = (uint32_t)(q * 64) + lane;
            const uint32_t msp = __shfl(sp, (int)tid - 1, 64), madd = __shfl(nadd, (int)tid - 1, 64);
            p[q] = i < P ? ix.sa[msp - 1 + (i - start)] + madd : 0xffffffffu;
        }
    }
    // bitonic sort ascending over N = 64*R elements, element e = q*64 + lane
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int q = 0; q < R; q++) {
                    if (q & (j / 64)) continue;
                    const int q2 = q + j / 64;
                    const bool up = ((q * 64) & k) == 0;          // k >= 128 here: decided by the chunk index
                    const uint32_t a = p[q], b = p[q2];
                    const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
                    p[q] = up ? mn : mx; p[q2] = up ? mx : mn;
                }
            } else {
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const uint32_t o = __shfl_xor(p[q], j, 64);
                    const bool up = k >= 64 ? (((q * 64) & k) == 0) : ((lane & k) == 0);
                    const bool lower = (lane & j) == 0;
                    const uint32_t mn = p[q] < o ? p[q] : o, mx = p[q] < o ? o : p[q];
                    p[q] = (lower == up) ? mn : mx;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < R; q++) srt[q * 64 + lane] = p[q];
    __builtin_amdgcn_wave_barrier();
    const uint32_t nthr = 2 + qty / 7;                    // sga.erl:46
    // pass 1: run starts, emit flags per chunk (ascending), carrying the last run start across chunks
    unsigned long long emask[R];
    uint32_t s_of[R];
    {
        uint32_t carry_s = 0;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const uint32_t e = (uint32_t)(q * 64) + lane;
            const uint32_t prev = e ? srt[e - 1] : 0u, next = e + 1 < (uint32_t)N ? srt[e + 1] : 0xffffffffu;
            const bool valid = e < P;
            const bool start = valid && (e == 0 || p[q] - prev > BWT_TOLERANCE);
            const bool end = valid && (e == P - 1 || next - p[q] > BWT_TOLERANCE);
            const unsigned long long smask = __ballot(start);
            const unsigned long long below = smask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
            uint32_t s_e = carry_s;
            if (below) s_e = (uint32_t)(q * 64 + (63 - __clzll(below)));
            if (smask) carry_s = (uint32_t)(q * 64 + (63 - __clzll(smask)));
            s_of[q] = s_e;
            emask[q] = __ballot(end && (e - s_e + 1 >= nthr));
        }
    }
    // pass 2: output in descending position order: rank = emitters above me
    uint32_t above = 0;
#pragma unroll
    for (int q = R - 1; q >= 0; q--) {
        const bool emit = (emask[q] >> lane) & 1ull;
        if (emit) {
            const uint32_t kk = above + (uint32_t)__popcll(lane == 63 ? 0ull : (emask[q] >> (lane + 1)));
            const uint64_t o = seed_off[r] + 2ull * kk;
            pos[o] = p[q];
            pos[o + 1] = p[q] - srt[s_of[q]];
        }
        above += (uint32_t)__popcll(emask[q]);
    }
    if (lane == 0) r_K[r] = above;
}
template <int R>
__global__ __launch_bounds__(256) void k_cluster_wave(DevIndex ix, uint32_t nq, const uint32_t *__restrict__ queue,
                                                       uint32_t maxT, const uint32_t *__restrict__ t_def,
                                                       const uint32_t *__restrict__ t_sp,
                                                       const uint32_t *__restrict__ t_info,
                                                       const uint32_t *__restrict__ t_cnt,
                                                       const uint32_t *__restrict__ r_P,
                                                       const uint32_t *__restrict__ r_qty,
                                                       const uint64_t *__restrict__ seed_off,
                                                       uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K,
                                                       const unsigned int *__restrict__ nq_dev, uint32_t lo, uint32_t hi) {
    // nq_dev != null: the queue length is read on the device (nq = the most the queue can hold)
    // lo / hi: the launch takes the reads of its queue with lo < P <= hi seeds (R = 8, 16 on the queue of reads past 256 seeds: a wave
    // per read and the sort in registers up to 1024 seeds -- 0.83 ms as one wave per read sorting in LDS, round 4)
    __shared__ uint32_t lds[4][2 * 64 * R];
    const uint32_t w = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (nq_dev) nq = *nq_dev;
    if (w >= nq) return;
    const uint32_t r = queue[w], P = r_P[r];
    if (P <= lo || P > hi) return;
    cluster_wave_read<R>(ix, r, maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, lds[threadIdx.x >> 6]);
}

// k_cluster_big: workgroup per read (P > 256): gather, sort (ascending-only bitonic network: virtual +inf padding past P), cluster.
// (Reads of up to 1024 seeds are k_cluster_wave<8 | 16>'s, launched on the same queue with a seed-count filter: a wave per read, sorted in
// registers.  Tried between the two: this kernel as ONE WAVE per read -- NT = 64, no barrier between the sort's stages, 5 KB of LDS, 32
// reads per CU in flight where blocks of 256 threads, most of them idle on 300..550 seeds, held 8 -- 0.92 ms against 1.38; the register
// sort takes 0.32.  NT stays a parameter.)
// Round 4: (i) the seeds are gathered straight into LDS and stay there; (ii) the runs are found by the whole block -- a thread takes
// a stretch of the sorted list, two block scans carry the last run start and the number of clusters across stretches -- where
// one thread walked the P elements (20 us for 500 seeds: most of a read's time); (iii) a read of more seeds than this kernel's LDS
// holds (a low-complexity read whose sub-reads die in not_found with whole intervals: thousands of seeds) is LISTED for
// k_cluster_huge -- 1024 threads, 120 KB of LDS -- instead of being sorted in HBM by 256 threads: one such read took as long as
// the other 70 000 of a repeat-rich batch together (2.2 ms).  What is larger still keeps the HBM path.
#define BIG_LDS 4096
#define HUGE_LDS 30720
// inclusive scan of one value per thread over the block (NT threads, tmp: NT words of LDS); MAXOP: maximum, else sum
template <int NT, bool MAXOP>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *tmp) {
    tmp[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (uint32_t d = 1; d < (uint32_t)NT; d <<= 1) {
        const uint32_t o = threadIdx.x >= d ? tmp[threadIdx.x - d] : 0u;
        __syncthreads();
        v = MAXOP ? (o > v ? o : v) : v + o;
        tmp[threadIdx.x] = v;
        __syncthreads();
    }
    return v;
}
// get_similar (src/sga.erl:52-64) over the sorted list a[0..P) in LDS by the whole block: runs of neighbours at most BWT_TOLERANCE
// apart, a run of at least nthr seeds is a cluster {its last seed, its width}; clusters go to reg in DESCENDING order (sga.erl prepends).
template <int NT>
__device__ __forceinline__ void cluster_runs_block(const uint32_t *a, uint32_t P, uint32_t nthr, uint32_t *__restrict__ reg, uint32_t *__restrict__ r_K_r,
                                                   uint32_t *tmp) {
    const uint32_t C = (P + NT - 1) / NT;
    const uint32_t lo = threadIdx.x * C < P ? threadIdx.x * C : P, hi = lo + C < P ? lo + C : P;
    uint32_t last = 0;                                        // index + 1 of the last run start in my stretch
    for (uint32_t i = lo; i < hi; i++)
        if (i == 0 || a[i] - a[i - 1] > BWT_TOLERANCE) last = i + 1;
    (void)block_scan_incl<NT, true>(last, tmp);
    // the last run start in front of my stretch: the inclusive maximum of the thread before me
    const uint32_t start0 = threadIdx.x ? tmp[threadIdx.x - 1] : 0u;
    __syncthreads();
    uint32_t cur = start0, cnt = 0;
    for (uint32_t i = lo; i < hi; i++) {
        if (i == 0 || a[i] - a[i - 1] > BWT_TOLERANCE) cur = i + 1;
        const bool end = i + 1 == P || a[i + 1] - a[i] > BWT_TOLERANCE;
        if (end && i + 2 - cur >= nthr) cnt++;                // seeds in the run: i - (cur - 1) + 1
    }
    const uint32_t incl = block_scan_incl<NT, false>(cnt, tmp);
    const uint32_t K = tmp[NT - 1];
    uint32_t rank = incl - cnt;
    cur = start0;
    for (uint32_t i = lo; i < hi; i++) {
        if (i == 0 || a[i] - a[i - 1] > BWT_TOLERANCE) cur = i + 1;
        const bool end = i + 1 == P || a[i + 1] - a[i] > BWT_TOLERANCE;
        if (end && i + 2 - cur >= nthr) {
            const uint32_t k = K - 1 - rank;
            reg[2 * k] = a[i]; reg[2 * k + 1] = a[i] - a[cur - 1];
            rank++;
        }
    }
    if (threadIdx.x == 0) *r_K_r = K;
    __syncthreads();                                          // (tmp and a are reused by the block's next read)
}
// (one read, by a whole block of NT threads; sh = cap words of block-shared LDS, tmp = NT more.  Every thread of the block calls it.
// hugeq / n_huge: where a read of more than cap seeds is listed instead; null: it is sorted in HBM and clustered by one thread)
template <int NT>
__device__ __forceinline__ void cluster_big_read(const DevIndex &ix, uint32_t r, uint32_t maxT, const uint32_t *__restrict__ t_def,
                                                 const uint32_t *__restrict__ t_sp, const uint32_t *__restrict__ t_info,
                                                 const uint32_t *__restrict__ t_cnt, const uint32_t *__restrict__ r_P,
                                                 const uint32_t *__restrict__ r_qty, const uint64_t *__restrict__ seed_off,
                                                 uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K, uint32_t *sh, uint32_t cap, uint32_t *tmp,
                                                 uint32_t *__restrict__ hugeq, unsigned int *__restrict__ n_huge) {
    __syncthreads();                                      // (sh is reused from one read to the next)
    const uint32_t P = r_P[r];
    const bool in_lds = P <= cap;
    if (!in_lds && hugeq) {
        if (threadIdx.x == 0) hugeq[atomicAdd(n_huge, 1u)] = r;
        return;
    }
    uint32_t *reg = pos + seed_off[r];
    uint32_t *a = in_lds ? sh : reg;
    if (maxT <= 64) {
        // the tasks' rows first, all at once (a thread per task), then every seed of the read in one sweep: two round trips to memory
        // per read where a loop over the tasks made two per TASK (12 tasks: ~20 us of a 25 us read)
        if (threadIdx.x < maxT) {
            const uint64_t t = (uint64_t)r * maxT + threadIdx.x;
            const uint32_t cnt = t_cnt[t];
            tmp[threadIdx.x] = cnt;
            tmp[64 + threadIdx.x] = t_sp[t];
            tmp[128 + threadIdx.x] = ((t_info[t] >> 8) & 0xff) + (t_def[t] >> 16);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < P; i += NT) {
            uint32_t k = 0, o = 0;
            for (; k + 1 < maxT && i >= o + tmp[k]; k++) o += tmp[k];
            a[i] = ix.sa[tmp[64 + k] - 1 + (i - o)] + tmp[128 + k];
        }
    } else {
        uint32_t o = 0;
        for (uint32_t k = 0; k < maxT; k++) {
            const uint64_t t = (uint64_t)r * maxT + k;
            const uint32_t cnt = t_cnt[t];
            if (!cnt) continue;
            const uint32_t sp = t_sp[t], nadd = ((t_info[t] >> 8) & 0xff) + (t_def[t] >> 16);
            for (uint32_t i = threadIdx.x; i < cnt; i += NT) a[o + i] = ix.sa[sp - 1 + i] + nadd;
            o += cnt;
        }
    }
    __syncthreads();
    uint32_t n2 = 1;
    while (n2 < P) n2 <<= 1;
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        for (uint32_t i = threadIdx.x; i < P; i += NT) {      // flip step
            uint32_t j = i ^ (k - 1);
            if (j > i && j < P) { uint32_t x = a[i], y = a[j]; if (x > y) { a[i] = y; a[j] = x; } }
        }
        __syncthreads();
        for (uint32_t js = k >> 2; js > 0; js >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += NT) {
                uint32_t j = i ^ js;
                if (j > i && j < P) { uint32_t x = a[i], y = a[j]; if (x > y) { a[i] = y; a[j] = x; } }
            }
            __syncthreads();
        }
    }
    const uint32_t nthr = 2 + (r_qty[r] & 0x7fffffffu) / 7;
    if (in_lds) { cluster_runs_block<NT>(a, P, nthr, reg, r_K + r, tmp); return; }
    if (threadIdx.x == 0) {
        uint32_t K = 0, p1 = a[0], first = a[0], count = 1;
        for (uint32_t i = 1; i <= P; i++) {
            const bool endl = (i == P);
            const uint32_t p2 = endl ? 0 : a[i];
            if (!endl && p2 - p1 <= BWT_TOLERANCE) { count++; p1 = p2; continue; }
            if (count >= nthr) { reg[2 * K] = p1; reg[2 * K + 1] = p1 - first; K++; }   // 2K+1 < i: already consumed
            p1 = p2; first = p2; count = 1;
        }
        for (uint32_t x = 0, y = K ? K - 1 : 0; x < y; x++, y--) {       // descending order (sga.erl prepends)
            uint32_t t0 = reg[2 * x], t1 = reg[2 * x + 1];
            reg[2 * x] = reg[2 * y]; reg[2 * x + 1] = reg[2 * y + 1];
            reg[2 * y] = t0; reg[2 * y + 1] = t1;
        }
        r_K[r] = K;
    }
}
// lo / hi: the kernel takes the reads of its queue with lo < P <= hi seeds (the others are another launch's)
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void k_cluster_big(DevIndex ix, uint32_t maxT, const uint32_t *__restrict__ bigq,
                                                     const uint32_t *__restrict__ t_def,
                                                     const uint32_t *__restrict__ t_sp,
                                                     const uint32_t *__restrict__ t_info,
                                                     const uint32_t *__restrict__ t_cnt,
                                                     const uint32_t *__restrict__ r_P,
                                                     const uint32_t *__restrict__ r_qty,
                                                     const uint64_t *__restrict__ seed_off,
                                                     uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K,
                                                     const unsigned int *__restrict__ nq_dev, uint32_t nq, uint32_t *__restrict__ hugeq,
                                                     unsigned int *__restrict__ n_huge, uint32_t lo, uint32_t hi) {
    __shared__ uint32_t sh[CAP + 256];
    // grid-stride over the queue (nq_dev != null: its length is read on the device)
    if (nq_dev) nq = *nq_dev;
    for (uint32_t bi = blockIdx.x; bi < nq; bi += gridDim.x) {
        const uint32_t r = bigq[bi], P = r_P[r];
        if (P <= lo || P > hi) continue;
        cluster_big_read<NT>(ix, r, maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, sh, CAP, sh + CAP, hugeq, n_huge);
    }
}
// the reads k_cluster_big listed: up to HUGE_LDS seeds each in LDS (dynamic: HUGE_LDS + 1024 words), a block of 1024 threads per read
__global__ __launch_bounds__(1024) void k_cluster_huge(DevIndex ix, uint32_t maxT, const uint32_t *__restrict__ hugeq, const unsigned int *__restrict__ n_huge,
                                                        const uint32_t *__restrict__ t_def, const uint32_t *__restrict__ t_sp,
                                                        const uint32_t *__restrict__ t_info, const uint32_t *__restrict__ t_cnt,
                                                        const uint32_t *__restrict__ r_P, const uint32_t *__restrict__ r_qty,
                                                        const uint64_t *__restrict__ seed_off, uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K) {
    extern __shared__ uint32_t dyn_huge[];
    const uint32_t nq = *n_huge;
    for (uint32_t bi = blockIdx.x; bi < nq; bi += gridDim.x)
        cluster_big_read<1024>(ix, hugeq[bi], maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, dyn_huge, HUGE_LDS, dyn_huge + HUGE_LDS,
                               (uint32_t *)nullptr, (unsigned int *)nullptr);
}
// the three queues of reads with more than 64 seeds in ONE launch (the small-batch path: queue lengths read on the device, a
// launch saved twice): waves take the entries of the 65..128 and 129..256 queues, then blocks those of the workgroup-per-read one
__global__ __launch_bounds__(256) void k_cluster_rest(DevIndex ix, uint32_t maxT, const uint32_t *__restrict__ mid2q, const uint32_t *__restrict__ mid4q,
                                                       const uint32_t *__restrict__ bigq, const Counters *__restrict__ ctr,
                                                       const uint32_t *__restrict__ t_def, const uint32_t *__restrict__ t_sp,
                                                       const uint32_t *__restrict__ t_info, const uint32_t *__restrict__ t_cnt,
                                                       const uint32_t *__restrict__ r_P, const uint32_t *__restrict__ r_qty,
                                                       const uint64_t *__restrict__ seed_off, uint32_t *__restrict__ pos, uint32_t *__restrict__ r_K) {
    __shared__ uint32_t lds[BIG_LDS + 256];                // 4 waves x 2 x 256 words for the wave kernels = 2048; BIG_LDS (+ 256 of scan space) for the block one
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n2 = ctr->n_mid2, n4 = ctr->n_mid4, nb = ctr->n_big;
    uint32_t *mine = lds + (threadIdx.x >> 6) * 512;
    for (uint32_t w = wave; w < n2; w += nwaves) {
        cluster_wave_read<2>(ix, mid2q[w], maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, mine);
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t w = wave; w < n4; w += nwaves) {
        cluster_wave_read<4>(ix, mid4q[w], maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, mine);
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t bi = blockIdx.x; bi < nb; bi += gridDim.x)
        cluster_big_read<256>(ix, bigq[bi], maxT, t_def, t_sp, t_info, t_cnt, r_P, r_qty, seed_off, pos, r_K, lds, BIG_LDS, lds + BIG_LDS,
                              (uint32_t *)nullptr, (unsigned int *)nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_candidates: lane per read: staged clusters -> dense candidate arrays + the window of
// worker_bwt.erl:86-88:  Ref_len = Lq + D;  Start = S - Ref_len + 200 (offset into Ref1).
__global__ __launch_bounds__(256) void k_candidates(uint32_t n, const uint64_t *__restrict__ off,
                                                     const uint32_t *__restrict__ r_K,
                                                     const uint64_t *__restrict__ seed_off,
                                                     const uint64_t *__restrict__ clus_off,
                                                     const uint32_t *__restrict__ pos, const uint32_t *__restrict__ r_qty,
                                                     const uint32_t *__restrict__ r_cl, uint32_t *__restrict__ c_read,
                                                     uint32_t *__restrict__ c_S, uint32_t *__restrict__ c_D,
                                                     uint64_t *__restrict__ c_qoff, uint32_t *__restrict__ c_qlen,
                                                     int64_t *__restrict__ c_woff, uint32_t *__restrict__ c_wlen,
                                                     Counters *ctr) {
    unsigned long long cells = 0, wbytes = 0;
    uint32_t maxw = 0, maxlq = 0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {   // grid-stride, capped grid
        const uint32_t K = r_K[r];
        if (K) {
            const uint64_t qo = off[r];
            const uint32_t lq = (uint32_t)(off[r + 1] - qo);
            const uint64_t co = clus_off[r];
            // clusters staged by k_seed_tasks (RQ_DONE) sit in the read's r_cl row, the cluster kernels' in its slice of pos
            const uint32_t *cl = (r_qty[r] & RQ_DONE) ? r_cl + (uint64_t)r * (2 * CL_STAGE) : pos + seed_off[r];
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t S = cl[2 * k], D = cl[2 * k + 1];
                const uint32_t wl = lq + D;
                c_read[co + k] = r; c_S[co + k] = S; c_D[co + k] = D;
                c_qoff[co + k] = qo; c_qlen[co + k] = lq;
                c_wlen[co + k] = wl;
                c_woff[co + k] = (int64_t)S - (int64_t)wl + BWT_REF_EXT;
                cells += (unsigned long long)lq * wl; wbytes += wl;
                maxw = wl > maxw ? wl : maxw;
            }
            maxlq = lq > maxlq ? lq : maxlq;
        }
    }
    cells = wave_sum_u64(cells); wbytes = wave_sum_u64(wbytes);
    maxw = wave_max_u32(maxw); maxlq = wave_max_u32(maxlq);
    if ((threadIdx.x & 63) == 0 && maxw) {
        atomicAdd(&ctr->sw_cells, cells); atomicAdd(&ctr->window_bytes, wbytes);
        atomicMax(&ctr->max_w, maxw); atomicMax(&ctr->max_lq, maxlq);
    }
}

// ------------------------------------------------------------------------------------------------
// Smith-Waterman (semi-global, direction-dependent gap cost) -- one LANE per alignment.
// The query is cut into strips of SW_R rows held in registers; a strip sweeps all W columns; the
// row between two strips is kept per wave as [j][lane] -- in LDS (16-bit packed cells) for the
// common window widths, in an HBM slab otherwise.
//
// Cell (src/sw.erl:88-98):  {V,Dir} = lists:max([{Vul+sigma,up_left},{Lvalue,left},{Uvalue,up}])
// Term order makes that a lexicographic max on (V, Dir) with left < up < up_left, so the integer
// path packs a cell as 4*V + code (left=0, up=1, up_left=2, undef=3) and one max3 does both.
// Lvalue = Vleft + (Dleft==left ? -1 : -2), Uvalue = Vup + (Dup==up ? -1 : -2)  (sw.erl:103-107).
// The per-row abort test (sw.erl:69) is equivalent to the test after the last row because
// RowMax_{i+1} <= RowMax_i + 2, so only the last row's maximum is tracked.
//
// Three instantiations:
//   SW_SCORE  integer, no direction matrix: finds the score and the start column, then checks whether
//             the pure diagonal ending there already reaches the score.  If it does, every cell on
//             that diagonal was stored as up_left (ties prefer up_left), so the reference's traceback
//             walks it and the CIGAR is "<Lq>M"; the alignment is finished without any HBM traffic
//             beyond its query and window.  Otherwise the candidate is queued for SW_TRACE.
//   SW_TRACE  integer, 2-bit directions to a per-wave HBM slab [strip][j][lane], fused traceback.
//   SW_FP     the same with IEEE doubles (any N / IUPAC: sigma may return 1.6), fused traceback.
#define SW_R 16
#define CIG_DIAG 0x80000000u      // c_ciglen flag: the text is "<window length>M" and was not written to the slot
enum { DIR_LEFT = 0, DIR_UP = 1, DIR_UL = 2, DIR_UNDEF = 3 };
enum { SW_SCORE = 0, SW_TRACE = 1, SW_FP = 2 };

__device__ __forceinline__ double sigma_gen(uint32_t s, uint32_t r) {   // src/sw.erl:164-202
    if (s == 'N' && r == 'N') return 1.6;
    if (s == r) return 2.0;
    if (s == 'N' || r == 'N') return 1.6;
    switch (r) {
    case 'R': return (s == 'A' || s == 'G') ? 2.0 : -1.0;
    case 'Y': return (s == 'C' || s == 'T') ? 2.0 : -1.0;
    case 'B': return s == 'A' ? -1.0 : 2.0;
    case 'D': return s == 'C' ? -1.0 : 2.0;
    case 'K': return (s == 'G' || s == 'T') ? 2.0 : -1.0;
    case 'M': return (s == 'A' || s == 'C') ? 2.0 : -1.0;
    case 'S': return (s == 'G' || s == 'C') ? 2.0 : -1.0;
    case 'W': return (s == 'A' || s == 'T') ? 2.0 : -1.0;
    case 'V': return s == 'T' ? -1.0 : 2.0;
    default: return -1.0;
    }
}

struct SwIO {
    const uint8_t *qbuf, *rbuf;
    uint64_t rbuf_len;
    const uint64_t *c_qoff;
    const uint32_t *c_qlen;
    const int64_t *c_woff;
    const uint32_t *c_wlen;
    int32_t *c_status, *c_score;
    uint32_t *c_ciglen;
    uint8_t *cig;              // n_cand x cig_stride, text right-aligned in its slot
    uint32_t cig_stride;
    uint32_t *fpq;             // queue of candidates for the fp64 kernel
    uint32_t *tq;              // queue of candidates that need a real traceback
    uint32_t *dq;              // queue of candidates that need the DP at all (filled by k_sw_prefilter)
    uint32_t *tq2;             // traceback queue of the packed kernels (same read length, window <= cap)
    const uint32_t *qplanes, *wplanes;   // bit-planes over qbuf / rbuf (k_read_planes layout)
    uint32_t wave_upto;        // a DP queue of at most this many candidates is k_sw_wave's: the score kernels leave it alone
    const uint32_t *tq_in;     // k_sw<SW_TRACE>: its queue and count when they are not tq / ctr->tq_count (what the wide pass of k_sw_pk left)
    const unsigned int *tq_in_count;
    Counters *ctr;
    uint32_t *dq_lo = nullptr; // k_sw_prefilter: DP candidates on a SQUARE window with 3..lo_max_m mismatches on its diagonal go here instead of dq
                               // (count: ctr->dq_lo_count): the 12-diagonal tier of the banded DP settles exactly those with up to 6; null: all to dq
    uint32_t lo_max_m = 6;     // ... 6 without the certificate kernels in front of that tier, as many as their widest instance takes with them
};

// writes "<n><op>" right to left, returns the new write position
__device__ __forceinline__ void cig_emit(uint8_t *slot, int &p, int &len, uint32_t n, char op) {
    len++; if (--p >= 0) slot[p] = (uint8_t)op;
    do { len++; if (--p >= 0) slot[p] = (uint8_t)('0' + n % 10); n /= 10; } while (n);
}
// ... by every lane of a wave that walks a traceback together (same p / len in all of them), the bytes by the lane `wr` picks
__device__ __forceinline__ void cig_emit_if(bool wr, uint8_t *slot, int &p, int &len, uint32_t n, char op) {
    len++; if (--p >= 0 && wr) slot[p] = (uint8_t)op;
    do { len++; if (--p >= 0 && wr) slot[p] = (uint8_t)('0' + n % 10); n /= 10; } while (n);
}

// (the kernel's work as a device function of (wave, nwaves): k_sw is the plain launch, k_sw_tail -- the small-batch path -- runs the
// traced and the fp64 instance side by side in one launch)
template <int MODE, bool LDSB>
__device__ __forceinline__ void sw_lanes(const SwIO &io, uint32_t n_items, uint32_t maxW1, uint32_t max_strips,
                                         uint32_t *__restrict__ slab_bnd, double *__restrict__ slab_bndv,
                                         uint32_t *__restrict__ slab_dir, const uint32_t wave, const uint32_t nwaves) {
    constexpr bool FP = MODE == SW_FP;
    constexpr bool DIRS = MODE != SW_SCORE;
    extern __shared__ uint16_t lds_bnd[];
    const uint32_t lane = threadIdx.x & 63;
    uint16_t *bl = LDSB ? lds_bnd + (uint64_t)(threadIdx.x >> 6) * maxW1 * 64 + lane : nullptr;   // [j*64]
    uint32_t *bnd = LDSB ? nullptr : slab_bnd + (uint64_t)wave * maxW1 * 64 + lane;                // [j*64]
    double *bndv = FP ? slab_bndv + (uint64_t)wave * maxW1 * 64 + lane : nullptr;
    uint32_t *dirs = DIRS ? slab_dir + (uint64_t)wave * max_strips * maxW1 * 64 + lane : nullptr;  // [(s*maxW1+j)*64]
    const uint32_t n_total = MODE == SW_FP ? io.ctr->fp_count : MODE == SW_TRACE ? (io.tq_in ? *io.tq_in_count : io.ctr->tq_count) : io.dq ? io.ctr->dq_count : n_items;
    if (MODE == SW_SCORE && io.dq && n_total <= io.wave_upto) return;

    for (uint32_t g = wave; (uint64_t)g * 64 < n_total; g += nwaves) {
        const uint32_t item = g * 64 + lane;
        if (item >= n_total) continue;
        const uint32_t c = MODE == SW_FP ? io.fpq[item] : MODE == SW_TRACE ? (io.tq_in ? io.tq_in[item] : io.tq[item]) : io.dq ? io.dq[item] : item;
        const uint32_t lq = io.c_qlen[c], w = io.c_wlen[c];
        const int64_t wo = io.c_woff[c];
        if (wo < 0 || (uint64_t)wo + w > io.rbuf_len || w == 0 || lq == 0) {
            io.c_status[c] = CS_WINDOW_OOR; io.c_ciglen[c] = 0;
            continue;
        }
        if (w + 1 > maxW1 || (lq + SW_R - 1) / SW_R > max_strips) {
            // wider than the slabs were sized for: never in a run sized by the batch's own maxima; the small-batch path sizes by
            // fixed capacities and redoes such a batch through the general path (err_flags bit 16)
            io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0;
            atomicOr(&io.ctr->err_flags, 16ull);
            continue;
        }
        const uint8_t *q = io.qbuf + io.c_qoff[c];
        const uint8_t *rw = io.rbuf + wo;
        const uint32_t ns = (lq + SW_R - 1) / SW_R;
        bool needfp = false;

        for (uint32_t j = 0; j <= w; j++) {                 // row 0: {0,undef}
            if (LDSB) bl[j * 64] = DIR_UNDEF; else bnd[j * 64] = DIR_UNDEF;
            if (FP) bndv[j * 64] = 0.0;
        }
        // leftmost maximum of the last row; starts at column 0: V(Lq,0) = -(Lq+1)
        int bestp = -(int)(lq + 1) * 4 + DIR_UP;
        double bestv = -(double)(lq + 1);
        uint32_t bestj = 0;

        for (uint32_t s = 0; s < ns && !needfp; s++) {
            const uint32_t i0 = s * SW_R;
            uint32_t qc[SW_R];
            int H[SW_R];        // int path: packed 4V+code of column j-1 ; FP path: code only
            double Hv[SW_R];
#pragma unroll
            for (int t = 0; t < SW_R; t++) {
                const uint32_t i = i0 + t;
                qc[t] = i < lq ? q[i] : 0xffu;
                if (!FP && i < lq && !is_acgt(qc[t])) needfp = true;
                // column 0: V(i,0) = -(i+1) for row i = i0+t+1, dir up (sw.erl:73-77)
                if (FP) { Hv[t] = -(double)(i0 + t + 2); H[t] = DIR_UP; }
                else H[t] = -(int)(i0 + t + 2) * 4 + DIR_UP;
            }
            if (needfp) break;
            // (i0, 0): row 0 is {0,undef}; otherwise column 0 of row i0
            int dcar = s == 0 ? DIR_UNDEF : (FP ? DIR_UP : -(int)(i0 + 1) * 4 + DIR_UP);
            double dcarv = s == 0 ? 0.0 : -(double)(i0 + 1);
            const int tl = (int)(lq - 1 - i0);          // register slot of the last query row (last strip only)
            const bool last = (s == ns - 1);
            // Columns in chunks of four, one chunk ahead: the window bytes and the boundary values (row i0 of the strip above) of
            // columns jc + 4 .. jc + 7 are asked for before the cells of columns jc .. jc + 3 are computed, so that a lane meets
            // one memory latency per chunk and most of it behind those cells.  (A lane per alignment leaves the chip a wave or
            // two per SIMD when this kernel gets the few hundred-base-wide windows of a batch: a dependent byte load and a
            // dependent boundary load per COLUMN were most of its time, 5.7 ms for 56 k windows on the repeat-rich batch.)
            // A column's boundary value is read a chunk before the column is written: nothing of this strip is read back.
            auto fetch = [&](uint32_t jc, uint32_t (&rcn)[4], int (&un)[4], double (&uvn)[4]) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t col = jc + (uint32_t)k;
                    const bool in = col <= w;
                    rcn[k] = in ? (uint32_t)rw[col - 1] : 0u;
                    if (LDSB) un[k] = in ? (int)(int16_t)bl[col * 64] : 0; else un[k] = in ? (int)bnd[col * 64] : 0;
                    uvn[k] = 0.0;
                    if (FP) uvn[k] = in ? bndv[col * 64] : 0.0;
                }
            };
            uint32_t rcA[4]; int uA[4]; double uvA[4];
            fetch(1u, rcA, uA, uvA);
            for (uint32_t jc = 1; jc <= w && !needfp; jc += 4) {
                uint32_t rcB[4]; int uB[4]; double uvB[4];
                fetch(jc + 4, rcB, uB, uvB);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t j = jc + (uint32_t)k;
                    if (j > w || needfp) continue;
                    const uint32_t rc = rcA[k];
                    if (!FP && s == 0 && !is_acgt(rc)) { needfp = true; continue; }
                    int d = dcar, u = uA[k];
                    double dv = dcarv, uv = uvA[k];
                    dcar = u; dcarv = uv;
                    uint32_t dirw = 0;
#pragma unroll
                    for (int t = 0; t < SW_R; t++) {
                        if (FP) {
                            const int lcode = H[t]; const double lv = Hv[t];
                            const double cd = dv + sigma_gen(qc[t], rc);
                            const double cl = lv + (lcode == DIR_LEFT ? -1.0 : -2.0);
                            const double cu = uv + (u == DIR_UP ? -1.0 : -2.0);
                            double hv = cd; int hc = DIR_UL;
                            if (cu > hv) { hv = cu; hc = DIR_UP; }
                            if (cl > hv) { hv = cl; hc = DIR_LEFT; }
                            dirw |= (uint32_t)hc << (2 * t);
                            d = lcode; dv = lv; u = hc; uv = hv; H[t] = hc; Hv[t] = hv;
                        } else {
                            const int old = H[t];
                            const int sig = (qc[t] == rc) ? 7 : -5;                       // 4*sigma - 1
                            const int cd = (d | 3) + sig;                                 // 4(Vul+sigma) + up_left
                            const int cl = old + (int)(int8_t)(0xF5F6F7FCu >> ((old & 3) * 8));   // {-4,-9,-10,-11}
                            const int cu = u + (int)(int8_t)(0xF6F7FCF9u >> ((u & 3) * 8));       // {-7,-4,-9,-10}
                            int h = cd > cl ? cd : cl;
                            h = h > cu ? h : cu;
                            if (DIRS) dirw |= (uint32_t)(h & 3) << (2 * t);
                            d = old; u = h; H[t] = h;
                        }
                    }
                    if (LDSB) bl[j * 64] = (uint16_t)H[SW_R - 1]; else bnd[j * 64] = (uint32_t)H[SW_R - 1];
                    if (FP) bndv[j * 64] = Hv[SW_R - 1];
                    if (DIRS) dirs[((uint64_t)s * maxW1 + j) * 64] = dirw;
                    if (last) {       // value of query row Lq in this column (register slot tl of the last strip)
                        int lastp = H[0]; double lastv = Hv[0];
#pragma unroll
                        for (int t = 1; t < SW_R; t++) { if (t == tl) { lastp = H[t]; lastv = Hv[t]; } }
                        if (FP) { if (lastv > bestv) { bestv = lastv; bestj = j; } }
                        else if ((lastp >> 2) > (bestp >> 2)) { bestp = lastp; bestj = j; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) { rcA[k] = rcB[k]; uA[k] = uB[k]; uvA[k] = uvB[k]; }
            }
        }
        if (!FP && needfp) {          // a base outside ACGT: sigma may return 1.6 -> the fp64 kernel (runs last)
            io.c_status[c] = CS_NEED_FP;
            io.fpq[atomicAdd(&io.ctr->fp_count, 1u)] = c;
            continue;
        }
        // sw.erl:69 after the last row: no_match iff RowMax < -2*(Lq div 2) + 2*Lq
        const int thr = -2 * (int)(lq / 2) + 2 * (int)lq;
        int score;
        if (FP) { if (bestv < (double)thr) { io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0; continue; } score = (int)bestv; }
        else { score = bestp >> 2; if (score < thr) { io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0; continue; } }

        uint8_t *slot = io.cig + (uint64_t)c * io.cig_stride;
        int p = (int)io.cig_stride, len = 0;
        if (MODE == SW_SCORE) {
            // does the pure diagonal ending at (Lq, bestj) reach the score?  then the CIGAR is "<Lq>M"
            bool diag = bestj >= lq;
            if (diag) {
                int sum = 0;
                const uint8_t *rd = rw + (bestj - lq);
                for (uint32_t i = 0; i < lq; i++) sum += (q[i] == rd[i]) ? 2 : -1;
                diag = sum == score;
            }
            if (!diag) {
                io.c_status[c] = CS_PENDING;
                io.tq[atomicAdd(&io.ctr->tq_count, 1u)] = c;
                continue;
            }
            cig_emit(slot, p, len, lq, 'M');
        } else {
            // traceback (sw.erl:116-160): from the leftmost maximum of the last row up to row 0
            uint32_t i = lq, j = bestj;
            int runop = -1; uint32_t runlen = 0;
            for (;;) {
                int code = -1;
                if (i > 0) {
                    if (j == 0) code = DIR_UP;
                    else code = (int)((dirs[((uint64_t)((i - 1) / SW_R) * maxW1 + j) * 64] >> (2 * ((i - 1) % SW_R))) & 3u);
                }
                if (code == runop) { runlen++; }
                else {
                    if (runop >= 0) cig_emit(slot, p, len, runlen, runop == DIR_UL ? 'M' : runop == DIR_UP ? 'I' : 'D');
                    runop = code; runlen = 1;
                }
                if (code < 0) break;
                if (code == DIR_UL) { i--; j--; } else if (code == DIR_UP) i--; else j--;
            }
        }
        io.c_score[c] = score;
        io.c_ciglen[c] = (uint32_t)len;
        io.c_status[c] = len > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
    }
}
template <int MODE, bool LDSB>
__global__ __launch_bounds__(256) void k_sw(SwIO io, uint32_t n_items, uint32_t maxW1, uint32_t max_strips,
                                            uint32_t *__restrict__ slab_bnd, double *__restrict__ slab_bndv,
                                            uint32_t *__restrict__ slab_dir) {
    sw_lanes<MODE, LDSB>(io, n_items, maxW1, max_strips, slab_bnd, slab_bndv, slab_dir,
                         (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), (gridDim.x * blockDim.x) >> 6);
}
// blockIdx.y = 0: the traced integer instance on the trace queue (io.tq_in when set), 1: the fp64 instance on the fp queue
__global__ __launch_bounds__(256) void k_sw_tail(SwIO io, uint32_t n_items, uint32_t maxW1, uint32_t max_strips,
                                                 uint32_t *__restrict__ bnd_t, uint32_t *__restrict__ dir_t,
                                                 uint32_t *__restrict__ bnd_fp, double *__restrict__ bndv_fp, uint32_t *__restrict__ dir_fp) {
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (gridDim.x * blockDim.x) >> 6;
    if (blockIdx.y == 0) sw_lanes<SW_TRACE, false>(io, n_items, maxW1, max_strips, bnd_t, nullptr, dir_t, wave, nwaves);
    else sw_lanes<SW_FP, false>(io, n_items, maxW1, max_strips, bnd_fp, bndv_fp, dir_fp, wave, nwaves);
}

// ------------------------------------------------------------------------------------------------
// k_sw_prefilter: QUAD per candidate, on the bit-planes of reads and Ref1 only.
//   * a base outside ACGT in the read or the window  -> fp64 queue (sigma may return 1.6);
//   * the read matches the window exactly on some diagonal -> done: 2*Lq is the largest score any
//     alignment of Lq bases can have (sigma <= 2, gaps cost), only a gap-free all-match diagonal reaches
//     it, so the leftmost such diagonal IS the leftmost maximum of the last row and its cells all chose
//     up_left: {2*Lq, "<Lq>M"} without running the DP;
//   * square window (W = Lq, i.e. D = 0) and at most one mismatch on its only full diagonal -> done as
//     well: every other path into the last row contains an insertion (loss >= 4 > 3), so the diagonal is
//     the unique optimum, every cell on it ties or wins as up_left, and no column left of Lq reaches it;
//     two mismatches -> done unless one of the two single-insertion detours (see the code) exists;
//   * everything else -> DP queue (dense, so the DP kernels' lanes stay full).
// Lane q of the quad takes the 32-base chunks q, q+4, ...: its plane words arrive with one 16-byte load (all four
// planes), the word after it from the next lane by DPP (lane 3 loads it), so a candidate costs ~6 L2 requests instead
// of 16; mismatch counts and positions meet in quad reductions.
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    return v;
}
__device__ __forceinline__ uint32_t quad_min(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = v < o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); return v < o ? v : o;
}
__device__ __forceinline__ uint32_t quad_max(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); return v > o ? v : o;
}
// all four planes of the 32 bases from bit position pos + 32 * ql (ql = lane in quad): word ql of the run that starts at
// pos's word, completed by the next lane's word
__device__ __forceinline__ uint4 quad_planes4(const uint32_t *__restrict__ planes, uint64_t pos, int ql, uint32_t &lo1, uint32_t &hi1) {
    const uint64_t w = (pos >> 5) + (uint64_t)ql;
    const uint32_t sh = (uint32_t)pos & 31u;
    const uint4 a = ((const uint4 *)planes)[w];
    uint4 x = make_uint4(0, 0, 0, 0);
    if (ql == 3) x = ((const uint4 *)planes)[w + 1];
    uint4 b;                                                   // the next lane's word (quad_perm [1,2,3,3]); lane 3: its extra load
    b.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.x, 0xF9, 0xF, 0xF, true); b.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.y, 0xF9, 0xF, 0xF, true);
    b.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.z, 0xF9, 0xF, 0xF, true); b.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.w, 0xF9, 0xF, 0xF, true);
    if (ql == 3) b = x;
    uint4 r;
    r.x = __builtin_amdgcn_alignbit(b.x, a.x, sh); r.y = __builtin_amdgcn_alignbit(b.y, a.y, sh);
    r.z = __builtin_amdgcn_alignbit(b.z, a.z, sh); r.w = __builtin_amdgcn_alignbit(b.w, a.w, sh);
    // the code planes one base further on (the same two words: no second load for the neighbouring diagonal)
    lo1 = sh == 31 ? b.x : __builtin_amdgcn_alignbit(b.x, a.x, sh + 1);
    hi1 = sh == 31 ? b.y : __builtin_amdgcn_alignbit(b.y, a.y, sh + 1);
    return r;
}
__device__ __forceinline__ uint4 quad_planes4(const uint32_t *__restrict__ planes, uint64_t pos, int ql) {
    uint32_t lo1, hi1;
    return quad_planes4(planes, pos, ql, lo1, hi1);
}
// FROM_READS: the candidates are made on the way (k_candidates folded in): a quad takes a READ, walks its clusters
// (src: the cluster stage's per-read results), does the window arithmetic of worker_loop (src/worker_bwt.erl:86-90), writes
// the candidate arrays the later kernels read and goes on with the candidate -- one launch and one round trip of the
// candidate arrays less.  n_items = reads then; the candidate index is clus_off[r] + k.
struct CandSrc {
    const uint64_t *off; const uint32_t *r_K; const uint64_t *seed_off, *clus_off; const uint32_t *pos, *r_qty, *r_cl;
    uint32_t *c_S, *c_D; uint64_t *c_qoff; uint32_t *c_qlen; int64_t *c_woff; uint32_t *c_wlen;
    // FROM_READS: the candidates past the first PF_KCAP of a read are only described (window arithmetic) and listed here;
    // !FROM_READS with list != null: the pass over that list, quad per candidate
    uint32_t *ovq; unsigned int *ov_count;
    const uint32_t *list; const unsigned int *list_count;
    uint32_t ulen;                               // > 0: every read has this length (no offset loads)
    uint32_t kcap;                               // FROM_READS: 1 + the candidates of a read the fused pass tests itself (0 = PF_KCAP)
};
// A wave of the fused pass takes 16 reads and loops to the LARGEST cluster count among them: on a repeat-rich genome one read in
// twelve has a dozen or more clusters, three waves in four hold such a read, and the other fifteen quads idle through its loop
// (8.8 ms for 29 M candidates where 12 M take 1.0 ms).  So the fused pass tests at most PF_KCAP candidates per read; the others
// get their descriptors written (four per quad step, stores only) and go into a list that a second launch works through with
// a quad per candidate.  (Two per read until the queues took their space 256 entries per atomic: while the list's one atomic per 64
// entries bound the kernel, a longer list cost more than the idle quads of a second turn -- 5.76 against 5.40 ms; since then ONE per
// read is better, 2.91 against 3.46 ms on the repeat-rich batch, the same on batches of one cluster per read.)
#define PF_KCAP 1u
// (Held to more waves per SIMD -- 6: 80 VGPRs, 8: 64 VGPRs, a few dwords of scratch each -- the kernel is slower: 1.12 / 1.55 ms
// against 1.01 at the compiler's own 91 VGPRs and 5 waves, although it waits on dependent loads ~80 % of its time.)
template <bool FROM_READS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8)))
void k_sw_prefilter(SwIO io, uint32_t n_items, CandSrc src) {
    constexpr int FL = 256;
    __shared__ uint32_t wq_[4][4][FL + 64];
    WaveQ qfp, qdp, qov, qlo;
    qfp.buf = wq_[threadIdx.x >> 6][0]; qfp.have = 0; qdp.buf = wq_[threadIdx.x >> 6][1]; qdp.have = 0;
    qov.buf = wq_[threadIdx.x >> 6][2]; qov.have = 0; qlo.buf = wq_[threadIdx.x >> 6][3]; qlo.have = 0;
    if (!FROM_READS && src.list) n_items = *src.list_count;
    const uint32_t *qp = io.qplanes, *wp = io.wplanes;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t stride = (gridDim.x * blockDim.x) >> 2;                  // candidates (reads) per sweep
    const uint32_t first = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >> 2;
    unsigned long long cells = 0, wbytes = 0;                               // FROM_READS: what k_candidates totals
    uint32_t maxw = 0, maxlq = 0;
    // FROM_READS: everything a read's first candidate needs is asked for at once -- its cluster count, offsets, flags and the first
    // staged cluster of its r_cl row, whether or not that row turns out to be in use -- and ONE PASS AHEAD: the pass that tests
    // read i's windows starts by asking for read i + stride's words, so that they travel together with this pass's window planes
    // and a pass is one round trip to memory, not two dependent ones (the kernel is bound by the latency of its loads)
    // (the read offsets are not carried ahead: batches of one read length compute them, the others load them in the pass itself)
    struct Pre { uint32_t K, rq; uint64_t co; uint2 cl0; };
    auto prefetch = [&](uint32_t it) {
        Pre p;
        const bool inr = it < n_items;
        p.K = inr ? src.r_K[it] : 0u;
        p.co = inr ? src.clus_off[it] : 0ull;
        p.rq = inr ? src.r_qty[it] : 0u;
        p.cl0 = make_uint2(0u, 0u);
        if (inr && src.r_cl) p.cl0 = *(const uint2 *)(src.r_cl + (uint64_t)it * (2 * CL_STAGE));
        return p;
    };
    Pre nx{};
    if (FROM_READS) nx = prefetch(first + ((threadIdx.x & 63) >> 2));
    for (uint32_t c0 = first; c0 < n_items; c0 += stride) {                 // whole waves iterate together
        const uint32_t item = c0 + ((threadIdx.x & 63) >> 2);
        uint32_t K_r = 1, kmax = 1;
        uint64_t co_r = 0, qo_r = 0;
        uint32_t lq_r = 0;
        const uint32_t *cl_r = nullptr;
        uint2 cl0 = make_uint2(0u, 0u);
        bool cl0_ok = false;
        if (FROM_READS) {
            const Pre cur = nx;
            nx = prefetch(item + stride);
            K_r = cur.K;
            uint64_t o0, o1;
            if (src.ulen) { o0 = (uint64_t)item * src.ulen; o1 = o0 + src.ulen; }
            else { const bool inr = item < n_items; o0 = inr ? src.off[item] : 0ull; o1 = inr ? src.off[item + 1] : 0ull; }
            const uint64_t co = cur.co;
            const uint32_t rq = cur.rq;
            cl0 = cur.cl0;
            kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(K_r));
            if (K_r) {
                qo_r = o0; lq_r = (uint32_t)(o1 - o0);
                co_r = co;
                // clusters staged by the seed kernels (RQ_DONE) sit in the read's r_cl row, the cluster kernels' in its slice of pos
                cl0_ok = (rq & RQ_DONE) != 0 && src.r_cl != nullptr;
                cl_r = (rq & RQ_DONE) ? src.r_cl + (uint64_t)item * (2 * CL_STAGE) : src.pos + src.seed_off[item];
                maxlq = lq_r > maxlq ? lq_r : maxlq;
            }
        }
        const uint32_t kcap = src.kcap ? src.kcap - 1u : PF_KCAP;
        if (FROM_READS && kmax > kcap) {
            // the candidates past the first PF_KCAP of every read of this pass: descriptors and list entries, four per quad step
            for (uint32_t kb = kcap; kb < kmax; kb += 4) {
                const uint32_t kc = kb + (uint32_t)ql;
                const bool have = kc < K_r;
                const uint32_t c = (uint32_t)(co_r + kc);
                if (have) {
                    const uint32_t S = cl_r[2 * kc], D = cl_r[2 * kc + 1], w = lq_r + D;
                    src.c_S[c] = S; src.c_D[c] = D; src.c_qoff[c] = qo_r; src.c_qlen[c] = lq_r; src.c_wlen[c] = w;
                    src.c_woff[c] = (int64_t)S - (int64_t)w + BWT_REF_EXT;
                    cells += (unsigned long long)lq_r * w; wbytes += w;
                    maxw = w > maxw ? w : maxw;
                }
                waveq_push<FL>(qov, have, c, src.ovq, src.ov_count);
            }
            kmax = kcap;
            K_r = K_r < kcap ? K_r : kcap;
        }
        for (uint32_t kc = 0; kc < kmax; kc++) {                              // (one pass unless FROM_READS)
            const uint32_t c = FROM_READS ? (uint32_t)(co_r + kc) : src.list ? (item < n_items ? src.list[item] : 0u) : item;
            int route = 0;                 // 0 nothing (idle / out of range / settled), 1 fp64 queue, 2 DP queue
            // quad-uniform from here on: the four lanes of a quad hold the same candidate
            const bool have = FROM_READS ? kc < K_r : item < n_items;
            uint32_t lq, w;
            int64_t wo;
            uint64_t qo;
            if (FROM_READS) {
                const bool spec = kc == 0 && cl0_ok;
                const uint32_t S = !have ? 0u : spec ? cl0.x : cl_r[2 * kc], D = !have ? 0u : spec ? cl0.y : cl_r[2 * kc + 1];
                lq = have ? lq_r : 0u; w = have ? lq_r + D : 0u; qo = qo_r;
                wo = (int64_t)S - (int64_t)w + BWT_REF_EXT;
                if (have && ql == 0) {
                    // (what the hit tables need of every candidate; read offset / length and window offset only matter to the DP
                    // kernels and are written below for the candidates that are queued for one: 20 bytes less per settled candidate)
                    src.c_S[c] = S; src.c_D[c] = D; src.c_wlen[c] = w;
                    cells += (unsigned long long)lq * w; wbytes += w;
                    maxw = w > maxw ? w : maxw;
                }
            } else {
                lq = have ? io.c_qlen[c] : 0u; w = have ? io.c_wlen[c] : 0u;
                wo = have ? io.c_woff[c] : 0;
                qo = have ? io.c_qoff[c] : 0;
            }
            const bool oor = wo < 0 || (uint64_t)wo + w > io.rbuf_len || w == 0 || lq == 0;
            const bool live = have && !oor;
            if (have && oor && ql == 0) { io.c_status[c] = CS_WINDOW_OOR; io.c_ciglen[c] = 0; }
            const bool square = w == lq;
            // one pass over read and window: bases outside ACGT anywhere, and (square window) the mismatches of the diagonal
            // (square window) also the first mismatch f1 of diagonal +1 (q[k] against r[k+1]) and the last one, h, of diagonal -1
            // (q[k+1] against r[k]): the two-mismatch case below needs them, and they come out of the words loaded here
            uint32_t bad = 0, mism = 0, p1 = 0xffffffffu, p2 = 0, f1 = 0xffffffffu, h = 0;
            const uint32_t lqmax = __builtin_amdgcn_readfirstlane((int)wave_max_u32(live ? lq : 0u));
            for (uint32_t i0 = 0; i0 < lqmax; i0 += 128) {                      // wave-uniform trip count, DPP inside
                const uint32_t i = i0 + 32u * (uint32_t)ql;
                const bool in = live && i0 < lq;                               // (a shorter read of the wave: stay inside its planes)
                uint32_t qlo1, qhi1, wlo1, whi1;
                const uint4 q4 = quad_planes4(qp, in ? qo + i0 : 0, ql, qlo1, qhi1);
                const uint4 w4 = quad_planes4(wp, in && square ? (uint64_t)wo + i0 : 0, ql, wlo1, whi1);
                if (live && i < lq) {
                    const uint32_t mk = lq - i < 32 ? (1u << (lq - i)) - 1u : 0xffffffffu;
                    bad |= (q4.z | q4.w) & mk;
                    if (square) {
                        bad |= (w4.z | w4.w) & mk;
                        const uint32_t e = ((q4.x ^ w4.x) | (q4.y ^ w4.y)) & mk;
                        if (e) {
                            const uint32_t f = i + (uint32_t)__ffs((int)e) - 1u, l = i + 31u - (uint32_t)__clz((int)e);
                            p1 = f < p1 ? f : p1; p2 = l > p2 ? l : p2;
                            mism += __popc(e);
                        }
                        if (i + 1 < lq) {                                       // the Lq-1 pairs of the neighbouring diagonals
                            const uint32_t mk1 = lq - 1 - i < 32 ? (1u << (lq - 1 - i)) - 1u : 0xffffffffu;
                            const uint32_t eu = ((q4.x ^ wlo1) | (q4.y ^ whi1)) & mk1, ed = ((qlo1 ^ w4.x) | (qhi1 ^ w4.y)) & mk1;
                            if (eu) { const uint32_t f = i + (uint32_t)__ffs((int)eu) - 1u; f1 = f < f1 ? f : f1; }
                            if (ed) { const uint32_t l = i + 32u - (uint32_t)__clz((int)ed); h = l > h ? l : h; }   // k = bit + 1
                        }
                    }
                }
            }
            const uint32_t wmax = __builtin_amdgcn_readfirstlane((int)wave_max_u32(live && !square ? w : 0u));
            for (uint32_t i0 = 0; i0 < wmax; i0 += 128) {
                const uint32_t i = i0 + 32u * (uint32_t)ql;
                const uint4 w4 = quad_planes4(wp, live && !square && i0 < w ? (uint64_t)wo + i0 : 0, ql);
                if (live && !square && i < w) {
                    const uint32_t mk = w - i < 32 ? (1u << (w - i)) - 1u : 0xffffffffu;
                    bad |= (w4.z | w4.w) & mk;
                }
            }
            bad = quad_or(bad); mism = quad_sum(mism); p1 = quad_min(p1); p2 = quad_max(p2);
            bool done = false;
            int score = 2 * (int)lq;
            const bool clean = live && !bad;
            if (live && bad) { if (ql == 0) io.c_status[c] = CS_NEED_FP; route = 1; }
            // square window: one full diagonal with m mismatches at p1 <= p2, score 2*Lq - 3m.  Every other path
            // into a diagonal cell or into the last row holds u >= 1 insertions, and a run of g gap steps costs
            // g + 1: u >= 2 loses >= 7, u = 1 loses >= 4.  m <= 1 (loss <= 3): the diagonal is the unique optimum.
            // m = 2 (loss 6): only two kinds of path can reach it -- one insertion and otherwise all matches:
            //   (a) into a diagonal cell past p2: start one column in, ride diagonal +1 (q[k] = r[k+1]) for every
            //       k < t, insert q[t], t >= p2: exists iff the first f1 = #leading matches on diagonal +1 >= p2;
            //   (b) into the last row at column Lq-1 (a tie there would be the LEFTMOST maximum): main diagonal
            //       for k < t <= p1, insert q[t], then diagonal -1 (q[k] = r[k-1]) for every k > t: exists iff
            //       the highest mismatch on diagonal -1 sits at h <= p1.
            // (one insertion + one deletion only ties a diagonal cell, where up_left wins the tie.)  Neither exists:
            // every diagonal cell keeps its prefix score as up_left and the last row peaks at column Lq alone.
            if (clean && square) { done = mism <= 1 && lq >= 8; score -= 3 * (int)mism; }
            if (clean && square && mism == 2 && lq >= 8) {
                f1 = quad_min(f1); h = quad_max(h);
                if (f1 == 0xffffffffu) f1 = lq - 1;
                done = !(f1 >= p2) && !(h <= p1);
            }
            const bool slide = clean && !square && w > lq && w - lq <= 64;
            if (__ballot(slide)) {
                // exact match on some diagonal d = 0 .. W - Lq: lane q tries d = q, q+4, ... on its own (most diagonals fail in
                // their first 32 bases); any hit settles the candidate -- {2Lq, "<Lq>M"} does not depend on which
                uint32_t hit = 0;
                if (slide)
                    for (uint32_t d = (uint32_t)ql; d + lq <= w && !hit; d += 4) {
                        bool all = true;
                        for (uint32_t i = 0; i < lq && all; i += 32) {
                            const uint4 q4 = planes4(qp, qo + i), w4 = planes4(wp, (uint64_t)wo + d + i);
                            uint32_t e = (q4.x ^ w4.x) | (q4.y ^ w4.y);
                            if (lq - i < 32) e &= (1u << (lq - i)) - 1u;
                            all = e == 0;
                        }
                        if (all) hit = 1;
                    }
                hit = quad_or(hit);
                if (slide) done = hit != 0;
            }
            if (clean) {
                if (!done) { if (ql == 0) io.c_status[c] = CS_PENDING; route = 2; }
                else if (ql == 0) {
                    // "<Lq>M": only its length is recorded (CIG_DIAG: k_hits_write prints it from the window length, which equals
                    // Lq here) -- no scattered write into the candidate's text slot
                    uint32_t len = 2;
                    for (uint32_t v = lq; v >= 10; v /= 10) len++;
                    io.c_score[c] = score;
                    io.c_ciglen[c] = len | (square ? CIG_DIAG : 0u);
                    if (!square) {                                   // (exact match on a shifted diagonal: the window is wider than Lq)
                        uint8_t *slot = io.cig + (uint64_t)c * io.cig_stride;
                        int p = (int)io.cig_stride, l2 = 0;
                        cig_emit(slot, p, l2, lq, 'M');
                    }
                    io.c_status[c] = len > io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
                }
            }
            if (FROM_READS && route != 0 && ql == 0) { src.c_qoff[c] = qo; src.c_qlen[c] = lq; src.c_woff[c] = wo; }
            // the two queues, staged in LDS (one atomic per 64 entries)
            waveq_push<FL>(qfp, route == 1 && ql == 0, c, io.fpq, &io.ctr->fp_count);
            // (a square window whose diagonal has at most six mismatches scores >= 2 Lq - 18 on it: what the 12-diagonal band's
            // certificate covers -- 0.6 of the first tier's cells)
            const bool lo_q = route == 2 && io.dq_lo != nullptr && square && mism <= io.lo_max_m && lq >= 24;
            waveq_push<FL>(qdp, route == 2 && !lo_q && ql == 0, c, io.dq, &io.ctr->dq_count);
            if (io.dq_lo) waveq_push<FL>(qlo, lo_q && ql == 0, c, io.dq_lo, &io.ctr->dq_lo_count);
        }
        }
    waveq_drain<FL>(qfp, io.fpq, &io.ctr->fp_count);
    waveq_drain<FL>(qdp, io.dq, &io.ctr->dq_count);
    if (io.dq_lo) waveq_drain<FL>(qlo, io.dq_lo, &io.ctr->dq_lo_count);
    if (FROM_READS) {
        waveq_drain<FL>(qov, src.ovq, src.ov_count);
        const unsigned long long sums[2] = {wave_sum_u64(cells), wave_sum_u64(wbytes)};
        unsigned long long *const g64[2] = {&io.ctr->sw_cells, &io.ctr->window_bytes};
        unsigned int *const g32[2] = {nullptr, nullptr};
        const uint32_t mx[2] = {wave_max_u32(maxw), wave_max_u32(maxlq)};
        unsigned int *const gmx[2] = {&io.ctr->max_w, &io.ctr->max_lq};
        block_counters(sums, g64, g32, mx, gmx);
    }
}

// ------------------------------------------------------------------------------------------------
// k_sw_score: the fast path of the extend stage (integer, read and window all ACGT, W <= the LDS cap).
// Same recurrence as k_sw<SW_SCORE>, but nothing is read from HBM inside the DP:
//   * prologue, per group of 64 candidates: the wave loads each candidate's window and query with
//     COALESCED 64-byte reads, turns them into two bit-planes per sequence with __ballot (bit b of the
//     low/high plane = low/high bit of base b's 2-bit code) and parks the planes in LDS [word][lane];
//   * strip: 16 query rows = 16 bits of each query plane; column: one bit of each window plane; the
//     16 row-matches of a column are one xor/or/not (eq mask), each cell takes its bit;
//   * the row between strips lives in LDS as 16-bit packed cells [j][lane].
// A candidate that holds a base outside ACGT is handed to the fp64 kernel, one whose alignment is
// not the plain diagonal (or whose window is wider than the cap) to the trace kernel.
__global__ __launch_bounds__(256) void k_sw_score(SwIO io, uint32_t n_items, uint32_t wcap, uint32_t lqcap) {
    extern __shared__ uint32_t lds_sw[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t PW = (wcap + 31) / 32, QW = (lqcap + 31) / 32;          // plane words
    const uint32_t words_per_wave = (2 * PW + 2 * QW) * 64 + ((wcap + 1) * 64 + 1) / 2;
    uint32_t *pl = lds_sw + (uint64_t)(threadIdx.x >> 6) * words_per_wave;  // planes [word*64 + lane]
    uint32_t *const wlo = pl, *const whi = pl + PW * 64, *const qlo = pl + 2 * PW * 64, *const qhi = pl + (2 * PW + QW) * 64;
    uint16_t *bl = (uint16_t *)(pl + (2 * PW + 2 * QW) * 64) + lane;        // boundary row [j*64]

    if (io.dq) { n_items = io.ctr->dq_count; if (n_items <= io.wave_upto) return; }
    for (uint32_t g = wave; (uint64_t)g * 64 < n_items; g += nwaves) {
        const uint32_t item = g * 64 + lane;
        const uint32_t c = item < n_items ? (io.dq ? io.dq[item] : item) : 0;
        uint32_t lq = 0, w = 0;
        int64_t wo = 0;
        uint64_t qo = 0;
        int state = 0;       // 0 = run the DP here, 1 = idle lane, 2 = out of range, 3 = to the trace kernel (too wide)
        if (item < n_items) {
            lq = io.c_qlen[c]; w = io.c_wlen[c]; wo = io.c_woff[c]; qo = io.c_qoff[c];
            if (wo < 0 || (uint64_t)wo + w > io.rbuf_len || w == 0 || lq == 0) state = 2;
            else if (w > wcap || lq > lqcap) state = 3;
        } else state = 1;
        // ---- prologue: bit-planes of every candidate's window and query into LDS
        unsigned long long badmask = 0;
        for (int k = 0; k < 64; k++) {
            const uint32_t kw = __shfl(state == 0 ? w : 0u, k, 64);
            if (kw == 0) continue;
            const uint32_t klq = __shfl(lq, k, 64);
            const uint64_t kro = (uint64_t)__shfl((long long)wo, k, 64), kqo = (uint64_t)__shfl((long long)qo, k, 64);
            for (uint32_t c0 = 0; c0 < kw; c0 += 64) {
                const uint32_t idx = c0 + lane;
                const int code = idx < kw ? base_code(io.rbuf[kro + idx]) : 0;
                const unsigned long long lo = __ballot(code & 1), hi = __ballot(code & 2), bad = __ballot(code < 0);
                if (bad) badmask |= 1ull << k;
                if (lane == 0) {
                    const uint32_t wd = 2 * (c0 >> 6);
                    wlo[wd * 64 + k] = (uint32_t)lo; whi[wd * 64 + k] = (uint32_t)hi;
                    if (wd + 1 < PW) { wlo[(wd + 1) * 64 + k] = (uint32_t)(lo >> 32); whi[(wd + 1) * 64 + k] = (uint32_t)(hi >> 32); }
                }
            }
            for (uint32_t c0 = 0; c0 < klq; c0 += 64) {
                const uint32_t idx = c0 + lane;
                const int code = idx < klq ? base_code(io.qbuf[kqo + idx]) : 0;
                const unsigned long long lo = __ballot(code & 1), hi = __ballot(code & 2), bad = __ballot(code < 0);
                if (bad) badmask |= 1ull << k;
                if (lane == 0) {
                    const uint32_t wd = 2 * (c0 >> 6);
                    qlo[wd * 64 + k] = (uint32_t)lo; qhi[wd * 64 + k] = (uint32_t)hi;
                    if (wd + 1 < QW) { qlo[(wd + 1) * 64 + k] = (uint32_t)(lo >> 32); qhi[(wd + 1) * 64 + k] = (uint32_t)(hi >> 32); }
                }
            }
        }
        if (state == 1) continue;
        if (state == 2) { io.c_status[c] = CS_WINDOW_OOR; io.c_ciglen[c] = 0; continue; }
        if (state == 3) { io.c_status[c] = CS_PENDING; io.tq[atomicAdd(&io.ctr->tq_count, 1u)] = c; continue; }
        if ((badmask >> lane) & 1ull) {
            io.c_status[c] = CS_NEED_FP;
            io.fpq[atomicAdd(&io.ctr->fp_count, 1u)] = c;
            continue;
        }
        // ---- DP
        const uint32_t ns = (lq + SW_R - 1) / SW_R;
        for (uint32_t j = 0; j <= w; j++) bl[j * 64] = DIR_UNDEF;             // row 0: {0,undef}
        int bestp = -(int)(lq + 1) * 4 + DIR_UP;                                // V(Lq,0) = -(Lq+1)
        uint32_t bestj = 0;
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t i0 = s * SW_R;
            const uint32_t q_lo = (qlo[(i0 >> 5) * 64 + lane] >> (i0 & 31)) & 0xffffu;
            const uint32_t q_hi = (qhi[(i0 >> 5) * 64 + lane] >> (i0 & 31)) & 0xffffu;
            int H[SW_R];
#pragma unroll
            for (int t = 0; t < SW_R; t++) H[t] = -(int)(i0 + t + 2) * 4 + DIR_UP;   // column 0 (sw.erl:73-77)
            int dcar = s == 0 ? DIR_UNDEF : -(int)(i0 + 1) * 4 + DIR_UP;
            const int tl = (int)(lq - 1 - i0);
            const bool last = (s == ns - 1);
            uint32_t r_lo = 0, r_hi = 0;
            for (uint32_t j = 1; j <= w; j++) {
                const uint32_t b = j - 1;
                if ((b & 31) == 0) { r_lo = wlo[(b >> 5) * 64 + lane]; r_hi = whi[(b >> 5) * 64 + lane]; }
                const uint32_t rl = 0u - ((r_lo >> (b & 31)) & 1u), rh = 0u - ((r_hi >> (b & 31)) & 1u);
                const uint32_t eq = ~((q_lo ^ rl) | (q_hi ^ rh));                    // bit t: row t matches this column
                int d = dcar, u = (int)(int16_t)bl[j * 64];
                dcar = u;
#pragma unroll
                for (int t = 0; t < SW_R; t++) {
                    const int old = H[t];
                    const int sig = (int)((eq >> t) & 1u) * 12 - 5;                   // 4*sigma - 1: 7 / -5
                    const int cd = (d | 3) + sig;
                    const int cl = old + (int)(int8_t)(0xF5F6F7FCu >> ((old & 3) * 8));
                    const int cu = u + (int)(int8_t)(0xF6F7FCF9u >> ((u & 3) * 8));
                    int h = cd > cl ? cd : cl;
                    h = h > cu ? h : cu;
                    d = old; u = h; H[t] = h;
                }
                bl[j * 64] = (uint16_t)H[SW_R - 1];
                if (last) {
                    int lastp = H[0];
#pragma unroll
                    for (int t = 1; t < SW_R; t++) { if (t == tl) lastp = H[t]; }
                    if ((lastp >> 2) > (bestp >> 2)) { bestp = lastp; bestj = j; }
                }
            }
        }
        const int thr = -2 * (int)(lq / 2) + 2 * (int)lq;                      // sw.erl:69 after the last row
        const int score = bestp >> 2;
        if (score < thr) { io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0; continue; }
        // pure diagonal ending at (Lq, bestj)?  matches counted on the bit-planes
        bool diag = bestj >= lq;
        if (diag) {
            const uint32_t sh = bestj - lq;           // window base of query base 0
            int m = 0;
            for (uint32_t i = 0; i < lq; i += 32) {
                const uint32_t b = sh + i, wd = b >> 5, bo = b & 31;
                uint32_t xlo = wlo[wd * 64 + lane] >> bo, xhi = whi[wd * 64 + lane] >> bo;
                if (bo && wd + 1 < PW) { xlo |= wlo[(wd + 1) * 64 + lane] << (32 - bo); xhi |= whi[(wd + 1) * 64 + lane] << (32 - bo); }
                uint32_t e = ~((qlo[(i >> 5) * 64 + lane] ^ xlo) | (qhi[(i >> 5) * 64 + lane] ^ xhi));
                if (lq - i < 32) e &= (1u << (lq - i)) - 1u;
                m += __popc(e);
            }
            diag = (3 * m - (int)lq) == score;        // 2m - (Lq - m)
        }
        if (!diag) { io.c_status[c] = CS_PENDING; io.tq[atomicAdd(&io.ctr->tq_count, 1u)] = c; continue; }
        uint8_t *slot = io.cig + (uint64_t)c * io.cig_stride;
        int p = (int)io.cig_stride, len = 0;
        cig_emit(slot, p, len, lq, 'M');
        io.c_score[c] = score;
        io.c_ciglen[c] = (uint32_t)len;
        io.c_status[c] = len > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
    }
}

// ------------------------------------------------------------------------------------------------
// k_sw_pk<LQ, TRACE>: the extend stage's workhorse for the common read lengths.  The extend stage is
// integer-VALU bound (SQ counters: VALU active ~95 % of the kernel at 4 cycles per wave64 op), so this
// version halves the instruction count per cell:
//   * TWO alignments per lane, cells packed as 2 x i16 (4V+code each) -> v_pk_add/mad/max/min_i16
//     update both at once (|4V| < 2^15 for every read length this is instantiated for);
//   * the WHOLE query column (LQ rows) lives in registers: no strips, no boundary row, no padding rows;
//     the last query row is register LQ-1 at compile time;
//   * per column the row-match bits of both alignments for 16 rows come from one xor/xor/bitop on
//     pre-interleaved query bit-planes; per cell the 4*sigma+2 term is (bit * 12 - 2) as one pk_mad.
// Cell algebra here: h = 8V + code with ONE-HOT codes left 1, up 2, up_left 4 (same order as the term
// order left < up < up_left, so a packed max still is lists:max); row 0 is stored as (V=0, left), which
// gives the same Uvalue as `undef` does in the reference (-2).  base = h & ~7:
//   up_left candidate = base(diag) + 8*sigma + 4                 = base + (bit*24 - 4)
//   left    candidate = base(left) + 8*(-2 + [code==left]) + 1   = base + 8*(h&1) - 15
//   up      candidate = base(up)   + 8*(-2 + [code==up])   + 2   = base + 4*(h&2) - 14
typedef short bwt_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(bwt_s16x2, a) + __builtin_bit_cast(bwt_s16x2, b));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(bwt_s16x2, a), __builtin_bit_cast(bwt_s16x2, b)));
}
// a*b+c on both halves in ONE op (hipcc splits power-of-two multipliers into shift + add otherwise)
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// TRACE = false: score pass over the DP queue (non-diagonal alignments go on to io.tq2).
// TRACE = true : the same DP over io.tq2 that also keeps the 2-bit direction of every cell (8 rows of both
//                alignments per dword, [column][row group][lane] in a per-wave HBM slab) and walks the traceback.
// in_q / in_count (TRACE): another queue than tq2 -- the WIDE pass: the windows the first pass found wider than its cap (they went
// to the generic trace queue io.tq) once more with a cap of up to ~1100 bases, i.e. as much window as LDS holds planes for; what is
// wider still, or of another read length, goes to over_q / over_count instead of io.tq.  (The generic kernel is a lane per
// alignment at ~18 instructions per cell, one wave per SIMD when a batch has a few ten thousand such windows: 5.3 ms for 56 k of
// them on the repeat-rich batch; two per lane at ~4 per cell here.)
template <int LQ, bool TRACE>
__global__ __launch_bounds__(256) void k_sw_pk(SwIO io, uint32_t n_items, uint32_t wcap, uint32_t *__restrict__ slab_dir,
                                               uint32_t maxW1, uint32_t skip_upto, const uint32_t *__restrict__ in_q,
                                               const unsigned int *__restrict__ in_count, uint32_t *__restrict__ over_q,
                                               unsigned int *__restrict__ over_count) {
    constexpr int QW = (LQ + 31) / 32, NG = (LQ + 15) / 16, NDG = (LQ + 7) / 8;   // NDG: direction dwords per column
    extern __shared__ uint32_t lds_sw[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t PW = (wcap + 31) / 32;
    const uint32_t set_words = (2 * PW + 2 * QW) * 64;                      // one set = 64 candidates
    uint32_t *pl = lds_sw + (uint64_t)(threadIdx.x >> 6) * 2 * set_words;

    const uint32_t *const queue = in_q ? in_q : TRACE ? io.tq2 : io.dq;
    if (in_q) n_items = *in_count; else if (TRACE) n_items = io.ctr->tq2_count; else if (io.dq) n_items = io.ctr->dq_count;
    if (n_items <= skip_upto) return;                            // (a short queue: k_sw_wave took it)
    uint32_t *const dirs = TRACE ? slab_dir + (uint64_t)wave * maxW1 * NDG * 64 + lane : nullptr;   // [(j*NDG + g)*64]
    for (uint32_t g = wave; (uint64_t)g * 128 < n_items; g += nwaves) {
        uint32_t cc[2], lq[2], w[2];
        int64_t wo[2];
        uint64_t qo[2];
        int state[2];     // 0 run, 1 idle, 2 out of range, 3 to the trace kernel, 4 to the fp64 kernel
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t item = g * 128 + h * 64 + lane;
            cc[h] = 0;
            lq[h] = 0; w[h] = 0; wo[h] = 0; qo[h] = 0; state[h] = 1;
            if (item < n_items) {
                cc[h] = queue ? queue[item] : item;
                lq[h] = io.c_qlen[cc[h]]; w[h] = io.c_wlen[cc[h]]; wo[h] = io.c_woff[cc[h]]; qo[h] = io.c_qoff[cc[h]];
                if (wo[h] < 0 || (uint64_t)wo[h] + w[h] > io.rbuf_len || w[h] == 0 || lq[h] == 0) state[h] = 2;
                else if (w[h] > wcap || lq[h] != (uint32_t)LQ) state[h] = 3;
                else state[h] = 0;
            }
        }
        // ---- prologue: coalesced loads -> ballot bit-planes -> LDS, for both sets of 64 candidates
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t *const wlo = pl + h * set_words, *const whi = wlo + PW * 64, *const qlo = wlo + 2 * PW * 64,
                           *const qhi = wlo + (2 * PW + QW) * 64;
            unsigned long long badmask = 0;
            if (io.qplanes && io.wplanes) {
                // the bit-planes of reads and Ref1 exist already (interleaved, k_read_planes): every lane takes its own
                // candidate's words with independent 16-byte loads -- one memory latency for the whole prologue instead of a
                // dependent round per candidate
                if (state[h] == 0) {
                    uint32_t bad = 0;
                    for (uint32_t wd = 0; wd < PW; wd++) {
                        if (32 * wd >= w[h]) break;
                        const uint4 v = planes4(io.wplanes, (uint64_t)wo[h] + 32ull * wd);
                        const uint32_t mk = w[h] - 32 * wd < 32 ? (1u << (w[h] - 32 * wd)) - 1u : 0xffffffffu;
                        wlo[wd * 64 + lane] = v.x & mk; whi[wd * 64 + lane] = v.y & mk;
                        bad |= (v.z | v.w) & mk;
                    }
#pragma unroll
                    for (uint32_t wd = 0; wd < (uint32_t)QW; wd++) {
                        const uint4 v = planes4(io.qplanes, qo[h] + 32ull * wd);
                        const uint32_t mk = (uint32_t)LQ - 32 * wd < 32 ? (1u << ((uint32_t)LQ - 32 * wd)) - 1u : 0xffffffffu;
                        qlo[wd * 64 + lane] = v.x & mk; qhi[wd * 64 + lane] = v.y & mk;
                        bad |= (v.z | v.w) & mk;
                    }
                    if (bad) state[h] = 4;
                }
            } else
            for (int k = 0; k < 64; k++) {
                const uint32_t kw = __shfl(state[h] == 0 ? w[h] : 0u, k, 64);
                if (kw == 0) continue;
                const uint64_t kro = (uint64_t)__shfl((long long)wo[h], k, 64), kqo = (uint64_t)__shfl((long long)qo[h], k, 64);
                for (uint32_t c0 = 0; c0 < kw; c0 += 64) {
                    const uint32_t idx = c0 + lane;
                    const int code = idx < kw ? base_code(io.rbuf[kro + idx]) : 0;
                    const unsigned long long lo = __ballot(code & 1), hi = __ballot(code & 2), bad = __ballot(code < 0);
                    if (bad) badmask |= 1ull << k;
                    if (lane == 0) {
                        const uint32_t wd = 2 * (c0 >> 6);
                        wlo[wd * 64 + k] = (uint32_t)lo; whi[wd * 64 + k] = (uint32_t)hi;
                        if (wd + 1 < PW) { wlo[(wd + 1) * 64 + k] = (uint32_t)(lo >> 32); whi[(wd + 1) * 64 + k] = (uint32_t)(hi >> 32); }
                    }
                }
#pragma unroll
                for (uint32_t c0 = 0; c0 < (uint32_t)LQ; c0 += 64) {
                    const uint32_t idx = c0 + lane;
                    const int code = idx < (uint32_t)LQ ? base_code(io.qbuf[kqo + idx]) : 0;
                    const unsigned long long lo = __ballot(code & 1), hi = __ballot(code & 2), bad = __ballot(code < 0);
                    if (bad) badmask |= 1ull << k;
                    if (lane == 0) {
                        const uint32_t wd = 2 * (c0 >> 6);
                        qlo[wd * 64 + k] = (uint32_t)lo; qhi[wd * 64 + k] = (uint32_t)hi;
                        if (wd + 1 < (uint32_t)QW) { qlo[(wd + 1) * 64 + k] = (uint32_t)(lo >> 32); qhi[(wd + 1) * 64 + k] = (uint32_t)(hi >> 32); }
                    }
                }
            }
            if (state[h] == 0 && ((badmask >> lane) & 1ull)) state[h] = 4;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (state[h] == 2) { io.c_status[cc[h]] = CS_WINDOW_OOR; io.c_ciglen[cc[h]] = 0; }
            else if (state[h] == 3) {                                  // generic trace kernel
                io.c_status[cc[h]] = CS_PENDING;
                if (over_q) over_q[atomicAdd(over_count, 1u)] = cc[h]; else io.tq[atomicAdd(&io.ctr->tq_count, 1u)] = cc[h];
            }
            else if (state[h] == 4) { io.c_status[cc[h]] = CS_NEED_FP; io.fpq[atomicAdd(&io.ctr->fp_count, 1u)] = cc[h]; }
        }
        const bool runA = state[0] == 0, runB = state[1] == 0;
        if (!runA && !runB) continue;
        const uint32_t *const wloA = pl, *const whiA = pl + PW * 64, *const qloA = pl + 2 * PW * 64, *const qhiA = pl + (2 * PW + QW) * 64;
        const uint32_t *const wloB = wloA + set_words, *const whiB = whiA + set_words, *const qloB = qloA + set_words,
                       *const qhiB = qhiA + set_words;
        // query planes, rows 16m..16m+15 of A in the low half, of B in the high half
        uint32_t q2lo[NG], q2hi[NG];
#pragma unroll
        for (int m = 0; m < NG; m++) {
            const int wd = m >> 1, sh = (m & 1) * 16;
            q2lo[m] = ((qloA[wd * 64 + lane] >> sh) & 0xffffu) | ((qloB[wd * 64 + lane] >> sh) << 16);
            q2hi[m] = ((qhiA[wd * 64 + lane] >> sh) & 0xffffu) | ((qhiB[wd * 64 + lane] >> sh) << 16);
        }
        uint32_t H[LQ];
#pragma unroll
        for (int t = 0; t < LQ; t++) {                      // column 0: V(i,0) = -(i+1), dir up (sw.erl:73-77)
            const uint32_t v = (uint32_t)(-(t + 2) * 8 + 2) & 0xffffu;
            H[t] = v | (v << 16);
        }
        int bestA = -(LQ + 1), bestB = -(LQ + 1);           // V(Lq,0)
        uint32_t bestjA = 0, bestjB = 0;
        const uint32_t wA = runA ? w[0] : 0, wB = runB ? w[1] : 0, wmax = wA > wB ? wA : wB;
        uint32_t rloA = 0, rhiA = 0, rloB = 0, rhiB = 0;
        const uint32_t K8 = 0x00080008u, K4 = 0x00040004u, K24 = 0x00180018u, KM4 = 0xFFFCFFFCu, KM15 = 0xFFF1FFF1u,
                       KM14 = 0xFFF2FFF2u;
        for (uint32_t j = 1; j <= wmax; j++) {
            const uint32_t b = j - 1;
            if ((b & 31) == 0) {
                const uint32_t wd = (b >> 5) * 64 + lane;
                rloA = wloA[wd]; rhiA = whiA[wd]; rloB = wloB[wd]; rhiB = whiB[wd];
            }
            const uint32_t sh = b & 31;
            const uint32_t rl2 = (0u - ((rloA >> sh) & 1u)) >> 16 | (0u - ((rloB >> sh) & 1u)) << 16;
            const uint32_t rh2 = (0u - ((rhiA >> sh) & 1u)) >> 16 | (0u - ((rhiB >> sh) & 1u)) << 16;
            uint32_t e2[NG];
#pragma unroll
            for (int m = 0; m < NG; m++) e2[m] = ~((q2lo[m] ^ rl2) | (q2hi[m] ^ rh2));   // bit: row matches this column
            uint32_t dbase = 0;                              // row 0, column j-1: V = 0
            uint32_t cu = KM14;                              // row 0 as the up source: 8*(-2) + 2
            uint32_t dacc = 0;
#pragma unroll
            for (int t = 0; t < LQ; t++) {
                const uint32_t old = H[t];
                const uint32_t base_o = old & 0xFFF8FFF8u;
                const uint32_t cl = pk_add(pk_mad(old & 0x00010001u, K8, base_o), KM15);
                const uint32_t x = (e2[t >> 4] >> (t & 15)) & 0x00010001u;
                const uint32_t cd = pk_add(dbase, pk_mad(x, K24, KM4));
                const uint32_t h = pk_max(pk_max(cd, cl), cu);
                H[t] = h;
                dbase = base_o;
                cu = pk_add(pk_mad(h & 0x00020002u, K4, h & 0xFFF8FFF8u), KM14);
                if (TRACE) {        // one-hot 1/2/4 -> 0/1/2 (left/up/up_left), 8 rows per 16-bit half, newest in the low bits
                    dacc = ((dacc << 2) & 0xFFFCFFFCu) | ((h >> 1) & 0x00030003u);
                    if ((t & 7) == 7 || t == LQ - 1) { dirs[((uint64_t)j * NDG + (t >> 3)) * 64] = dacc; dacc = 0; }
                }
            }
            const uint32_t lastp = H[LQ - 1];
            const int vA = (int)(int16_t)(lastp & 0xffffu) >> 3, vB = (int)lastp >> 19;
            if (j <= wA && vA > bestA) { bestA = vA; bestjA = j; }
            if (j <= wB && vB > bestB) { bestB = vB; bestjB = j; }
        }
        const int thr = -2 * (LQ / 2) + 2 * LQ;             // sw.erl:69 after the last row
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (state[h] != 0) continue;
            const uint32_t c = cc[h];
            const int score = h ? bestB : bestA;
            const uint32_t bestj = h ? bestjB : bestjA;
            if (score < thr) { io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0; continue; }
            const uint32_t *wlo = h ? wloB : wloA, *whi = h ? whiB : whiA, *qlo = h ? qloB : qloA, *qhi = h ? qhiB : qhiA;
            uint8_t *slot = io.cig + (uint64_t)c * io.cig_stride;
            int p = (int)io.cig_stride, len = 0;
            if (TRACE) {
                // traceback (sw.erl:116-160) from the leftmost maximum of the last row up to row 0.  A step's direction word sits in the
                // slab (one word per column and 8 rows: a diagonal step needs a new one every time), and the step before decides which:
                // a dependent HBM load per step was most of this kernel's time on short queues (220 steps x ~1.5 us against 0.2 ms of DP
                // per wave).  So the words of the next eight DIAGONAL steps are asked for together -- most of a traceback is diagonal
                // runs -- and used as long as the walk stays on the diagonal; the first other step starts a new batch.
                uint32_t i = LQ, j = bestj;
                int runop = -1; uint32_t runlen = 0;
                bool walking = true;
                while (walking) {
                    uint32_t wd[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const bool in = i > (uint32_t)k && j > (uint32_t)k;
                        wd[k] = in ? dirs[((uint64_t)(j - k) * NDG + ((i - 1 - k) >> 3)) * 64] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        int code = -1;
                        if (i > 0) {
                            if (j == 0) code = DIR_UP;
                            else {
                                const uint32_t t = i - 1, grp = t >> 3;
                                const uint32_t rows = (grp == (uint32_t)(NDG - 1)) ? (uint32_t)(LQ - 8 * (NDG - 1)) : 8u;
                                code = (int)((wd[k] >> (16 * h + 2 * (rows - 1 - (t & 7)))) & 3u);
                            }
                        }
                        if (code == runop) { runlen++; }
                        else {
                            if (runop >= 0) cig_emit(slot, p, len, runlen, runop == DIR_UL ? 'M' : runop == DIR_UP ? 'I' : 'D');
                            runop = code; runlen = 1;
                        }
                        if (code < 0) { walking = false; break; }
                        if (code == DIR_UL) { i--; j--; } else { if (code == DIR_UP) i--; else j--; break; }
                    }
                }
                io.c_score[c] = score;
                io.c_ciglen[c] = (uint32_t)len;
                io.c_status[c] = len > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
                continue;
            }
            bool diag = bestj >= (uint32_t)LQ;
            if (diag) {                                      // pure diagonal ending at (Lq, bestj)?
                const uint32_t shd = bestj - LQ;
                int m = 0;
#pragma unroll
                for (int i = 0; i < LQ; i += 32) {
                    const uint32_t bb = shd + i, wd = bb >> 5, bo = bb & 31;
                    uint32_t xlo = wlo[wd * 64 + lane] >> bo, xhi = whi[wd * 64 + lane] >> bo;
                    if (bo && wd + 1 < PW) { xlo |= wlo[(wd + 1) * 64 + lane] << (32 - bo); xhi |= whi[(wd + 1) * 64 + lane] << (32 - bo); }
                    uint32_t e = ~((qlo[(i >> 5) * 64 + lane] ^ xlo) | (qhi[(i >> 5) * 64 + lane] ^ xhi));
                    if (LQ - i < 32) e &= (1u << (LQ - i)) - 1u;
                    m += __popc(e);
                }
                diag = (3 * m - LQ) == score;
            }
            if (!diag) { io.c_status[c] = CS_PENDING; io.tq2[atomicAdd(&io.ctr->tq2_count, 1u)] = c; continue; }
            cig_emit(slot, p, len, LQ, 'M');
            io.c_score[c] = score;
            io.c_ciglen[c] = (uint32_t)len;
            io.c_status[c] = len > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sw_band<B>: the same DP on a band of B diagonals, with a certificate that makes the result the full rectangle's.
//
// Cells in diagonal coordinates: o = j - i, band o in [lo, lo + B) with lo = -x, x = (B - 1 - D) div 2, D = W - Lq (the
// window is the read length plus the cluster's width: an alignment the cluster stands for starts and ends on diagonals
// 0..D).  A lane owns one candidate and keeps the band's row in B registers; in place, from k = 0 upwards, cell k takes its
// diagonal source from old[k], its up source from old[k+1] and its left source from new[k-1]; neighbours outside the band
// count as minus infinity.  Directions: 2 bits per cell, two words per row into a per-wave HBM slab ([row][half][lane]).
//
// Why that is exact when the band's best last-row value S* is at least 2 Lq - 3x - 3 (DESIGN.md section 4b has the proof):
// every value of the reference's table is the affine score (open -2, extend -1) of a real path, so whatever reaches a
// band cell through cells outside the band is bounded: from the left (diagonal < -x: at least x + 1 more insertions than
// deletions) by Z_L(i, o) = 2i - 3x - 5 - (o - lo), from the right (diagonal > D + x, to come back: insertions again) by
// Z_R(i, o) = 2i - 3 (hi + 1 - o) - 1.  By induction over the cells in table order, a band cell either has the same value
// AND direction in both tables or both values are <= Z = max(Z_L, Z_R); T - Z never decreases along a traceback walked
// backwards; so with S* > max Z(last row) = 2 Lq - 3x - 4 the maximum, its leftmost column and the whole traceback are
// those of the full table, and every row's maximum stays above the early-exit threshold of sw.erl:69 (3x + 3 <= 2 (Lq div 2)
// is checked).  Candidates that are not eligible or whose S* is lower go on to the full-rectangle kernels untouched.
#define SWB_XMIN 3
#define SWB_B3 64                  // diagonals of the third tier (72 would settle every square window of a 100-base read, no_match included -- see the settle
                                   // step -- but needs 172 VGPRs: two waves per SIMD, 6.0 ms where 64 diagonals take 4.0 on the repeat-rich batch; measured, round 4)
#define SWB_MAX_LQ 1500            // 8 V + code of both candidates of a lane in 16 bits each
// Two candidates per lane (low / high 16 bits of every register, the cell algebra of k_sw_pk): they share the row loop and
// the cell index k, each has its own band offset x, window and last row.
// The queue goes through three tiers: B = 20 first (a sixth fewer cells than 24; with D = 5 it covers a loss of 24 against the
// perfect score, e.g. a 5-base insertion and two substitutions), then B = 32 for what that certificate did not cover (a loss
// of up to 39-42), then B = 64 (a loss of up to 96 on a square window: with Lq = 100 everything sw.erl:62 does not abort --
// reads from diverged copies of a repeat family, 20-30 mismatches, are this tier's), then the full rectangle.  in_q / in_count:
// this tier's queue (in_q2 / in_count2: a second one behind it, or null); rest / ctl[1]: what it passes on; ctl[0]: the number of
// candidates it was given; ctl[2]: its work counter.  NEXT > 0: the next tier has NEXT diagonals, and a candidate that cannot be
// settled there -- not eligible, or run here with a best score below the next tier's own threshold: a wider band only adds paths, so
// its best score is at least this one, but for the substitution-rich candidates that fail here it is the same diagonal again --
// goes to skip_q / skip_ctr instead (the queue of the tier after the next), and the next tier does not compute it in vain.
#define SWB_WAVES(B) ((B) > 48 ? 2 : (B) > 24 ? 3 : 4)      // (least waves per SIMD asked of the compiler; 64 diagonals: 2..3 -- held to exactly three it spilled 68
                                                            // registers into scratch at 168 VGPRs; given the range it takes 162 and still runs three)
#define SWB_RB(B) ((B) > 32 ? 4 : 8)      // rows of directions a traceback stages in LDS at a time (B / 8 words per row and lane)
// is a candidate one a tier of BB diagonals runs?  (x: its band offset there)
template <int BB>
__device__ __forceinline__ bool swb_eligible(const SwIO &io, uint32_t lq, uint32_t w, int64_t wo, uint32_t slab_rows, int &x) {
    const int D = (int)w - (int)lq;
    x = (BB - 1 - D) >> 1;
    // (round 4: no "3 x + 3 <= 2 (Lq div 2)" here any more -- the kernel compares its best score with the abort threshold of sw.erl:62,69
    // itself, see the settle step; wo >= 64: the band's first cell lies x <= 35 bases in front of the window)
    return D >= 0 && D <= BB - 1 - 2 * SWB_XMIN && lq >= 2 && lq < slab_rows && lq <= SWB_MAX_LQ && wo >= 64 && (uint64_t)wo + w <= io.rbuf_len;
}
template <int B, int NEXT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SWB_WAVES(B), (B) > 48 ? 3 : SWB_WAVES(B))))
void k_sw_band(SwIO io, const uint32_t *__restrict__ in_q, const unsigned int *__restrict__ in_count, const uint32_t *__restrict__ in_q2,
               const unsigned int *__restrict__ in_count2, uint32_t *__restrict__ rest, unsigned int *__restrict__ ctl,
               uint32_t *__restrict__ skip_q, unsigned int *__restrict__ skip_ctr, uint32_t *__restrict__ slab, uint32_t slab_rows,
               uint32_t skip_upto) {
    static_assert(B >= 12 && B <= 96, "one to three match words per row");
    constexpr int XMAX = (B - 1) / 2, NDW = (B + 7) / 8, NW = (B + 31) / 32;     // NW: 32-cell match words of a row
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    constexpr int RB = SWB_RB(B);                                                   // rows of directions a traceback stages in LDS at a time
    __shared__ uint32_t tb_lds[4][RB * NDW * 64];
    uint32_t *const tb = tb_lds[threadIdx.x >> 6] + lane;                           // [(r * NDW + k / 8) * 64]: every lane its own column
    const uint32_t n1 = *in_count, n_items = n1 + (in_q2 ? *in_count2 : 0u);
    auto item_at = [&](uint32_t t) { return t < n1 ? in_q[t] : in_q2[t - n1]; };
    if (n_items <= skip_upto) {
        // a short queue is k_sw_wave's (a lane per alignment would run a handful of waves for the serial length of a DP): passed through
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += gridDim.x * blockDim.x) rest[t] = item_at(t);
        if (wave == 0 && lane == 0) { ctl[0] = n_items; ctl[1] = n_items; }
        return;
    }
    if (wave == 0 && lane == 0) ctl[0] = n_items;
    uint32_t *const dirs = slab + (uint64_t)wave * slab_rows * NDW * 64 + lane;     // [(i * NDW + k / 8) * 64]: 8 cells x 2 candidates per word
    const uint32_t F8 = 0xFFF8FFF8u, K8 = 0x00080008u, K4 = 0x00040004u, K24 = 0x00180018u, KM4 = 0xFFFCFFFCu, KM15 = 0xFFF1FFF1u,
                   KM14 = 0xFFF2FFF2u;
    // groups of 128 candidates are handed out by a counter: a group's time varies with its tracebacks, and the queue is a few
    // groups per resident wave (every wave leaves when the counter has passed the end of the queue)
    for (;;) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(&ctl[2], 1u);
        g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        if ((uint64_t)g * 128 >= n_items) break;
        uint32_t c[2], lq[2], w[2];
        int64_t wo[2];
        uint64_t qo[2];
        int x[2], D[2];
        bool have[2], run[2], settled[2] = {false, false}, ran[2], next_ok[2] = {true, true};
        int next_thr[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t item = g * 128 + h * 64 + lane;
            have[h] = item < n_items;
            c[h] = 0; lq[h] = 0; w[h] = 0; wo[h] = 0; qo[h] = 0;
            if (have[h]) { c[h] = item_at(item); lq[h] = io.c_qlen[c[h]]; w[h] = io.c_wlen[c[h]]; wo[h] = io.c_woff[c[h]]; qo[h] = io.c_qoff[c[h]]; }
            D[h] = (int)w[h] - (int)lq[h];
            run[h] = have[h] && swb_eligible<B>(io, lq[h], w[h], wo[h], slab_rows, x[h]);
            if (NEXT > 0 && have[h]) {                                                   // (before the idle half's values are replaced below)
                int xn;
                next_ok[h] = swb_eligible<(NEXT > 0 ? NEXT : B)>(io, lq[h], w[h], wo[h], slab_rows, xn);
                next_thr[h] = 2 * (int)lq[h] - 3 * xn - 3;
            }
            if (!run[h]) { lq[h] = 0; wo[h] = 64; qo[h] = 0; x[h] = XMAX; D[h] = 0; }     // (its half computes on, harmlessly, on words that exist)
        }
        int best[2] = {-(1 << 28), -(1 << 28)}, kb[2] = {0, 0};
        ran[0] = run[0]; ran[1] = run[1];
        if (run[0] || run[1]) {
            uint32_t H[B];
#pragma unroll
            for (int k = 0; k < B; k++) H[k] = 0x00010001u;                            // row 0: V = 0, stored as `left` (Uvalue -2 like undef)
            // window plane words: rlw[h][t] = the 32 bases from P0 + 32 (b / 32 + t - 1) of the low plane, rhw of the high one; a row
            // needs NW + 1 of them (cells k = 0 .. B - 1 sit on bases P0 + b + k); word 0 is filled by the first row's shift
            uint32_t qlw[2] = {0, 0}, qhw[2] = {0, 0}, rlw[2][NW + 1], rhw[2][NW + 1];
            uint64_t P0[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                P0[h] = (uint64_t)wo[h] - (uint32_t)x[h];                                // Ref1 base of cell (1, k = 0)'s column
                rlw[h][0] = 0; rhw[h][0] = 0;
#pragma unroll
                for (int t = 1; t <= NW; t++) {
                    const uint4 v = planes4(io.wplanes, P0[h] + 32 * (t - 1));
                    rlw[h][t] = v.x; rhw[h][t] = v.y;
                }
            }
            auto row = [&](uint32_t i, auto first) {
                const uint32_t b = i - 1, sh = b & 31u;
                uint32_t e[2][NW];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (sh == 0) {
                        // a half whose candidate has no row b (the shorter of the lane's two, a half that does not run) stays on
                        // words that exist: the rows of a 1500-base partner would walk it far past its own read and window
                        const uint32_t bb = b < lq[h] ? b : 0u;
                        const uint4 qv = planes4(io.qplanes, qo[h] + bb);
                        qlw[h] = qv.x; qhw[h] = qv.y;
#pragma unroll
                        for (int t = 0; t < NW; t++) { rlw[h][t] = rlw[h][t + 1]; rhw[h][t] = rhw[h][t + 1]; }
                        const uint4 v = planes4(io.wplanes, P0[h] + bb + 32 * NW);
                        rlw[h][NW] = v.x; rhw[h][NW] = v.y;
                    }
                    const uint32_t QL = 0u - ((qlw[h] >> sh) & 1u), QH = 0u - ((qhw[h] >> sh) & 1u);
#pragma unroll
                    for (int t = 0; t < NW; t++) {
                        const uint32_t rl = __builtin_amdgcn_alignbit(rlw[h][t + 1], rlw[h][t], sh), rh = __builtin_amdgcn_alignbit(rhw[h][t + 1], rhw[h][t], sh);
                        e[h][t] = ~((rl ^ QL) | (rh ^ QH));                              // bit k: the row's base equals the base of cell 32 t + k's column
                    }
                }
                // 16 cells of both candidates per word: cell k's bit of candidate 0 at bit k % 16, of candidate 1 at 16 + k % 16
                uint32_t E[2 * NW];
#pragma unroll
                for (int t = 0; t < NW; t++) {
                    E[2 * t] = (e[0][t] & 0xffffu) | (e[1][t] << 16);
                    E[2 * t + 1] = (e[0][t] >> 16) | (e[1][t] & 0xffff0000u);
                }
                const int kc0 = x[0] - (int)i, kc1 = x[1] - (int)i;                      // column 0 sits at k = kc in this row (rows <= x)
                const uint32_t col0 = (uint32_t)(-(int)(i + 1) * 8 + 2) & 0xffffu;       // V(i, 0) = -(i + 1), up (sw.erl:73-77)
                uint32_t left = 0x80008000u;
                uint32_t dacc = 0;
#pragma unroll
                for (int k = 0; k < B; k++) {
                    const uint32_t old = H[k];
                    const uint32_t xk = (E[k >> 4] >> (k & 15)) & 0x00010001u;
                    const uint32_t cd = pk_add(old & F8, pk_mad(xk, K24, KM4));          // 8 sigma + 4
                    uint32_t hh = pk_max(cd, left);
                    if (k + 1 < B) {
                        const uint32_t up = H[k + 1];
                        hh = pk_max(hh, pk_add(pk_mad(up & 0x00020002u, K4, up & F8), KM14));   // 8 (-2 + [up]) + 2
                    }
                    if (decltype(first)::value && k < XMAX) {
                        if (k == kc0) hh = (hh & 0xffff0000u) | col0;
                        if (k == kc1) hh = (hh & 0x0000ffffu) | (col0 << 16);
                    }
                    H[k] = hh;
                    left = pk_add(pk_mad(hh & 0x00010001u, K8, hh & F8), KM15);          // 8 (-2 + [left]) + 1
                    const uint32_t code2 = hh & 0x00060006u;                             // one-hot 1/2/4 -> 0/1/2, still shifted left by one
                    dacc = (k & 7) == 0 ? code2 >> 1 : dacc | (code2 << (2 * (k & 7) - 1));
                    if ((k & 7) == 7 || k == B - 1) dirs[(uint64_t)(i * NDW + (k >> 3)) * 64] = dacc;
                }
            };
            // a candidate's last row: columns 1..W are k <= D + x; leftmost maximum (sw.erl get_starting_point)
            auto last_row = [&](int h) {
                const int kmax = D[h] + x[h];
#pragma unroll
                for (int k = 0; k < B; k++) {
                    const int v = (h ? (int)H[k] >> 19 : (int)(int16_t)(H[k] & 0xffffu) >> 3);
                    if (k <= kmax && v > best[h]) { best[h] = v; kb[h] = k; }
                }
            };
            // rows up to the shorter candidate's last one, then on to the longer one's 
            const uint32_t lqm = lq[0] > lq[1] ? lq[0] : lq[1];
            const uint32_t lqs = run[0] && run[1] ? (lq[0] < lq[1] ? lq[0] : lq[1]) : lqm;
            const uint32_t nfirst = lqs < (uint32_t)XMAX ? lqs : (uint32_t)XMAX;
            for (uint32_t i = 1; i <= nfirst; i++) row(i, std::true_type{});
            for (uint32_t i = nfirst + 1; i <= lqs; i++) row(i, std::false_type{});
            if (run[0] && lq[0] == lqs) last_row(0);
            if (run[1] && lq[1] == lqs) last_row(1);
            if (lqm > lqs) {
                for (uint32_t i = lqs + 1; i <= lqm; i++) row(i, std::false_type{});
                if (lq[0] == lqm) last_row(0); else last_row(1);
            }
            // tracebacks (sw.erl:116-160) of the candidates the certificate covers, both of the lane side by side: the directions
            // come back from the slab RB rows at a time (one round trip per RB rows instead of one per step), parked in LDS
            bool tr[2], nomatch[2];
            uint32_t ti[2];
            int tk[2], trun[2] = {-1, -1}, tp[2], tlen[2] = {0, 0};
            uint32_t trl[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // Certificate: best >= 2 Lq - 3 x - 3 -- the banded maximum is the full table's.  It is a hit when it also reaches the
                // abort threshold of sw.erl:62,69 (then no row's maximum on the way fell below its own threshold either: a traceback cell
                // in row r holds >= best - 2 (Lq - r)).  Below the threshold the answer is no_match -- and so it is, without the
                // certificate, whenever best AND everything the band cannot see (last-row cells outside it and band cells reached from
                // outside: all <= 2 Lq - 3 x - 4) lie below the threshold: a 72-diagonal band on a square window of a 100-base
                // read (x = 35: 91 < 100) settles EVERY candidate, hit or not, and nothing of it is left for the full rectangle.
                const int thr_nm = 2 * (int)lq[h] - 2 * (int)(lq[h] / 2), zmax = 2 * (int)lq[h] - 3 * x[h] - 4;
                const bool cert = run[h] && best[h] > zmax;
                tr[h] = cert && best[h] >= thr_nm;
                nomatch[h] = run[h] && best[h] < thr_nm && (cert || zmax < thr_nm);
                ti[h] = tr[h] ? lq[h] : 0; tk[h] = kb[h]; tp[h] = (int)io.cig_stride;
            }
            const uint32_t top = ti[0] > ti[1] ? ti[0] : ti[1];
            for (int blk = top ? (int)((top - 1) / RB) : -1; blk >= 0; blk--) {
                const uint32_t r0 = (uint32_t)blk * RB + 1;                              // rows r0 .. r0 + RB - 1
#pragma unroll
                for (int r = 0; r < RB; r++)
#pragma unroll
                    for (int gq = 0; gq < NDW; gq++)
                        tb[(r * NDW + gq) * 64] = r0 + r <= top ? dirs[(uint64_t)((r0 + r) * NDW + gq) * 64] : 0u;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint8_t *const slot = io.cig + (uint64_t)c[h] * io.cig_stride;
                    while (tr[h] && ti[h] >= r0) {
                        // a diagonal run keeps k: the codes of this k in all staged rows at once, then as many up_left steps as
                        // there are in a row from row i downwards (stopping short of column 0, whose direction is not stored)
                        {
                            const uint32_t gsel = ((uint32_t)tk[h] >> 3) * 64, shb = 16 * h + 2 * (tk[h] & 7);
                            uint32_t ul = 0;
#pragma unroll
                            for (int r = 0; r < RB; r++) ul |= (((tb[r * NDW * 64 + gsel] >> shb) & 3u) == (uint32_t)DIR_UL ? 1u : 0u) << r;
                            const int rowmin = x[h] - tk[h] + 1;                          // rows below this one are column 0 or left of it
                            if (rowmin > (int)r0) ul &= ~0u << (rowmin - (int)r0);
                            const uint32_t top1 = ~(ul << (31u - (ti[h] - r0)));          // bit 31 = row i, zeros where up_left
                            const uint32_t nrun = top1 ? (uint32_t)__clz((int)top1) : 32u;
                            if (nrun) {
                                if (trun[h] == DIR_UL) trl[h] += nrun;
                                else {
                                    if (trun[h] >= 0) cig_emit(slot, tp[h], tlen[h], trl[h], trun[h] == DIR_UP ? 'I' : 'D');
                                    trun[h] = DIR_UL; trl[h] = nrun;
                                }
                                ti[h] -= nrun;
                                continue;
                            }
                        }
                        int code = DIR_UP;                                               // column 0
                        if ((int)ti[h] - x[h] + tk[h] != 0)
                            code = (int)((tb[((ti[h] - r0) * NDW + ((uint32_t)tk[h] >> 3)) * 64] >> (16 * h + 2 * (tk[h] & 7))) & 3u);
                        if (code == trun[h]) trl[h]++;
                        else {
                            if (trun[h] >= 0) cig_emit(slot, tp[h], tlen[h], trl[h], trun[h] == DIR_UL ? 'M' : trun[h] == DIR_UP ? 'I' : 'D');
                            trun[h] = code; trl[h] = 1;
                        }
                        if (code == DIR_UL) ti[h]--; else if (code == DIR_UP) { ti[h]--; tk[h]++; } else tk[h]--;
                        if (tk[h] < 0 || tk[h] >= B) tr[h] = false;                      // (cannot happen under the certificate)
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (tr[h]) {
                    uint8_t *const slot = io.cig + (uint64_t)c[h] * io.cig_stride;
                    cig_emit(slot, tp[h], tlen[h], trl[h], trun[h] == DIR_UL ? 'M' : trun[h] == DIR_UP ? 'I' : 'D');
                    io.c_score[c[h]] = best[h];
                    io.c_ciglen[c[h]] = (uint32_t)tlen[h];
                    io.c_status[c[h]] = tlen[h] > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
                    settled[h] = true;
                } else if (nomatch[h]) {
                    io.c_status[c[h]] = CS_NO_MATCH; io.c_ciglen[c[h]] = 0;
                    settled[h] = true;
                }
        }
        // what is left goes on: to the next tier, or past it when it cannot settle the candidate either
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool left = have[h] && !settled[h];
            const bool past = NEXT > 0 && left && (!next_ok[h] || (ran[h] && best[h] < next_thr[h]));
#pragma unroll
            for (int way = 0; way < (NEXT > 0 ? 2 : 1); way++) {
                const bool pass = way ? past : left && !past;
                const unsigned long long m = __ballot(pass);
                if (m) {
                    const int first = __ffsll((long long)m) - 1;
                    uint32_t base = 0;
                    if ((int)lane == first) base = atomicAdd(way ? skip_ctr : &ctl[1], (uint32_t)__popcll(m));
                    base = (uint32_t)__shfl((int)base, first, 64);
                    if (pass) (way ? skip_q : rest)[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c[h];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sw_cert<X, NW>: in front of the 12-diagonal tier.  Its queue holds the candidates the prefilter found on a SQUARE window with
// 3 .. X + 1 mismatches on the diagonal and no base outside ACGT: diverged copies of a repeat, whose answer is nearly always the
// plain diagonal -- {2 Lq - 3 m, "<Lq>M"} -- and for which the banded DP fills 12 x Lq cells, stores their directions and walks them
// back only to find that out.  A lane takes a candidate, loads the code planes of read and window (the window with 32 bases
// either side), and runs sw_cert.h on them: mismatch masks of the diagonals -X .. X, an upper bound on what any other path can
// gain over the diagonal, a coarse DP over the m + 1 stretches between mismatch rows (the proof is in that header; the same text is
// checked on the CPU against the oracle's full table, tests/cert_check/).  Where the certificate holds the candidate is settled like
// one of the prefilter's own (score, CIG_DIAG length, no text); the others go on to the banded tiers in a queue of their own.
// ~2 k lane-instructions per candidate against ~10 k in k_sw_band<12> -- and no direction slab, no traceback, no extra record.
// `over` / over_count: where candidates with more than X + 1 mismatches go (the queue of the next, wider instance); null: to `rest`
// with the others.  rest_count may be a queue that already holds entries (the wider instance appends to the 20-diagonal tier's DP queue).
template <int X, int NW>
__global__ __launch_bounds__(256) void k_sw_cert(SwIO io, const uint32_t *__restrict__ in_q, const unsigned int *__restrict__ in_count,
                                                 uint32_t *__restrict__ rest, unsigned int *__restrict__ rest_count, uint32_t *__restrict__ over,
                                                 unsigned int *__restrict__ over_count, unsigned int *__restrict__ done_count) {
    constexpr int FL = 256;
    __shared__ uint32_t wq_[4][2][FL + 64];
    WaveQ qr, qo;
    qr.buf = wq_[threadIdx.x >> 6][0]; qr.have = 0;
    qo.buf = wq_[threadIdx.x >> 6][1]; qo.have = 0;
    const uint32_t n_items = *in_count;          // (a short queue is run like any other: the kernel is a few microseconds then)
    uint32_t ndone = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); t0 < n_items; t0 += stride) {          // whole waves iterate together
        const uint32_t t = t0 + (threadIdx.x & 63u);
        const bool have = t < n_items;
        const uint32_t c = have ? in_q[t] : 0u;
        bool pass_on = false, too_many = false;
        if (have) {
            const uint32_t lq = io.c_qlen[c], w = io.c_wlen[c];
            const int64_t wo = io.c_woff[c];
            const uint64_t qo_ = io.c_qoff[c];
            // the planes of the 32 bases in front of the window and of up to 32 behind its last word must exist
            bool ok = w == lq && lq >= 24 && lq <= 32u * NW && wo >= 32 && (uint64_t)wo + 32u * (NW + 1) <= io.rbuf_len;
            int m = 0;
            if (ok) {
                uint32_t qlo[NW], qhi[NW], rlo[NW + 2], rhi[NW + 2];
#pragma unroll
                for (int k = 0; k < NW; k++) { const uint4 v = planes4(io.qplanes, qo_ + 32u * k); qlo[k] = v.x; qhi[k] = v.y; }
#pragma unroll
                for (int k = 0; k < NW + 2; k++) { const uint4 v = planes4(io.wplanes, (uint64_t)wo - 32u + 32u * k); rlo[k] = v.x; rhi[k] = v.y; }
                ok = sw_diag_cert<X, NW>(qlo, qhi, rlo, rhi, (int)lq, m);
            }
            if (ok) {
                uint32_t len = 2;
                for (uint32_t v = lq; v >= 10; v /= 10) len++;
                io.c_score[c] = 2 * (int)lq - 3 * m;
                io.c_ciglen[c] = len | CIG_DIAG;
                io.c_status[c] = len > io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
                ndone++;
            } else { too_many = over != nullptr && m > X + 1; pass_on = !too_many; }
        }
        waveq_push<FL>(qr, pass_on, c, rest, rest_count);
        if (over) waveq_push<FL>(qo, too_many, c, over, over_count);
    }
    waveq_drain<FL>(qr, rest, rest_count);
    if (over) waveq_drain<FL>(qo, over, over_count);
    const unsigned long long sums[1] = {wave_sum_u64(ndone)};
    unsigned long long *const g64[1] = {nullptr};
    unsigned int *const g32[1] = {done_count};
    const uint32_t mx[1] = {0};
    unsigned int *const gmx[1] = {nullptr};
    block_counters(sums, g64, g32, mx, gmx);
}

// ------------------------------------------------------------------------------------------------
// k_sw_wave: a WAVE per alignment, for DP queues too short to fill the chip with lane-per-alignment work (after the
// prefilter a batch of clean reads leaves a few hundred candidates: the packed kernels then run a handful of waves for
// the full serial length of a DP, ~0.5 ms).  Same cells as k_sw (packed 4V + code, ties up_left > up > left, gap cost by
// the neighbour's stored direction), evaluated along anti-diagonals: lane t owns query rows i0 + 2t and i0 + 2t + 1 of a 128-row
// strip and at step k their cells of column k - t; the value of the row above and the window base arrive from lane t - 1, the
// boundary row between strips and the 2-bit directions (16 steps per word) live in LDS.  Traceback by lane 0.
// Runs only when the DP queue holds at most `thresh` candidates (k_sw_pk takes the longer queues); the queue holds clean
// ACGT candidates only (k_sw_prefilter filled it).
// words of 2-bit directions a query row keeps in k_sw_wave (16 steps per word; a row's cells are filed by STEP, see the kernel)
__host__ __device__ __forceinline__ uint32_t swv_dwords(uint32_t max_w) { return ((max_w + 63u) >> 4) + 1u; }
// q_in / q_count: the queue when it is not the DP queue (null: io.dq, ctr->dq_count); q_over / q_over_count: where candidates wider
// than this launch's LDS go (null: io.tq, ctr->tq_count) -- the small-batch path runs the kernel twice, tight LDS first, then with
// all the LDS a block can have on what that pass found too wide, before anything is left to the lane-per-alignment kernel
__global__ __launch_bounds__(256) void k_sw_wave(SwIO io, uint32_t thresh, uint32_t maxW1, uint32_t dwords, uint32_t max_lq,
                                                  const uint32_t *__restrict__ q_in, const unsigned int *__restrict__ q_count,
                                                  uint32_t *__restrict__ q_over, unsigned int *__restrict__ q_over_count) {
    extern __shared__ uint32_t lds_wv[];
    const uint32_t n_total = q_in ? *q_count : io.ctr->dq_count;
    if (!q_in) q_in = io.dq;
    if (!q_over) { q_over = io.tq; q_over_count = &io.ctr->tq_count; }
    if (n_total == 0 || n_total > thresh) return;
    const uint32_t lane = threadIdx.x & 63;
    // (wave-uniform, and the compiler is told so: the item's lengths and the step loop then live in scalar registers)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t *bl = lds_wv + (size_t)(threadIdx.x >> 6) * (maxW1 + (size_t)max_lq * dwords);    // boundary row [j]
    uint32_t *dirs = bl + maxW1;                                                                  // [row][step >> 4]
    for (uint32_t item = wave; item < n_total; item += nwaves) {
#ifdef SWV_TIMING
        const uint64_t T0 = wall_clock64();
#endif
        const uint32_t c = q_in[item];
        const uint32_t lq = io.c_qlen[c], w = io.c_wlen[c];
        const int64_t wo = io.c_woff[c];
        if (wo < 0 || (uint64_t)wo + w > io.rbuf_len || w == 0 || lq == 0) {
            if (lane == 0) { io.c_status[c] = CS_WINDOW_OOR; io.c_ciglen[c] = 0; }
            continue;
        }
        if (w + 1 > maxW1 || lq > max_lq || w > 4094u) {         // wider than this launch's LDS was sized for (or than the column field of
                                                                 // the last row's key) -> the next kernel's queue
            if (lane == 0) { io.c_status[c] = CS_PENDING; q_over[atomicAdd(q_over_count, 1u)] = c; }
            continue;
        }
        const uint8_t *q = io.qbuf + io.c_qoff[c];
        const uint8_t *rw = io.rbuf + wo;
        for (uint32_t j = lane; j <= w; j += 64) bl[j] = DIR_UNDEF;           // row 0: {0, undef}
        __builtin_amdgcn_wave_barrier();
#ifdef SWV_TIMING
        const uint64_t T1 = wall_clock64();
#endif
        // leftmost maximum of the last row, as a key: (V << 12) | (4095 - column) -- the larger value wins, among equal values the
        // smaller column (a window of this kernel is never 4095 bases wide); V(Lq, 0) to begin with
        int bestkey = (int)(((uint32_t)(-(int)(lq + 1)) << 12) | 4095u);
        // Strips of 128 query rows, TWO ROWS PER LANE (lane t: rows i0 + 2t and i0 + 2t + 1), swept along anti-diagonals of lane
        // pairs: at step k lane t fills column k - t of its upper row and then of its lower one -- the lower cell's `up` is the
        // cell just made, its diagonal the upper row's previous cell, both in registers.  A 100-base read is ONE strip of
        // W + 49 steps (two strips of 64 rows: 2 W + 98), and the per-step overhead -- the hand-down of the lane above's value
        // and base, the next window base and boundary value for lane 0, the loop -- is paid once per two cells.
        // Per step a lane needs two values of the lane above it -- that lane's lower cell of the step before (the row above at my
        // column) and the window base it worked on -- and lane 0 needs the next window base and the next value of the boundary
        // row.  All of it stays in registers: window bases and boundary values are fetched 64 at a time, one per lane, and move
        // UP one lane per step (DPP wave_shl:1), so that lane 0 always holds the next one; the shift down the wave is a DPP move
        // too (wave_shr:1, lane 0 keeps the `old` operand).
        // The step is branch-free and as short as it gets (a single wave per SIMD issues one instruction every few cycles: the
        // kernel's time is its instruction count -- 33 us for a 100 x 110 window at ~90 instructions per step before this form):
        // a lane computes from the first step on, its cells are only KEPT once its column is >= 1 (what it makes past column W is
        // read by nobody: the lane below is past W by then as well); a row's 2-bit directions are shifted into a word from the top,
        // one v_alignbit per cell, filed by STEP (word k >> 4, bits 2 (k & 15); column j of lane t at step j + t), so that
        // every lane stores its words at the same steps -- a scalar branch.
        const uint32_t ns = (lq + 127) / 128;
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t i0 = s * 128, ia = i0 + 2 * lane, ib = ia + 1;       // 0-based query rows of this lane
            const bool oka = ia < lq, okb = ib < lq;
            const uint32_t qa = oka ? q[ia] : 0xffu, qb = okb ? q[ib] : 0xfeu;
            int ha = -(int)(ia + 2) * 4 + DIR_UP, hb = -(int)(ib + 2) * 4 + DIR_UP;   // column 0 of my rows
            // value of the row above my upper row at the column before mine (its diagonal input): column 0 of that row
            int da = lane == 0 ? (s == 0 ? (int)DIR_UNDEF : -(int)(i0 + 1) * 4 + DIR_UP) : -(int)(ia + 1) * 4 + DIR_UP;
            uint32_t rc = 0, dwa = 0, dwb = 0;
            const uint32_t nrow = lq - i0 < 128u ? lq - i0 : 128u;              // rows of this strip
            const uint32_t nlane = (nrow + 1) >> 1;                             // lanes that carry one
            const uint32_t ksteps = w + nlane - 1;                              // its last lane reaches column w at this step
            const bool has_next = s + 1 < ns;
            const bool last_is_b = ((lq - 1 - i0) & 1u) != 0;                   // (wave-uniform; only looked at in the last strip)
            const uint32_t ll = (lq - 1 - i0) >> 1;                             // ... the lane with the read's last row
            const bool lastlane = lane == ll;
            uint32_t *drow_a = dirs + (size_t)(oka ? ia : 0u) * dwords, *drow_b = dirs + (size_t)(okb ? ib : 0u) * dwords;
            // (steps in chunks of 64: the chunk's window bases and boundary values are loaded -- and waited for -- once, in front of
            // an inner loop that reads nothing from memory; HN: another strip follows, this one's last row is its boundary row)
            auto sweep = [&](auto HN) {
                for (uint32_t kb0 = 0; kb0 < ksteps; kb0 += 64) {
                    int wch = kb0 + lane < w ? (int)rw[kb0 + lane] : 0;             // window bases of columns kb0 + 1.., one per lane
                    int bch = (int)bl[kb0 + 1 + lane <= w ? kb0 + 1 + lane : 0];    // boundary row (the strip above) at those columns
                    const uint32_t kend = kb0 + 64 < ksteps ? kb0 + 64 : ksteps;
                    for (uint32_t k = kb0 + 1; k <= kend; k++) {
                        const int ua = __builtin_amdgcn_update_dpp(bch, hb, 0x138, 0xF, 0xF, false);             // wave_shr:1; lane 0: the boundary row
                        rc = (uint32_t)__builtin_amdgcn_update_dpp(wch, (int)rc, 0x138, 0xF, 0xF, false);        // ... lane 0: the next window base
                        wch = __builtin_amdgcn_update_dpp(wch, wch, 0x130, 0xF, 0xF, false);                     // wave_shl:1: the chunk moves up a lane
                        bch = __builtin_amdgcn_update_dpp(bch, bch, 0x130, 0xF, 0xF, false);
                        const bool on = k > lane;                                   // my column k - lane is >= 1
                        // upper row: diagonal da (row above, column j - 1), left ha, up ua.  (v_bfe takes the low five bits of its
                        // offset operand: h << 3 selects the byte of h's direction code)
                        const int cda = (da | 3) + ((qa == rc) ? 7 : -5);           // 4(Vul + sigma) + up_left
                        const int cla = ha + __builtin_amdgcn_sbfe((int)0xF5F6F7FCu, (uint32_t)ha << 3, 8u);   // {-4,-9,-10,-11}
                        const int cua = ua + __builtin_amdgcn_sbfe((int)0xF6F7FCF9u, (uint32_t)ua << 3, 8u);   // {-7,-4,-9,-10}
                        int hna = cda > cla ? cda : cla;
                        hna = hna > cua ? hna : cua;
                        // lower row: diagonal = the upper row's cell of column j - 1 (ha as it still is), left hb, up = the cell just made
                        const int cdb = (ha | 3) + ((qb == rc) ? 7 : -5);
                        const int clb = hb + __builtin_amdgcn_sbfe((int)0xF5F6F7FCu, (uint32_t)hb << 3, 8u);
                        const int cub = hna + __builtin_amdgcn_sbfe((int)0xF6F7FCF9u, (uint32_t)hna << 3, 8u);
                        int hnb = cdb > clb ? cdb : clb;
                        hnb = hnb > cub ? hnb : cub;
                        dwa = __builtin_amdgcn_alignbit((uint32_t)hna, dwa, 2u);    // the cell's direction in at the top
                        dwb = __builtin_amdgcn_alignbit((uint32_t)hnb, dwb, 2u);
                        ha = on ? hna : ha;
                        hb = on ? hnb : hb;
                        if (decltype(HN)::value) {
                            const uint32_t j = k - lane;
                            if (lane == 63 && j - 1u < w) bl[j] = (uint32_t)hnb;    // boundary row for the next strip (its 128th row)
                        } else {
                            // the last row's cells go to the boundary array as well (their columns are behind what this chunk took from
                            // it): its leftmost maximum is looked for once, after the sweep, not compared at every step
                            if (lastlane && k > ll) bl[k - ll] = (uint32_t)(last_is_b ? hnb : hna);
                        }
                        if ((k & 15u) == 15u || k == ksteps) {
                            const uint32_t sh = 30u - 2u * (k & 15u);               // a last word that is not full: down to its place
                            if (oka) drow_a[k >> 4] = dwa >> sh;
                            if (okb) drow_b[k >> 4] = dwb >> sh;
                        }
                        da = ua;                                                    // the row above at my column: next step's diagonal
                    }
                }
            };
            if (has_next) sweep(std::true_type{}); else sweep(std::false_type{});
            __builtin_amdgcn_wave_barrier();
        }
#ifdef SWV_TIMING
        const uint64_t T2 = wall_clock64();
#endif
        for (uint32_t j0 = 1; j0 <= w; j0 += 64) {
            const uint32_t j = j0 + lane;
            if (j <= w) {
                const int key = (int)(((uint32_t)((int)bl[j] >> 2) << 12) | (4095u - j));
                bestkey = key > bestkey ? key : bestkey;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(bestkey, o, 64); bestkey = other > bestkey ? other : bestkey; }
        const int best = (bestkey >> 12) * 4;                                   // (as 4 V + code: the score is best >> 2)
        const uint32_t bestj = 4095u - ((uint32_t)bestkey & 4095u);
        {
            // sw.erl:69 after the last row: no_match iff RowMax < -2*(Lq div 2) + 2*Lq
            const int thr = -2 * (int)(lq / 2) + 2 * (int)lq;
            const int score = best >> 2;
            if (score < thr) { if (lane == 0) { io.c_status[c] = CS_NO_MATCH; io.c_ciglen[c] = 0; } }
            else {
                uint8_t *slot = io.cig + (uint64_t)c * io.cig_stride;
                int p = (int)io.cig_stride, len = 0;
                // traceback (sw.erl:116-160): from the leftmost maximum of the last row up to row 0.  By the whole wave: lane l looks
                // at the cell l steps up the diagonal from (ti, tj), a ballot says how far the run of up_left reaches, and the walk
                // jumps it in one go -- a lane following one stored direction after the other paid an LDS round trip per cell, some
                // 200 in a row.  (All lanes keep the same ti / tj / run state; lane 0 alone writes the text.)
                uint32_t ti = lq, tj = bestj;
                int runop = -1; uint32_t runlen = 0;
                auto flush = [&]() { if (runop >= 0) cig_emit_if(lane == 0, slot, p, len, runlen, runop == DIR_UL ? 'M' : runop == DIR_UP ? 'I' : 'D'); };
                auto step = [&](int op, uint32_t cnt) { if (op == runop) runlen += cnt; else { flush(); runop = op; runlen = cnt; } };
                while (ti > 0) {
                    if (tj == 0) { step(DIR_UP, ti); ti = 0; break; }                  // column 0: up all the way (sw.erl:73-77)
                    const bool in = lane < ti && lane < tj;
                    const uint32_t r = in ? ti - 1 - lane : 0u, cc = in ? tj - lane : 1u;
                    const uint32_t kk = cc + ((r & 127u) >> 1);                         // the step at which row r reached column cc
                    const uint32_t code = in ? (dirs[(size_t)r * dwords + (kk >> 4)] >> (2 * (kk & 15u))) & 3u : 0xffu;
                    const unsigned long long ul = __ballot(code == DIR_UL);
                    const uint32_t nul = (uint32_t)__builtin_ctzll(~ul | (1ull << 63)) ;   // (a run of 64 stops at 63: the next round goes on)
                    if (nul) { step(DIR_UL, nul); ti -= nul; tj -= nul; }
                    else {
                        const int c0 = __builtin_amdgcn_readfirstlane((int)code);       // up or left
                        step(c0, 1u);
                        if (c0 == DIR_UP) ti--; else tj--;
                    }
                }
                flush();
                if (lane == 0) {
                    io.c_score[c] = score;
                    io.c_ciglen[c] = (uint32_t)len;
                    io.c_status[c] = len > (int)io.cig_stride ? CS_CIG_OVERFLOW : CS_OK;
                }
            }
        }
#ifdef SWV_TIMING
        if (lane == 0) {
            const uint64_t T3 = wall_clock64();
            atomicMax(&io.ctr->pad2_[0], (unsigned int)(T1 - T0)); atomicMax(&io.ctr->pad2_[1], (unsigned int)(T2 - T1));
            atomicMax(&io.ctr->pad2_[2], (unsigned int)(T3 - T2)); atomicMax(&io.ctr->band0_next, w);
        }
#endif
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// hits: the reference keeps, per read, the non-no_match results, prepending while it walks the
// clusters (descending position) => final list order is ascending position (worker_bwt.erl:84-97).
// The hit tables.  The device's own format is the COMPACT one (include/bwt_gpu.h): h_rec (16 bytes per hit, ascending by read,
// a read's hits in the reference's list order), one status byte per read, and an `extra` record + CIGAR text only for the hits
// that are not "plain" -- plain = settled on a square window with CIGAR "<Lq>M" (CIG_DIAG), for which window and text follow
// from the record.  The per-hit arrays of bwt_align_batch (window offset / length, CIGAR offsets, all texts) are expanded from
// it on demand (k_hits_expand) -- a batch whose results leave as compact tables never writes them.
// candidates of a read from which on its wave takes the read together (k_hits_count, k_hits_write).  Repeat-rich batch, count + write:
// 128 -> 0.67 + 2.25 ms, 16 -> 0.45 + 1.33, 8 -> 0.44 + 1.43 (a lane's loop over its read's candidates is the wave's time; a read
// taken together costs the wave a round of its own)
#ifndef HITS_FAT
#define HITS_FAT 16u
#endif
// one candidate of a read into the read's totals (the fold of src/worker_bwt.erl:84-97: no_match drops out, the rest is a hit)
__device__ __forceinline__ void hit_count_one(int32_t cs, const uint32_t *__restrict__ c_ciglen, uint64_t k, uint32_t &nh, uint32_t &nx,
                                              uint32_t &xcb, unsigned long long &cbr, int32_t &st, Counters *ctr) {
    if (cs == CS_OK) {
        const uint32_t clf = c_ciglen[k];
        nh++; cbr += clf & ~CIG_DIAG;
        if (!(clf & CIG_DIAG)) { nx++; xcb += clf; }
    } else if (cs == CS_WINDOW_OOR) st = -2;
    else if (cs != CS_NO_MATCH) atomicOr(&ctr->err_flags, cs == CS_CIG_OVERFLOW ? 2ull : 4ull);
}
__global__ __launch_bounds__(256) void k_hits_count(uint32_t n, const uint64_t *__restrict__ clus_off,
                                                     const int32_t *__restrict__ c_status,
                                                     const uint32_t *__restrict__ c_ciglen,
                                                     const uint32_t *__restrict__ r_qty,
                                                     uint32_t *__restrict__ r_nhit, uint32_t *__restrict__ r_nx,
                                                     uint32_t *__restrict__ r_xcb, int8_t *__restrict__ r_status, Counters *ctr,
                                                     uint32_t *__restrict__ r_nn) {
    // r_nn != null: the NARROW form of the table (include/bwt_gpu.h, bwt_batch_fetch_hits_narrow): a read whose answer is exactly one
    // plain hit takes no record in the hit stream -- r_nn[r] = 1, status 2 -- and k_hits_write files its window offset and score
    // in the narrow arrays instead (6 bytes where a record is 16)
    // grid-stride over a capped grid: one atomic per wave for the whole launch (same-address atomics are rated at ~88 per us:
    // one per wave of a 12.5 M-read batch would be 2 ms of them).  Lane per read -- but a read with more than HITS_FAT candidates
    // (low-complexity reads have tens of thousands) is taken by the whole wave, 64 candidates at a time: as one lane's loop it was
    // the kernel's duration (0.98 ms on the repeat-rich batch for 0.3 ms of work).
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long cb = 0;                        // bytes of all CIGAR texts (plain ones included): bwt_batch_counts.cigar_bytes
    for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += gridDim.x * blockDim.x) {   // whole waves iterate together
        const uint32_t r = base + lane;
        const bool valid = r < n;
        uint32_t nh = 0, nx = 0, xcb = 0;
        unsigned long long cbr = 0;
        int32_t st = 1;                                   // sga returned []
        if (valid && (r_qty[r] >> 31)) st = -1;
        const uint64_t a = valid ? clus_off[r] : 0ull, b = valid ? clus_off[r + 1] : 0ull;
        if (b > a) st = 0;
        const bool fat = b - a > HITS_FAT;
        auto one = [&](uint64_t k) { hit_count_one(c_status[k], c_ciglen, k, nh, nx, xcb, cbr, st, ctr); };
        if (!fat) for (uint64_t k = a; k < b; k++) one(k);
        for (unsigned long long fm = __ballot(fat); fm; fm &= fm - 1ull) {
            const int src = __ffsll((long long)fm) - 1;
            const uint64_t fa = (uint64_t)__shfl((long long)a, src, 64), fb = (uint64_t)__shfl((long long)b, src, 64);
            // every lane totals its share of the read's candidates; the read's own lane takes the sums
            uint32_t nh1 = 0, nx1 = 0, xcb1 = 0, oor1 = 0;
            unsigned long long cb1 = 0;
            {
                const uint32_t nh0 = nh, nx0 = nx, xcb0 = xcb; const unsigned long long cbr0 = cbr; const int32_t st0 = st;
                nh = 0; nx = 0; xcb = 0; cbr = 0; st = 0;
                for (uint64_t k = fa + lane; k < fb; k += 64) one(k);
                nh1 = nh; nx1 = nx; xcb1 = xcb; cb1 = cbr; oor1 = st == -2 ? 1u : 0u;
                nh = nh0; nx = nx0; xcb = xcb0; cbr = cbr0; st = st0;
            }
            const unsigned long long tnh = wave_sum_u64(nh1), tnx = wave_sum_u64(nx1), txcb = wave_sum_u64(xcb1), tcb = wave_sum_u64(cb1);
            const bool oor = __ballot(oor1 != 0) != 0;
            if ((int)lane == src) { nh = (uint32_t)tnh; nx = (uint32_t)tnx; xcb = (uint32_t)txcb; cbr = tcb; if (oor) st = -2; }
        }
        if (st == -2) { nh = 0; nx = 0; xcb = 0; cbr = 0; }
        cb += cbr;
        const bool nar = r_nn != nullptr && st == 0 && nh == 1 && nx == 0;
        if (nar) { nh = 0; st = 2; }
        if (valid) { r_nhit[r] = nh; r_nx[r] = nx; r_xcb[r] = xcb; r_status[r] = (int8_t)st; if (r_nn) r_nn[r] = nar ? 1u : 0u; }
    }
    cb = wave_sum_u64(cb);
    if (lane == 0 && cb) atomicAdd(&ctr->cig_total, cb);
}

struct HitRec { uint32_t read; int32_t score; int64_t pos; };   // the 16-byte record gathered over RCCL
struct HitExtra { uint32_t hit, win_len, cig_off, cig_len; };   // bwt_hit_extra

// one hit: its record, and for a hit that is not plain its extra record and text
// (score / S: the candidate's c_score / c_S, loaded by the caller -- with those of its neighbours where it can have them in flight together)
__device__ __forceinline__ void hit_put_v(uint32_t rd, uint64_t k, uint64_t ho1, uint64_t xo1, uint64_t co1, uint32_t wl, uint32_t clf, int32_t shift,
                                      int32_t score, uint32_t S, const uint8_t *__restrict__ cig,
                                      uint32_t cig_stride, HitRec *__restrict__ h_rec, HitExtra *__restrict__ x_rec, uint8_t *__restrict__ x_cig) {
    HitRec hr; hr.read = rd; hr.score = score;
    hr.pos = (int64_t)S - (int64_t)wl + shift;                 // worker_bwt.erl:94
    h_rec[ho1] = hr;
    if (!(clf & CIG_DIAG)) {
        HitExtra x; x.hit = (uint32_t)ho1; x.win_len = wl; x.cig_off = (uint32_t)co1; x.cig_len = clf;
        x_rec[xo1] = x;
        // the text sits right-aligned in its slot (slots are multiples of 32 bytes, 16-byte aligned): it is taken 16 bytes at a
        // time from the slot's end by aligned loads (a byte loop of dependent load / store pairs took 2.9 ms for the 15 M extras
        // of a repeat-rich batch) and stored in pieces of 8 / 4 / 2 / 1 bytes at its unaligned place -- byte by byte the
        // kernel wrote 2.6 GB for 0.85 GB of records and text
        const uint8_t *slot_end = cig + (k + 1) * (uint64_t)cig_stride;
        for (uint32_t done = 0; done < clf; done += 16) {
            const uint4 v = *(const uint4 *)(slot_end - done - 16);
            unsigned long long lo = ((unsigned long long)v.y << 32) | v.x, hi = ((unsigned long long)v.w << 32) | v.z;
            const uint32_t take = clf - done < 16 ? clf - done : 16;
            // the chunk's last `take` bytes go to x_cig[co1 + clf - done - take ..): shifted down to byte 0 first
            const uint32_t sft = 16 - take;
            if (sft >= 8) { lo = hi >> (8 * (sft - 8)); hi = 0; }
            else if (sft) { lo = (lo >> (8 * sft)) | (hi << (64 - 8 * sft)); hi >>= 8 * sft; }
            uint8_t *p = x_cig + co1 + (clf - done - take);
            if (take & 16) { __builtin_memcpy(p, &lo, 8); __builtin_memcpy(p + 8, &hi, 8); }
            if (take & 8) { __builtin_memcpy(p, &lo, 8); p += 8; lo = hi; }
            if (take & 4) { const uint32_t w4 = (uint32_t)lo; __builtin_memcpy(p, &w4, 4); p += 4; lo >>= 32; }
            if (take & 2) { const uint16_t w2 = (uint16_t)lo; __builtin_memcpy(p, &w2, 2); p += 2; lo >>= 16; }
            if (take & 1) *p = (uint8_t)lo;
        }
    }
}

__device__ __forceinline__ void hit_put(uint32_t rd, uint64_t k, uint64_t ho1, uint64_t xo1, uint64_t co1, uint32_t wl, uint32_t clf, int32_t shift,
                                    const int32_t *__restrict__ c_score, const uint32_t *__restrict__ c_S, const uint8_t *__restrict__ cig,
                                    uint32_t cig_stride, HitRec *__restrict__ h_rec, HitExtra *__restrict__ x_rec, uint8_t *__restrict__ x_cig) {
    hit_put_v(rd, k, ho1, xo1, co1, wl, clf, shift, c_score[k], c_S[k], cig, cig_stride, h_rec, x_rec, x_cig);
}

__global__ __launch_bounds__(256) void k_hits_write(uint32_t n, int32_t shift, const uint64_t *__restrict__ clus_off,
                                                     const uint64_t *__restrict__ hit_off, const uint64_t *__restrict__ x_off,
                                                     const uint64_t *__restrict__ xcb_off,
                                                     const int8_t *__restrict__ r_status,
                                                     const int32_t *__restrict__ c_status,
                                                     const int32_t *__restrict__ c_score,
                                                     const uint32_t *__restrict__ c_ciglen,
                                                     const uint32_t *__restrict__ c_S, const uint32_t *__restrict__ c_wlen,
                                                     const uint8_t *__restrict__ cig, uint32_t cig_stride,
                                                     HitRec *__restrict__ h_rec, HitExtra *__restrict__ x_rec,
                                                     uint8_t *__restrict__ x_cig, const uint64_t *__restrict__ nn_off,
                                                     uint32_t *__restrict__ n_woff, int16_t *__restrict__ n_score) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    if (nn_off && r < n && r_status[r] == 2) {
        // the narrow form: this read's one plain hit as {window offset into Ref1 (= Pos - Shift + 200, src/worker_bwt.erl:86-94), score}
        const uint64_t o = nn_off[r];
        for (uint64_t k = clus_off[r]; k < clus_off[r + 1]; k++)
            if (c_status[k] == CS_OK) { n_woff[o] = c_S[k] - c_wlen[k] + BWT_REF_EXT; n_score[o] = (int16_t)c_score[k]; break; }
    }
    const bool live0 = r < n && r_status[r] == 0;
    const uint64_t a = live0 ? clus_off[r] : 0ull, b = live0 ? clus_off[r + 1] : 0ull;
    uint64_t ho = live0 ? hit_off[r] : 0ull;
    const bool live = live0 && ho != hit_off[r + 1];
    uint64_t xo = live ? x_off[r] : 0ull, co = live ? xcb_off[r] : 0ull;
    auto put = [&](uint32_t rd, uint64_t k, uint64_t ho1, uint64_t xo1, uint64_t co1, uint32_t wl, uint32_t clf) {
        hit_put(rd, k, ho1, xo1, co1, wl, clf, shift, c_score, c_S, cig, cig_stride, h_rec, x_rec, x_cig);
    };
    const bool fat = live && b - a > HITS_FAT;
    if (live && !fat)
        for (uint64_t k = b; k-- > a;) {                  // reverse of the cluster order
            if (c_status[k] != CS_OK) continue;
            const uint32_t wl = c_wlen[k], clf = c_ciglen[k];
            put(r, k, ho, xo, co, wl, clf);
            if (!(clf & CIG_DIAG)) { xo++; co += clf; }
            ho++;
        }
    // a read with many candidates: the wave takes it together, 64 candidates at a time in the same (reverse) order -- lane l the
    // l-th of the chunk; where a hit goes follows from the lanes before it (ballots; a scan of the text lengths)
    for (unsigned long long fm = __ballot(fat); fm; fm &= fm - 1ull) {
        const int src = __ffsll((long long)fm) - 1;
        const uint64_t fa = (uint64_t)__shfl((long long)a, src, 64), fb = (uint64_t)__shfl((long long)b, src, 64);
        uint64_t fho = (uint64_t)__shfl((long long)ho, src, 64), fxo = (uint64_t)__shfl((long long)xo, src, 64), fco = (uint64_t)__shfl((long long)co, src, 64);
        const uint32_t fr = (uint32_t)__shfl((int)r, src, 64);
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (uint64_t kend = fb; kend > fa; kend = kend - fa > 64 ? kend - 64 : fa) {
            const bool in = kend - fa > lane;
            const uint64_t k = in ? kend - 1 - lane : fa;
            const bool ok = in && c_status[k] == CS_OK;
            const uint32_t wl = ok ? c_wlen[k] : 0u, clf = ok ? c_ciglen[k] : 0u;
            const bool ex = ok && !(clf & CIG_DIAG);
            const unsigned long long mok = __ballot(ok), mex = __ballot(ex);
            const uint32_t tl = ex ? clf : 0u, incl = dpp_scan_incl_u32(tl);
            if (ok) put(fr, k, fho + (uint64_t)__popcll(mok & lt), fxo + (uint64_t)__popcll(mex & lt), fco + (incl - tl), wl, clf);
            fho += (uint64_t)__popcll(mok); fxo += (uint64_t)__popcll(mex);
            fco += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
}

// ---- the narrow form back into the 16-byte form (on demand: a caller of the per-hit / compact / device-to-device entry points on a
// batch whose table was made narrow).  Lane per read: its records move up by the number of narrow reads before it, a narrow read's
// record is rebuilt; lane per extra: the hit it names moves likewise.
__global__ __launch_bounds__(256) void k_hits_widen(uint32_t n, int32_t shift, const int8_t *__restrict__ st_in, const uint64_t *__restrict__ hit_off,
                                                     const uint64_t *__restrict__ nn_off, const HitRec *__restrict__ h_in,
                                                     const uint32_t *__restrict__ n_woff, const int16_t *__restrict__ n_score,
                                                     HitRec *__restrict__ h_out, uint64_t *__restrict__ hit_off_out, int8_t *__restrict__ st_out) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += gridDim.x * blockDim.x) {
        const uint64_t o = hit_off[r] + nn_off[r];
        hit_off_out[r] = o;
        if (r == n) break;
        const int8_t st = st_in[r];
        st_out[r] = st == 2 ? 0 : st;
        if (st == 2) {
            HitRec h; h.read = r; h.score = n_score[nn_off[r]]; h.pos = (int64_t)n_woff[nn_off[r]] - BWT_REF_EXT + shift;
            h_out[o] = h;
        } else
            for (uint64_t k = hit_off[r]; k < hit_off[r + 1]; k++) h_out[o + (k - hit_off[r])] = h_in[k];
    }
}
__global__ __launch_bounds__(256) void k_extras_widen(uint64_t nx, const HitExtra *__restrict__ x_in, const HitRec *__restrict__ h_in,
                                                       const uint64_t *__restrict__ nn_off, HitExtra *__restrict__ x_out) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nx; k += (uint64_t)gridDim.x * blockDim.x) {
        HitExtra x = x_in[k];
        x.hit += (uint32_t)nn_off[h_in[x.hit].read];
        x_out[k] = x;
    }
}

// ---- expansion of the compact table into the per-hit arrays of bwt_align_batch (on demand)
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }
__global__ __launch_bounds__(256) void k_hits_xidx(const HitExtra *__restrict__ x_rec, uint64_t nx, uint32_t *__restrict__ h_xidx) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nx; k += (uint64_t)gridDim.x * blockDim.x)
        h_xidx[x_rec[k].hit] = (uint32_t)k;
}
__global__ __launch_bounds__(256) void k_hits_len(uint64_t nh, const HitRec *__restrict__ h_rec, const uint32_t *__restrict__ h_xidx,
                                                   const HitExtra *__restrict__ x_rec, const uint64_t *__restrict__ off, uint32_t ulen,
                                                   uint32_t *__restrict__ h_len) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t xi = h_xidx[h];
        if (xi != 0xffffffffu) { h_len[h] = x_rec[xi].cig_len; continue; }
        const uint32_t r = h_rec[h].read;
        const uint32_t lq = ulen ? ulen : (uint32_t)(off[r + 1] - off[r]);
        h_len[h] = dec_digits(lq) + 1;
    }
}
__global__ __launch_bounds__(256) void k_hits_expand(uint64_t nh, int32_t shift, const HitRec *__restrict__ h_rec,
                                                      const uint32_t *__restrict__ h_xidx, const HitExtra *__restrict__ x_rec,
                                                      const uint8_t *__restrict__ x_cig, const uint64_t *__restrict__ off, uint32_t ulen,
                                                      const uint64_t *__restrict__ h_cigoff, int64_t *__restrict__ h_woff,
                                                      int32_t *__restrict__ h_wlen, uint8_t *__restrict__ h_cig) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += (uint64_t)gridDim.x * blockDim.x) {
        const HitRec hr = h_rec[h];
        const uint32_t xi = h_xidx[h];
        const uint64_t co = h_cigoff[h];
        h_woff[h] = hr.pos - shift + BWT_REF_EXT;                  // Start = S - Ref_len + 200, Pos = S - Ref_len + Shift
        if (xi != 0xffffffffu) {
            const HitExtra x = x_rec[xi];
            h_wlen[h] = (int32_t)x.win_len;
            for (uint32_t i = 0; i < x.cig_len; i++) h_cig[co + i] = x_cig[x.cig_off + i];
        } else {
            const uint32_t lq = ulen ? ulen : (uint32_t)(off[hr.read + 1] - off[hr.read]);
            h_wlen[h] = (int32_t)lq;
            const uint32_t cl = dec_digits(lq) + 1;                 // "<Lq>M"
            h_cig[co + cl - 1] = (uint8_t)'M';
            uint32_t v = lq;
            for (uint32_t x = cl - 1; x-- > 0;) { h_cig[co + x] = (uint8_t)('0' + v % 10); v /= 10; }
        }
    }
}

// ---- packed reads (include/bwt_gpu.h): 2-bit codes + exceptions -> the bit-planes (and, on request, the ASCII bytes)
// lane = one 32-base word: 8 bytes of codes in, 16 bytes of planes (+ 32 bytes of ASCII) out
__device__ __forceinline__ uint32_t up_spread4(uint32_t x4) { return (x4 * 0x00204081u) & 0x01010101u; }   // bit i -> bit 8i
__global__ __launch_bounds__(256) void k_unpack_reads(const uint2 *__restrict__ codes, uint64_t nwords, uint64_t total,
                                                       uint32_t *__restrict__ planes, uint8_t *__restrict__ ascii) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint2 c = codes[w];
        if (w == nwords - 1 && (total & 31)) { const uint32_t m = (1u << (total & 31)) - 1u; c.x &= m; c.y &= m; }   // nothing behind the last base
        if (planes) ((uint4 *)planes)[w] = make_uint4(c.x, c.y, 0u, 0u);
        if (ascii) {
            uint32_t o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t L = up_spread4((c.x >> (4 * k)) & 15u), H = up_spread4((c.y >> (4 * k)) & 15u);
                o[k] = 0x41414141u + 2u * L + 6u * H + 0x0Bu * (L & H);        // A 41, C 43, G 47, T 54
            }
            uint4 *dst = (uint4 *)(ascii + w * 32);
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
            dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
}
// the exceptions: is-N / is-other plane bits, code bits cleared, the byte itself into the ASCII stream.  err: bit 0 = a position
// past the end, bit 1 = a byte that is one of ACGT (not an exception)
__global__ __launch_bounds__(256) void k_unpack_exc(const uint64_t *__restrict__ exc, uint64_t n_exc, uint64_t total,
                                                     uint32_t *__restrict__ planes, uint8_t *__restrict__ ascii, unsigned int *__restrict__ err) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_exc; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = exc[k], pos = e & 0x00ffffffffffffffull;
        const uint32_t c = (uint32_t)(e >> 56);
        if (pos >= total) { atomicOr(err, 1u); continue; }
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') { atomicOr(err, 2u); continue; }
        // the list is ascending by position (include/bwt_gpu.h): two records for one base would set both the N and the `other`
        // plane bit and race on the byte
        if (k && (exc[k - 1] & 0x00ffffffffffffffull) >= pos) { atomicOr(err, 4u); continue; }
        const uint64_t w = pos >> 5;
        const uint32_t bit = 1u << (pos & 31);
        atomicAnd(&planes[4 * w + RP_LO], ~bit); atomicAnd(&planes[4 * w + RP_HI], ~bit);
        atomicOr(&planes[4 * w + (c == 'N' ? RP_N : RP_O)], bit);
        ascii[pos] = (uint8_t)c;
    }
}
__global__ __launch_bounds__(256) void k_off_uniform(uint64_t *__restrict__ off, uint32_t n, uint32_t len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) off[i] = i * len;
}
// ASCII bytes of single reads out of the planes (packed batches keep no ASCII copy of their reads: only the kernels of the generic
// DP path read bytes, and they get those of the candidates queued for them).  Eight lanes per listed candidate, 16 bases per
// lane and pass; positions flagged `other` keep the byte k_unpack_exc stored.  Queues longer than `limit` are left alone (the
// consumer of such a queue reads planes).
__global__ __launch_bounds__(256) void k_ascii_reads(const uint32_t *__restrict__ queue, const unsigned int *__restrict__ count, uint32_t limit,
                                                      const uint64_t *__restrict__ c_qoff, const uint32_t *__restrict__ c_qlen,
                                                      const uint32_t *__restrict__ planes, uint8_t *__restrict__ ascii) {
    const uint32_t nq = *count;
    if (nq > limit) return;
    const uint32_t sub = threadIdx.x & 7;
    for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; q < nq; q += (gridDim.x * blockDim.x) >> 3) {
        const uint32_t c = queue[q];
        const uint64_t qo = c_qoff[c];
        const uint32_t lq = c_qlen[c];
        for (uint32_t i = 16 * sub; i < lq; i += 128) {
            const uint4 v = planes4(planes, qo + i);
            const uint32_t n = lq - i < 16 ? lq - i : 16;
            uint8_t *dst = ascii + qo + i;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t L = up_spread4((v.x >> (4 * k)) & 15u), H = up_spread4((v.y >> (4 * k)) & 15u), N = up_spread4((v.z >> (4 * k)) & 15u);
                uint32_t o = 0x41414141u + 2u * L + 6u * H + 0x0Bu * (L & H);
                o = (o & ~(N * 0xffu)) | (N * 0x4Eu);
                const uint32_t oth = (v.w >> (4 * k)) & 15u;
#pragma unroll
                for (uint32_t b = 0; b < 4; b++)
                    if (4 * k + b < n && !((oth >> b) & 1u)) dst[4 * k + b] = (uint8_t)(o >> (8 * b));
            }
        }
    }
}
