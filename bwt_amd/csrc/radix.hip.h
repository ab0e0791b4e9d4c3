// Hand-written device primitives of the index builder: a stable LSD radix sort of (u64 key, u32 value)
// pairs and u32 scans (inclusive max, exclusive sum).  wave64 throughout.
//
// Sort, one pass per 8-bit digit, three kernels:
//   kr_hist     every block counts the digits of its tile (LDS atomics) -> hist[digit][block]
//   kr_scan*    exclusive sum over the digit-major table: (digit, block) -> first output slot
//   kr_scatter  the block ranks its tile stably -- per round of 256 elements each wave finds, with 8 ballots,
//               the lanes that hold the same digit (rank inside the wave = popcount of the lower peers), the
//               waves are chained through a per-wave count table in LDS -- and writes key and value to
//               base[digit][block] + rank.
// Tiles are RS_TILE consecutive elements, element order inside a tile = round*256 + thread: stable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)

__global__ __launch_bounds__(RS_THREADS) static void kr_hist(const uint64_t *__restrict__ keys, uint64_t n, int shift,
                                                              uint32_t *__restrict__ hist, uint32_t nblk) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint64_t e = base + (uint64_t)i * RS_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) static void kr_scatter(const uint64_t *__restrict__ keys,
                                                                 const uint32_t *__restrict__ vals, uint64_t n, int shift,
                                                                 const uint32_t *__restrict__ base, uint32_t nblk,
                                                                 uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t running[256];          // next free slot of every digit for this block
    __shared__ uint32_t wavecnt[4][256];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    running[threadIdx.x] = base[(uint64_t)threadIdx.x * nblk + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) wavecnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint64_t tile = (uint64_t)blockIdx.x * RS_TILE;
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint64_t e = tile + (uint64_t)i * RS_THREADS + threadIdx.x;
        const bool valid = e < n;
        uint64_t k = 0; uint32_t v = 0, d = 0;
        if (valid) { k = keys[e]; v = vals[e]; d = (uint32_t)(k >> shift) & 0xffu; }
        // lanes of this wave holding the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wavecnt[wv][d] = (uint32_t)__popcll(peers);      // the lowest peer reports the count
        __syncthreads();
        if (valid) {
            uint32_t o = running[d] + rank;
            for (uint32_t w = 0; w < wv; w++) o += wavecnt[w][d];
            keys_out[o] = k; vals_out[o] = v;
        }
        __syncthreads();
        {
            const uint32_t t = threadIdx.x;
            running[t] += wavecnt[0][t] + wavecnt[1][t] + wavecnt[2][t] + wavecnt[3][t];
            wavecnt[0][t] = 0; wavecnt[1][t] = 0; wavecnt[2][t] = 0; wavecnt[3][t] = 0;
        }
        __syncthreads();
    }
}

// ---- u32 scans (out may alias in).  MAXOP: running maximum, else sum.  EXCL: exclusive, else inclusive.
#define SC_ITEMS 8
#define SC_TILE (256 * SC_ITEMS)
template <bool MAXOP>
__device__ __forceinline__ uint32_t sc_op(uint32_t a, uint32_t b) { return MAXOP ? (a > b ? a : b) : a + b; }

template <bool MAXOP, bool EXCL>
__global__ __launch_bounds__(256) static void kr_scan_tile(const uint32_t *in, uint32_t *out,     // may alias
                                                            uint32_t *__restrict__ bsum, uint64_t n) {
    __shared__ uint32_t sh[256];
    const uint64_t b0 = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) { v[k] = b0 + k < n ? in[b0 + k] : 0u; s = sc_op<MAXOP>(s, v[k]); }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] = sc_op<MAXOP>(sh[threadIdx.x], t);
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? sh[threadIdx.x - 1] : 0u;       // everything before this thread (0 is neutral for both)
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        const uint32_t incl = sc_op<MAXOP>(run, v[k]);
        if (b0 + k < n) out[b0 + k] = EXCL ? run : incl;
        run = incl;
    }
    if (threadIdx.x == 255) bsum[blockIdx.x] = sh[255];
}
// single block: EXCLUSIVE scan of up to 65536 tile totals in place
template <bool MAXOP>
__global__ __launch_bounds__(256) static void kr_scan_sums(uint32_t *bsum, uint32_t nb) {
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] = sc_op<MAXOP>(sh[threadIdx.x], t);
            __syncthreads();
        }
        const uint32_t before = threadIdx.x ? sh[threadIdx.x - 1] : 0u;
        if (i < nb) bsum[i] = sc_op<MAXOP>(carry, before);
        __syncthreads();
        if (threadIdx.x == 255) carry = sc_op<MAXOP>(carry, sh[255]);
        __syncthreads();
    }
}
template <bool MAXOP>
__global__ __launch_bounds__(256) static void kr_scan_apply(uint32_t *__restrict__ out, const uint32_t *__restrict__ bsum,
                                                             uint64_t n) {
    const uint64_t b0 = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_ITEMS;
    const uint32_t add = bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++)
        if (b0 + k < n) out[b0 + k] = sc_op<MAXOP>(out[b0 + k], add);
}

static inline uint64_t dev_scan_u32_tmp_words(uint64_t n) {
    const uint64_t nb = (n + SC_TILE - 1) / SC_TILE;
    return nb + (nb + SC_TILE - 1) / SC_TILE + 8;
}
// MAXOP: out[i] = max(in[0..i]) ; else out[i] = sum(in[0..i-1]).  tmp: dev_scan_u32_tmp_words(n) u32.
template <bool MAXOP>
static hipError_t dev_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *tmp, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t nb = (uint32_t)((n + SC_TILE - 1) / SC_TILE);
    hipLaunchKernelGGL((kr_scan_tile<MAXOP, !MAXOP>), dim3(nb), dim3(256), 0, s, in, out, tmp, n);
    if (nb > 1) {
        if (nb <= 65536) {
            hipLaunchKernelGGL(kr_scan_sums<MAXOP>, dim3(1), dim3(256), 0, s, tmp, nb);
        } else {                                     // the totals themselves need a tiled exclusive scan
            uint32_t *tmp2 = tmp + nb;
            const uint32_t nb2 = (nb + SC_TILE - 1) / SC_TILE;
            hipLaunchKernelGGL((kr_scan_tile<MAXOP, true>), dim3(nb2), dim3(256), 0, s, tmp, tmp, tmp2, (uint64_t)nb);
            hipLaunchKernelGGL(kr_scan_sums<MAXOP>, dim3(1), dim3(256), 0, s, tmp2, nb2);
            hipLaunchKernelGGL(kr_scan_apply<MAXOP>, dim3(nb2), dim3(256), 0, s, tmp, tmp2, (uint64_t)nb);
        }
        hipLaunchKernelGGL(kr_scan_apply<MAXOP>, dim3(nb), dim3(256), 0, s, out, tmp, n);
    }
    return hipGetLastError();
}

// Stable LSD radix sort of (key, value) pairs on key bits [bit_lo, bit_hi).  k[*cur]/v[*cur] hold the input and,
// on return, the result (*cur flips once per 8-bit pass).  hist: 256 * ceil(n / RS_TILE) u32; tmp: dev_scan_u32_tmp_words(256 * tiles) u32.
static hipError_t dev_radix_sort_pairs(uint64_t *k[2], uint32_t *v[2], int *cur, uint64_t n, int bit_lo, int bit_hi, uint32_t *hist,
                                       uint32_t *tmp, hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        const int c = *cur;
        hipLaunchKernelGGL(kr_hist, dim3(nblk), dim3(RS_THREADS), 0, s, k[c], n, shift, hist, nblk);
        hipError_t e = dev_scan_u32<false>(hist, hist, (uint64_t)256 * nblk, tmp, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kr_scatter, dim3(nblk), dim3(RS_THREADS), 0, s, k[c], v[c], n, shift, hist, nblk, k[c ^ 1], v[c ^ 1]);
        *cur = c ^ 1;
    }
    return hipGetLastError();
}
