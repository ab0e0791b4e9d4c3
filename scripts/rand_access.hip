// Microbenchmark: random reads over a buffer of S bytes -- how the random-access rate of HBM depends on the
// footprint (TLB reach) and on the access shape.  hipcc --offload-arch=gfx950 -O3 -o rand_access rand_access.hip
//   shape 0: quad per 64-byte line, 16 B per lane (k_search's rank-block reads)
//   shape 1: lane per 4-byte word at a random 64-byte line (SA gathers, table reads)
//   shape 2: lane per 8-byte word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
template <int SHAPE>
__global__ __launch_bounds__(256) void k_rand(const uint4 *__restrict__ buf, uint64_t nlines, int iters, uint32_t *out, uint64_t seed) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        if (SHAPE == 0) {
            const uint64_t line = mix((tid >> 2) * 1315423911ull + (uint64_t)i * 0x9E3779B97F4A7C15ull + seed) % nlines;
            const uint4 v = buf[line * 4 + (tid & 3)];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        } else if (SHAPE == 1) {
            const uint64_t h = mix(tid * 1315423911ull + (uint64_t)i * 0x9E3779B97F4A7C15ull + seed);
            acc += ((const uint32_t *)buf)[(h % nlines) * 16 + ((h >> 40) & 15)];
        } else {
            const uint64_t h = mix(tid * 1315423911ull + (uint64_t)i * 0x9E3779B97F4A7C15ull + seed);
            const uint2 v = ((const uint2 *)buf)[(h % nlines) * 8 + ((h >> 40) & 7)];
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char **argv) {
    const double gbs[] = {0.25, 1, 2, 4, 8, 12, 24, 48, 96};
    uint32_t *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double g : gbs) {
        const uint64_t bytes = (uint64_t)(g * (1ull << 30));
        uint4 *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc %.2f GB failed\n", g); break; }
        hipMemset(buf, 1, bytes);
        const uint64_t nlines = bytes / 64;
        for (int shape = 0; shape < 3; shape++) {
            const int iters = 64, blocks = 256 * 8 * 4;
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(k_rand<0>, dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out, (uint64_t)rep);
                else if (shape == 1) hipLaunchKernelGGL(k_rand<1>, dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out, (uint64_t)rep);
                else hipLaunchKernelGGL(k_rand<2>, dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out, (uint64_t)rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double acc = (double)blocks * 256 * iters / (shape == 0 ? 4 : 1);
            printf("footprint %6.2f GB shape %d: %7.2f G lines/s (%.2f TB/s of 64-B lines) %.3f ms\n", g, shape, acc / best / 1e6,
                   acc * 64 / best / 1e9, best);
            fflush(stdout);
        }
        hipFree(buf);
    }
    return 0;
}
