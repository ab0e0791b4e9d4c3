// What a small batch pays outside its arithmetic on this part (profiles/r4/launch_probe.txt): the price list behind the
// small-batch path of libbwtgpu (bwt_small_*).  Build: hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip
//   a  one empty kernel + hipStreamSynchronize
//   b  one empty kernel that raises a flag in pinned host memory, the host spinning on the flag
//   c  a chain of N dependent empty kernels on one stream (+ sync), N = 5, 10, 20, 40
//   d  the same chain as ONE instantiated hipGraph launch
//   e  one kernel of G blocks that passes N grid-wide barriers (atomic counter, every block resident)
//   f  64 KB read from pinned host memory by a kernel / written to it (zero-copy), against hipMemcpyAsync of the same
//   g  a batch's worth of packed reads (312.5 MB) from pinned memory: DMA against a kernel reading it, by grid
//   h  straight-line code run once by one wave: first pass (cold instruction cache) against second
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }

__global__ void k_empty(unsigned int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *p = 1; }
__global__ void k_flag(volatile unsigned int *flag, unsigned int v) {
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void k_barriers(unsigned int *ctr, int nbar, volatile unsigned int *flag, unsigned int v, unsigned int *bail) {
    unsigned int target = 0;
    for (int b = 0; b < nbar; b++) {
        __syncthreads();
        target += gridDim.x;
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(ctr, 1u);
            unsigned int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { atomicOr(bail, 1u); break; }      // never expected: every wave still leaves
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, volatile unsigned int *flag, unsigned int v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    (void)flag; (void)v;
}

// h: a stretch of straight-line code a wave runs through ONCE (what the many small kernels of a small batch mostly are): N dependent
// steps of three instructions, timed inside the kernel on the first pass (instructions come from memory) and on the second (from the instruction cache)
template <int N>
__global__ void k_code(unsigned int *out, unsigned long long *ticks, unsigned int x) {
    unsigned int a = x + threadIdx.x;
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long t0, t1;                     // (the clock reads are tied to the chain's value: nothing moves across them)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a));
#pragma unroll
        for (int i = 0; i < N; i++) a = (a * 1664525u) ^ (a >> 13);          // (not affine: the compiler cannot fold the chain)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a));
        if (threadIdx.x == 0) ticks[blockIdx.x * 2 + rep] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int N>
static void probe_code(hipStream_t s, unsigned int *dflag) {
    unsigned long long *dt = nullptr, ht[2] = {0, 0};
    unsigned int *dout = nullptr;
    CK(hipMalloc(&dt, 64)); CK(hipMalloc(&dout, 4096));
    double first = 0, second = 0;
    const int reps = 20;
    for (int i = 0; i < reps + 3; i++) {
        // (something else runs in between, as the other kernels of a batch do)
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, dflag);
        hipLaunchKernelGGL(k_code<N>, dim3(1), dim3(64), 0, s, dout, dt, (unsigned int)i);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost));
        if (i >= 3) { first += (double)ht[0]; second += (double)ht[1]; }
    }
    printf("h  %5d dependent steps (v_mul_lo, v_lshrrev, v_xor), one wave: first pass %8.2f us, second pass %8.2f us  (%.1f / %.1f ns per step)\n", N,
           first / reps / 100.0, second / reps / 100.0, first / reps * 10.0 / N, second / reps * 10.0 / N);
    CK(hipFree(dt)); CK(hipFree(dout));
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned int *dflag = nullptr, *hflag = nullptr, *dctr = nullptr;
    CK(hipMalloc(&dflag, 64)); CK(hipMalloc(&dctr, 64));
    CK(hipHostMalloc((void **)&hflag, 4096, hipHostMallocDefault));
    memset(hflag, 0, 4096);
    const int reps = 200;
    for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, dflag); CK(hipStreamSynchronize(s)); }
    {   // a
        auto t = clk::now();
        for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, dflag); CK(hipStreamSynchronize(s)); }
        printf("a  empty kernel + hipStreamSynchronize                 %8.2f us\n", us_since(t) / reps);
    }
    {   // b
        auto t = clk::now();
        for (int i = 0; i < reps; i++) {
            const unsigned int v = (unsigned int)i + 1;
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, (volatile unsigned int *)hflag, v);
            while (*(volatile unsigned int *)hflag != v) { }
        }
        printf("b  flag kernel, host spins on pinned flag              %8.2f us\n", us_since(t) / reps);
        CK(hipStreamSynchronize(s));
    }
    for (int n : {5, 10, 20, 40}) {   // c
        auto t = clk::now();
        for (int i = 0; i < reps; i++) {
            for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, dflag);
            CK(hipStreamSynchronize(s));
        }
        printf("c  chain of %2d empty kernels (64 blocks) + sync         %8.2f us  (%.2f per kernel)\n", n, us_since(t) / reps, us_since(t) / reps / n);
    }
    for (int n : {5, 10, 20, 40}) {   // d
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, dflag);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        auto t = clk::now();
        for (int i = 0; i < reps; i++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        printf("d  the same %2d kernels as one hipGraph launch + sync    %8.2f us  (%.2f per kernel)\n", n, us_since(t) / reps, us_since(t) / reps / n);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    unsigned int *dbail = nullptr;
    CK(hipMalloc(&dbail, 64)); CK(hipMemset(dbail, 0, 64));
    for (int G : {16, 64, 256}) for (int nbar : {0, 10, 20}) {   // e
        double tot = 0;
        for (int i = 0; i < reps; i++) {
            CK(hipMemsetAsync(dctr, 0, 4, s));
            CK(hipStreamSynchronize(s));
            const unsigned int v = (unsigned int)(i + 1) | 0x40000000u;
            auto t = clk::now();
            hipLaunchKernelGGL(k_barriers, dim3(G), dim3(256), 0, s, dctr, nbar, (volatile unsigned int *)hflag, v, dbail);
            while (*(volatile unsigned int *)hflag != v) { }
            tot += us_since(t);
        }
        unsigned int bail = 0;
        CK(hipMemcpy(&bail, dbail, 4, hipMemcpyDeviceToHost));
        printf("e  one kernel, %3d blocks, %2d grid barriers, flag wait    %8.2f us%s\n", G, nbar, tot / reps, bail ? "  (BAILED)" : "");
    }
    {   // f
        const size_t bytes = 64 << 10;
        uint8_t *hp = nullptr, *dp = nullptr;
        CK(hipHostMalloc((void **)&hp, bytes, hipHostMallocDefault)); CK(hipMalloc(&dp, bytes));
        memset(hp, 1, bytes);
        auto t = clk::now();
        for (int i = 0; i < reps; i++) { CK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        printf("f  hipMemcpyAsync 64 KB pinned -> device + sync         %8.2f us\n", us_since(t) / reps);
        t = clk::now();
        for (int i = 0; i < reps; i++) { CK(hipMemcpyAsync(hp, dp, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        printf("f  hipMemcpyAsync 64 KB device -> pinned + sync         %8.2f us\n", us_since(t) / reps);
        t = clk::now();
        for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, s, (const uint4 *)hp, (uint4 *)dp, bytes / 16, (volatile unsigned int *)hflag, 0u); CK(hipStreamSynchronize(s)); }
        printf("f  kernel reads 64 KB of pinned host memory + sync      %8.2f us\n", us_since(t) / reps);
        t = clk::now();
        for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, s, (const uint4 *)dp, (uint4 *)hp, bytes / 16, (volatile unsigned int *)hflag, 0u); CK(hipStreamSynchronize(s)); }
        printf("f  kernel writes 64 KB to pinned host memory + sync     %8.2f us\n", us_since(t) / reps);
        std::vector<uint8_t> pg(bytes, 2);
        t = clk::now();
        for (int i = 0; i < reps; i++) { CK(hipMemcpyAsync(dp, pg.data(), bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        printf("f  hipMemcpyAsync 64 KB PAGEABLE -> device + sync       %8.2f us\n", us_since(t) / reps);
        t = clk::now();
        for (int i = 0; i < reps; i++) { CK(hipMemcpyAsync(pg.data(), dp, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        printf("f  hipMemcpyAsync 64 KB device -> PAGEABLE + sync       %8.2f us\n", us_since(t) / reps);
    }
    {   // g: a batch's worth of packed reads (312.5 MB) from pinned host memory: DMA against a kernel reading it (zero-copy), by grid
        const size_t bytes = 312500000ull / 16 * 16;
        uint8_t *hp = nullptr, *dp = nullptr;
        CK(hipHostMalloc((void **)&hp, bytes, hipHostMallocDefault)); CK(hipMalloc(&dp, bytes));
        memset(hp, 3, bytes);
        for (int i = 0; i < 2; i++) { CK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        auto t = clk::now();
        for (int i = 0; i < 5; i++) { CK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        double us = us_since(t) / 5;
        printf("g  hipMemcpyAsync 312.5 MB pinned -> device             %8.0f us  (%.1f GB/s)\n", us, bytes / us / 1e3);
        for (int G : {16, 64, 256, 1024, 4096}) {
            hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, s, (const uint4 *)hp, (uint4 *)dp, bytes / 16, (volatile unsigned int *)hflag, 0u);
            CK(hipStreamSynchronize(s));
            t = clk::now();
            for (int i = 0; i < 5; i++) { hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, s, (const uint4 *)hp, (uint4 *)dp, bytes / 16, (volatile unsigned int *)hflag, 0u); CK(hipStreamSynchronize(s)); }
            us = us_since(t) / 5;
            printf("g  kernel reads 312.5 MB of pinned memory, %4d blocks   %8.0f us  (%.1f GB/s)\n", G, us, bytes / us / 1e3);
        }
    }
    probe_code<256>(s, dflag); probe_code<1024>(s, dflag); probe_code<4096>(s, dflag);
    return 0;
}
