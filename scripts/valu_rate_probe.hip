// Issue rate of the VALU instructions the DP kernels are built from, per SIMD, on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -o scripts/valu_rate_probe scripts/valu_rate_probe.hip && ./scripts/valu_rate_probe
// Every wave runs ITER x 32 independent instances of one instruction (8 accumulators, so no dependency stalls);
// 4 waves per SIMD on every SIMD.  Prints wave-instructions per SIMD per microsecond: 4-cycle issue at 2.4 GHz is 600.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(INS)                                                                                   \
    uint32_t a[8];                                                                                  \
    for (int t = 0; t < 8; t++) a[t] = in[threadIdx.x + 64 * t];                                    \
    const uint32_t k1 = in[600], k2 = in[601];                                                      \
    for (int it = 0; it < ITER; it++) {                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                             \
            _Pragma("unroll") for (int t = 0; t < 8; t++) asm volatile(INS : "+v"(a[t]) : "v"(k1), "v"(k2)); \
        }                                                                                           \
    }                                                                                               \
    uint32_t s = 0;                                                                                 \
    for (int t = 0; t < 8; t++) s ^= a[t];                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
__global__ void p_add(const uint32_t *in, uint32_t *out) { BODY("v_add_u32 %0, %0, %1") }
__global__ void p_and(const uint32_t *in, uint32_t *out) { BODY("v_and_b32 %0, %0, %1") }
__global__ void p_lshl_add(const uint32_t *in, uint32_t *out) { BODY("v_lshl_add_u32 %0, %0, 3, %1") }
__global__ void p_mad24(const uint32_t *in, uint32_t *out) { BODY("v_mad_u32_u24 %0, %0, %1, %2") }
__global__ void p_max3(const uint32_t *in, uint32_t *out) { BODY("v_max3_i32 %0, %0, %1, %2") }
__global__ void p_add3(const uint32_t *in, uint32_t *out) { BODY("v_add3_u32 %0, %0, %1, %2") }
__global__ void p_pk_add(const uint32_t *in, uint32_t *out) { BODY("v_pk_add_u16 %0, %0, %1") }
__global__ void p_pk_mad(const uint32_t *in, uint32_t *out) { BODY("v_pk_mad_i16 %0, %0, %1, %2") }
__global__ void p_pk_max(const uint32_t *in, uint32_t *out) { BODY("v_pk_max_i16 %0, %0, %1") }
__global__ void p_pk_maxu(const uint32_t *in, uint32_t *out) { BODY("v_pk_max_u16 %0, %0, %1") }
__global__ void p_bfe(const uint32_t *in, uint32_t *out) { BODY("v_bfe_u32 %0, %0, %1, 1") }
__global__ void p_perm(const uint32_t *in, uint32_t *out) { BODY("v_perm_b32 %0, %0, %1, %2") }
__global__ void p_cndmask(const uint32_t *in, uint32_t *out) { BODY("v_cndmask_b32 %0, %0, %1, vcc") }
__global__ void p_max(const uint32_t *in, uint32_t *out) { BODY("v_max_i32 %0, %0, %1") }
int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t h[1024], *din, *dout;
    for (int i = 0; i < 1024; i++) h[i] = 0x9e3779b9u * (uint32_t)(i + 1);
    const int blocks = cus * 4;                          // 4 blocks of 256 per CU = 4 waves per SIMD
    (void)hipMalloc(&din, sizeof h); (void)hipMalloc(&dout, (size_t)blocks * 256 * 4);
    (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; void (*k)(const uint32_t *, uint32_t *); } ks[] = {
        {"v_add_u32", p_add}, {"v_and_b32", p_and}, {"v_lshl_add_u32", p_lshl_add}, {"v_mad_u32_u24", p_mad24}, {"v_max_i32", p_max},
        {"v_max3_i32", p_max3}, {"v_add3_u32", p_add3}, {"v_bfe_u32", p_bfe}, {"v_perm_b32", p_perm}, {"v_cndmask_b32", p_cndmask},
        {"v_pk_add_u16", p_pk_add}, {"v_pk_mad_i16", p_pk_mad}, {"v_pk_max_i16", p_pk_max}, {"v_pk_max_u16", p_pk_maxu}};
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, din, dout);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, din, dout);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = 4.0 * ITER * 32;         // wave-instructions of the 4 waves of a SIMD
        printf("%-16s %8.3f ms  %7.1f wave-instr / SIMD / us\n", k.name, ms, per_simd / (ms * 1000.0));
    }
    return 0;
}
