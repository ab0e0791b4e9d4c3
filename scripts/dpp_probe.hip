// Probe of the cross-lane pieces k_seed_fast relies on (DPP wave scan, wave_shr:1, the 3-way sort, sf_emit), on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -o scripts/dpp_probe scripts/dpp_probe.hip && ./scripts/dpp_probe
#include "../bwt_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t *in, uint32_t *out, uint32_t Pj, uint32_t qj) {
    const uint32_t lane = threadIdx.x;
    uint32_t a = lane < Pj ? in[lane] : 0xffffffffu, b = in[64 + lane], c = in[128 + lane];
    { uint32_t v[3] = {a, b, c}; mirror_sort64(v, sort_masks()); a = v[0]; b = v[1]; c = v[2]; }
    out[640 + lane] = b; out[704 + lane] = c;
    out[lane] = a;
    uint32_t cv0 = 0xffffffffu, cv1 = 0;
    const uint32_t Kc = sf_emit(a, Pj, qj, 1, cv0, cv1);
    out[64 + lane] = cv0; out[128 + lane] = cv1; out[192 + lane] = Kc;
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xF, 0xF, false);
    out[256 + lane] = prev;
    const unsigned long long sm = __ballot(a - prev > BWT_TOLERANCE);
    out[320 + lane] = (uint32_t)((sm >> lane) & 1);
    out[384 + lane] = dpp_scan_incl_u32(in[lane] & 15u);
    uint32_t d1, d2;
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(d1) : "v"(a));
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(d2) : "v"(a));
    out[448 + lane] = d1; out[512 + lane] = d2;
    // the masks of sf_emit, step by step
    const unsigned long long vmask = Pj >= 64 ? ~0ull : ((1ull << Pj) - 1ull);
    const unsigned long long sm2 = (__ballot(d1 > BWT_TOLERANCE) | 1ull) & vmask;
    const unsigned long long A = ~sm2;
    unsigned long long ok = A;
    const uint32_t nthr = 2 + qj / 7;
    for (uint32_t k = 1; k + 1 < nthr; k++) ok &= A << k;
    const unsigned long long em = ((sm2 >> 1) | (1ull << (Pj - 1))) & vmask & ok;
    if (lane == 0) { out[576] = (uint32_t)sm2; out[577] = (uint32_t)ok; out[578] = (uint32_t)em; out[579] = nthr; }
}
int main() {
    const uint32_t s1[21] = {0x1aba34, 0x2d6f12, 0x1ee45f, 0x229c63, 0x256c18, 0xca0f0, 0x2591fa, 0x1bd1d5, 0x3dc20, 0x9505, 0x10c99, 0x3dc20, 0x3dc21, 0x3dc20,
                             0x14af7a, 0x181a46, 0x1c8685, 0x246109, 0x299ce6, 0x2b4234, 0x2ccb26};
    uint32_t h[192], o[768], *din, *dout;
    for (int i = 0; i < 192; i++) h[i] = 0x12345u * (uint32_t)(i + 7) % 3000017u;
    for (int i = 0; i < 21; i++) h[i] = s1[i];
    (void)hipMalloc(&din, sizeof h); (void)hipMalloc(&dout, sizeof o);
    (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, 21u, 12u);
    (void)hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    const char *nm[] = {"sorted", "cv0", "cv1", "Kc", "prev", "start", "scan", "subrev_dpp", "mov_dpp_bc"};
    printf("sm %x ok %x em %x nthr %u\n", o[576], o[577], o[578], o[579]);
    for (int k = 0; k < 9; k++) { printf("%s:", nm[k]); for (int i = 0; i < 24; i++) printf(" %x", o[64 * k + i]); printf("\n"); }
    {   // the three sorted registers against std::sort
        int bads = 0;
        for (int k = 0; k < 3; k++) {
            uint32_t ref[64];
            for (int i = 0; i < 64; i++) ref[i] = (k == 0 && i >= 21) ? 0xffffffffu : h[64 * k + i];
            for (int i = 0; i < 64; i++) for (int j = i + 1; j < 64; j++) if (ref[j] < ref[i]) { uint32_t t = ref[i]; ref[i] = ref[j]; ref[j] = t; }
            const uint32_t *got = k == 0 ? o : o + 640 + 64 * (k - 1);
            for (int i = 0; i < 64; i++) bads += got[i] != ref[i];
        }
        printf("mirror_sort64 mismatches %d\n", bads);
    }
    uint32_t run = 0; int bad = 0;
    for (int i = 0; i < 64; i++) { run += h[i] & 15u; bad += o[384 + i] != run; }
    printf("scan mismatches %d; expected cluster (3dc21, 1) in lane 8\n", bad);
    return 0;
}
