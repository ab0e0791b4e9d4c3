// libbwtgpu: host side of the C ABI declared in include/bwt_gpu.h (HIP runtime only; no torch types).
#include "common.h"
#include "kernels.hip.h"
#include "etf_fm.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

static thread_local std::string g_err;
extern "C" const char *bwt_last_error(void) { return g_err.c_str(); }
int bwt_fail(int code, const std::string &msg) { g_err = msg; return code; }

extern "C" void bwt_options_init(bwt_options *o) {
    if (!o) return;
    memset(o, 0xff, sizeof *o);                       // every field -1: the library's own choice
    o->struct_size = (uint32_t)sizeof *o;
}
static int options_copy(bwt_options &dst, const bwt_options *src, const char *who) {
    if (!src) return fail(BWT_EINVAL, std::string(who) + ": null argument");
    if (src->struct_size != sizeof(bwt_options)) return fail(BWT_EINVAL, std::string(who) + ": bwt_options.struct_size does not match this library (call bwt_options_init)");
    dst = *src;
    return 0;
}
extern "C" int bwt_index_set_options(bwt_index *ix, const bwt_options *o) {
    if (!ix) return fail(BWT_EINVAL, "bwt_index_set_options: null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    return options_copy(ix->opt, o, "bwt_index_set_options");
}

extern "C" void bwt_constants(bwt_constants_t *o) {
    o->tolerance = BWT_TOLERANCE; o->min_len = BWT_MIN_LEN; o->max_range = BWT_MAX_RANGE; o->max_len = BWT_MAX_LEN;
    o->match = 2; o->mismatch = -1; o->gap_penalty = -2; o->gap_ext_penalty = -1; o->unknown = 1.6;
    o->ref_extension_len = BWT_REF_EXT;
}

// ------------------------------------------------------------------------------------------------
// bit-planes of a resident byte stream (k_read_planes); the words past the last base are zeroed so that windows may run on
static int launch_planes(const uint8_t *bases, uint64_t total, uint32_t *planes, int device, hipStream_t stream) {
    const uint64_t words = rp_words(total), used = ((total + 31) / 32) * 4;
    HIPCHK(hipMemsetAsync(planes + used, 0, (words - used) * 4, stream));
    if (total) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total / 16 + 256) / 256, (uint64_t)cus * 16));
        hipLaunchKernelGGL(k_read_planes, dim3(grid), dim3(256), 0, stream, bases, total, planes);
    }
    return 0;
}

int bwt_index_make_refplanes(bwt_index *ix, hipStream_t stream) {
    int rc = ix->refplanes.exact(rp_words(ix->ref1_len) * 4);
    if (rc) return rc;
    if ((rc = launch_planes(ix->ref1.as<uint8_t>(), ix->ref1_len, ix->refplanes.as<uint32_t>(), ix->device, stream))) return rc;
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    ix->d.rplanes = ix->refplanes.as<uint32_t>();
    ix->d.rpw = rp_words(ix->ref1_len);
    return 0;
}

// Suffix-interval table (kernels.hip.h: kt_build), levels 8..K.  K = 0 drops it; K < 0 picks the default: the
// K in 8..15 with 4^K closest above the row count (deeper levels would mostly
// repeat "not found") -- raised to 15 = MAX_LEN, where the table answers every task, once that K is >= 13 (more than
// 16.7 M rows) and the 11.5 GB leave >= 32 GB of HBM free.  GRCh38: K = 15.
// wide: the top level as 64-byte entries that carry the first 14 SA values of the range (a task that ends there needs
// no SA read, and the seed kernel clusters such reads itself); only useful at K = 15, where every full-length task
// ends.  < 0: on when K = 15 and the 64 * 4^15 = 68.7 GB leave >= 32 GB of HBM free.
extern "C" int bwt_index_set_ktable(bwt_index *ix, int K, int wide) { return bwt_index_set_ktable_ex(ix, K, wide, 1); }
extern "C" int bwt_index_set_ktable_ex(bwt_index *ix, int K, int wide, int summary) {
    if (!ix) return fail(BWT_EINVAL, "bwt_index_set_ktable: null index");
    if (K < 0) {
        {
            K = KT_MIN;
            while (K < BWT_MAX_LEN && (1ull << (2 * K)) < (uint64_t)ix->Last) K++;
            size_t free_b = 0, total_b = 0;
            (void)hipSetDevice(ix->device);
            if (K >= 13 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= kt_entries(BWT_MAX_LEN) * 8 + (32ull << 30))
                K = BWT_MAX_LEN;
        }
    }
    if (K != 0 && (K < KT_MIN || K > BWT_MAX_LEN)) return fail(BWT_EINVAL, "bwt_index_set_ktable: K must be 0 or 8..15");
    std::unique_lock<std::shared_mutex> lk(ix->mu);        // waits for running batches; none starts while the table is swapped
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->d.ktab = nullptr; ix->d.ktw = nullptr; ix->d.kt_k = 0; ix->d.ktsum = nullptr; ix->d.ktsum_levels = 0;
    for (int k = 0; k < 8; k++) ix->d.ktw_off[k] = KTW_NONE;
    ix->ktw_bytes = 0;
    ix->ktab.release(); ix->ktw.release(); ix->ktsum.release();
    if (K == 0) return 0;
    const uint64_t wide_bytes = 64ull << (2 * K);
    if (wide < 0) {
        {
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            wide = (K == BWT_MAX_LEN && free_b >= wide_bytes + (32ull << 30)) ? 2 : 0;
        }
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device);
    // the 8-byte levels first (every level but a wide top one); then, wide == 2, which of the LOWER levels are worth keeping as
    // 64-byte entries as well: those where at least a quarter of the keys give a task that ends there a seed list -- on a
    // chromosome-sized index the 12-mers (the last sub-read of a 100-bp read: its ~14 seeds were a dozen random SA reads per
    // read, more lines than the eleven top-level lookups together), on a genome-sized one the 14-mers (150-bp reads) -- and
    // that fit next to a 32 GB reserve
    int rc = ix->ktab.exact((wide ? kt_level_off(K) : kt_entries(K)) * 8);
    if (rc) return rc;
    for (int m = KT_MIN; m <= K; m++) {
        if (wide && m == K) break;
        const uint64_t quads = 1ull << (2 * m);
        const uint32_t grid = (uint32_t)std::min<uint64_t>((quads * 4 + 255) / 256, (uint64_t)cus * 8);
        hipLaunchKernelGGL(kt_build, dim3(grid), dim3(256), 0, ix->stream, ix->d, ix->ktab.as<uint64_t>(), m, (uint4 *)nullptr);
    }
    uint64_t off_entries[8];
    for (int k = 0; k < 8; k++) off_entries[k] = KTW_NONE;
    uint64_t total_entries = wide ? (1ull << (2 * K)) : 0;
    if (wide == 2 && K == BWT_MAX_LEN) {
        DevBuf cnt;
        if ((rc = cnt.ensure(16 * 8))) { ix->ktab.release(); return rc; }
        HIPCHK(hipMemsetAsync(cnt.p, 0, 16 * 8, ix->stream));
        for (int m = KT_MIN + 1; m < K; m++)
            hipLaunchKernelGGL(kt_count_seeded, dim3((uint32_t)std::min<uint64_t>(((1ull << (2 * m)) + 255) / 256, (uint64_t)cus * 8)), dim3(256), 0, ix->stream,
                               (const uint64_t *)ix->ktab.as<uint64_t>(), m, cnt.as<unsigned long long>());
        unsigned long long hc[16] = {0};
        hipError_t e = hipMemcpyAsync(hc, cnt.p, sizeof hc, hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        cnt.release();
        if (e != hipSuccess) { ix->ktab.release(); return fail(BWT_EHIP, std::string("suffix-interval table: ") + hipGetErrorString(e)); }
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = free_b > wide_bytes + (32ull << 30) ? free_b - wide_bytes - (32ull << 30) : 0;
        for (int m = K - 1; m > KT_MIN; m--) {
            const uint64_t keys = 1ull << (2 * m);
            if (hc[m] * 4 >= keys && keys * 64 <= budget && total_entries + keys < 0xfffffff0ull) {
                off_entries[m - 8] = total_entries; total_entries += keys; budget -= keys * 64;
            }
        }
    }
    if (wide) {
        off_entries[K - 8] = 0;
        if ((rc = ix->ktw.exact(total_entries * 64))) { ix->ktab.release(); return rc; }
        ix->ktw_bytes = total_entries * 64;
        const uint64_t quads = 1ull << (2 * K);
        hipLaunchKernelGGL(kt_build, dim3((uint32_t)std::min<uint64_t>((quads * 4 + 255) / 256, (uint64_t)cus * 8)), dim3(256), 0, ix->stream, ix->d,
                           ix->ktab.as<uint64_t>(), K, ix->ktw.as<uint4>());
        for (int m = KT_MIN; m < K; m++)
            if (off_entries[m - 8] != KTW_NONE)
                hipLaunchKernelGGL(kt_widen, dim3((uint32_t)std::min<uint64_t>(((1ull << (2 * m)) * 4 + 255) / 256, (uint64_t)cus * 8)), dim3(256), 0, ix->stream, ix->d,
                                   (const uint64_t *)ix->ktab.as<uint64_t>(), m, ix->ktw.as<uint4>() + off_entries[m - 8] * 4);
    }
    HIPCHK(hipStreamSynchronize(ix->stream));
    HIPCHK(hipGetLastError());
    ix->d.ktab = ix->ktab.as<uint64_t>(); ix->d.kt_k = K;
    ix->d.ktw = wide ? ix->ktw.as<uint4>() : nullptr;
    if (wide) for (int k = 0; k < 8; k++) ix->d.ktw_off[k] = (uint32_t)off_entries[k];
    // Summary of the 8-byte levels (one bit per 64 entries: "not all plain too-many-seeds entries"): on a large genome every
    // 12-mer occurs far more than MAX_RANGE times, and the lookup of a read's short last sub-read is answered by a cached
    // bit instead of a random table line.  Used per level only where fewer than a quarter of the bits are set.
    const int top8 = wide ? K - 1 : K;                            // highest level kept as 8-byte entries
    if (top8 >= KT_MIN && summary != 0) {
        const uint64_t words = (kt_level_off(top8 + 1) >> 11) + 2;
        if ((rc = ix->ktsum.exact(words * 4 + 64 * 4))) return rc;
        HIPCHK(hipMemsetAsync(ix->ktsum.p, 0, words * 4 + 64 * 4, ix->stream));
        unsigned int *nset = ix->ktsum.as<unsigned int>() + words;                       // 16 counters behind the bitmap
        for (int m = KT_MIN; m <= top8; m++) {
            const uint64_t groups = (1ull << (2 * m)) >> 6;
            const uint32_t grid = (uint32_t)std::min<uint64_t>((groups * 64 + 255) / 256, (uint64_t)cus * 8);
            hipLaunchKernelGGL(k_kt_summary, dim3(grid), dim3(256), 0, ix->stream, (const uint64_t *)ix->ktab.as<uint64_t>(), m,
                               ix->ktsum.as<uint32_t>(), nset);
        }
        unsigned int hset[16] = {0};
        HIPCHK(hipMemcpyAsync(hset, nset, sizeof hset, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        HIPCHK(hipGetLastError());
        uint32_t levels = 0;
        for (int m = KT_MIN; m <= top8; m++)
            if ((uint64_t)hset[m] * 4 < ((1ull << (2 * m)) >> 6) || summary == 2) levels |= 1u << m;    // (2: every level, for tests)
        if (levels) { ix->d.ktsum = ix->ktsum.as<uint32_t>(); ix->d.ktsum_levels = levels; }
        else ix->ktsum.release();
    }
    return 0;
}
extern "C" int bwt_index_ktable(const bwt_index *ix) {
    if (!ix) return 0;
    int v = ix->d.kt_k | (ix->d.ktw ? 16 : 0);
    if (ix->d.ktw)
        for (int m = KT_MIN; m < ix->d.kt_k; m++) if (ix->d.ktw_off[m - 8] != KTW_NONE) v |= 1 << (m - 8 + 8);     // bit 8 + (m - 8): level m widened
    return v;
}

static int index_build(const FmHost &fm, const uint8_t *ref, uint64_t ref_len, int device, bwt_index **out) {
    const uint32_t n = (uint32_t)fm.L.size();
    if (n < 8 || fm.Last != n) return fail(BWT_EINVAL, "index: Last must equal the number of rows");
    if (!(1 < fm.Pc && fm.Pc < fm.Pg && fm.Pg < fm.Pt && fm.Pt <= fm.Last))
        return fail(BWT_EINVAL, "index: every base must occur (need 1 < Pc < Pg < Pt <= Last)");
    if ((uint64_t)n + 70000ull >= 0xffffffffull) return fail(BWT_EINVAL, "index: too many rows for u32 positions");
    if (ref_len + 1 != n) return fail(BWT_EINVAL, "index: .ref length must be rows-1");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "index: no such HIP device");
    HIPCHK(hipSetDevice(device));

    // 64-byte rank blocks (layout: kernels_index.h)
    const uint32_t nblk = n / BWT_BLOCK_ROWS + 1;
    std::vector<uint32_t> blk((size_t)nblk * 16, 0);
    uint32_t cnt[4] = {0, 0, 0, 0}, dollar = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t *w = &blk[(size_t)b * 16];
        w[0] = cnt[0]; w[4] = cnt[1]; w[8] = cnt[2]; w[12] = cnt[3];
        for (uint32_t i = 0; i < BWT_BLOCK_ROWS; i++) {
            const uint64_t row0 = (uint64_t)b * BWT_BLOCK_ROWS + i;     // 0-based row
            if (row0 >= n) break;
            const uint8_t c = fm.L[row0];
            uint32_t code;
            switch (c) {
            case 'A': code = 0; break;
            case 'C': code = 1; break;
            case 'G': code = 2; break;
            case 'T': code = 3; break;
            case '$':
                if (dollar) return fail(BWT_EINVAL, "index: more than one $ in L");
                dollar = (uint32_t)row0 + 1; code = 0; break;
            default: return fail(BWT_EINVAL, "index: L outside $ACGT");
            }
            if (c != '$') cnt[code]++;
            blk_set_code(w, i, code);
        }
    }
    if (!dollar) return fail(BWT_EINVAL, "index: no $ in L");
    if (2 + cnt[0] != fm.Pc || fm.Pc + cnt[1] != fm.Pg || fm.Pg + cnt[2] != fm.Pt || fm.Pt + cnt[3] != (uint64_t)n + 1)
        return fail(BWT_EINVAL, "index: pointers disagree with the base counts of L");

    bwt_index *ix = new bwt_index();
    ix->device = device;
    ix->Pc = fm.Pc; ix->Pg = fm.Pg; ix->Pt = fm.Pt; ix->Last = fm.Last; ix->shift = fm.shift;
    ix->ref1_len = ref_len + 2 * BWT_REF_EXT;
    std::vector<uint8_t> r1(ix->ref1_len, (uint8_t)'N');      // worker_bwt.erl:54-55
    if (ref_len) memcpy(r1.data() + BWT_REF_EXT, ref, ref_len);
    int rc = 0;
    if ((rc = ix->blocks.ensure(blk.size() * 4)) || (rc = ix->sa.ensure((size_t)n * 4)) ||
        (rc = ix->ref1.ensure(r1.size()))) { bwt_index_close(ix); return rc; }
    hipError_t e;
    if ((e = hipMemcpy(ix->blocks.p, blk.data(), blk.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(ix->sa.p, fm.SA.data(), (size_t)n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(ix->ref1.p, r1.data(), r1.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking)) != hipSuccess) {
        bwt_index_close(ix);
        return fail(BWT_EHIP, std::string("index upload: ") + hipGetErrorString(e));
    }
    ix->d.blocks = ix->blocks.as<uint4>(); ix->d.sa = ix->sa.as<uint32_t>(); ix->d.ref1 = ix->ref1.as<uint8_t>();
    ix->d.ref1_len = ix->ref1_len;
    ix->d.Pc = fm.Pc; ix->d.Pg = fm.Pg; ix->d.Pt = fm.Pt; ix->d.Last = fm.Last; ix->d.dollar = dollar; ix->d.nblk = nblk;
    ix->d.c0[0] = 1; ix->d.c0[1] = fm.Pc - 1; ix->d.c0[2] = fm.Pg - 1; ix->d.c0[3] = fm.Pt - 1;
    if ((rc = bwt_index_make_refplanes(ix, ix->stream)) || (rc = bwt_index_set_ktable(ix, -1, -1))) { bwt_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

extern "C" int bwt_index_open(const char *fm_path, const char *ref_path, int device, bwt_index **out) {
    if (!fm_path || !ref_path || !out) return fail(BWT_EINVAL, "bwt_index_open: null argument");
    FmHost fm;
    std::string err;
    int rc = fm_read_etf(fm_path, fm, err);
    if (rc) return fail(rc, err);
    FILE *f = fopen(ref_path, "rb");
    if (!f) return fail(BWT_EIO, std::string("cannot open ") + ref_path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> ref((size_t)sz);
    if (sz > 0 && fread(ref.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return fail(BWT_EIO, "short read on .ref"); }
    fclose(f);
    return index_build(fm, ref.data(), ref.size(), device, out);
}

extern "C" int bwt_index_from_arrays(const uint8_t *L, const uint32_t *SA, uint32_t n_rows, const uint32_t pointers[4],
                                     int32_t shift, const uint8_t *ref, uint64_t ref_len, int device, bwt_index **out) {
    if (!L || !SA || !pointers || !out || (!ref && ref_len)) return fail(BWT_EINVAL, "bwt_index_from_arrays: null argument");
    FmHost fm;
    fm.L.assign(L, L + n_rows);
    fm.SA.assign(SA, SA + n_rows);
    fm.Pc = pointers[0]; fm.Pg = pointers[1]; fm.Pt = pointers[2]; fm.Last = pointers[3]; fm.shift = shift;
    return index_build(fm, ref, ref_len, device, out);
}

extern "C" int bwt_fm_check(const char *fm_path, uint32_t out[5]) {
    if (!fm_path || !out) return fail(BWT_EINVAL, "bwt_fm_check: null argument");
    FmHost fm;
    std::string err;
    int rc = fm_read_etf(fm_path, fm, err);
    if (rc) return fail(rc, err);
    out[0] = fm.Pc; out[1] = fm.Pg; out[2] = fm.Pt; out[3] = fm.Last; out[4] = (uint32_t)fm.shift;
    return 0;
}

extern "C" int bwt_fm_rewrite(const char *in_path, const char *out_path) {
    if (!in_path || !out_path) return fail(BWT_EINVAL, "bwt_fm_rewrite: null argument");
    FmHost fm;
    std::string err;
    int rc = fm_read_etf(in_path, fm, err);
    if (rc) return fail(rc, err);
    if ((rc = fm_write_etf(out_path, fm, err))) return fail(rc, err);
    return 0;
}

extern "C" void bwt_index_close(bwt_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    ix->blocks.release(); ix->sa.release(); ix->ref1.release(); ix->refplanes.release(); ix->ktab.release(); ix->ktw.release();
    delete ix;
}

extern "C" int bwt_index_meta(const bwt_index *ix, uint32_t out[5]) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_index_meta: null argument");
    out[0] = ix->Pc; out[1] = ix->Pg; out[2] = ix->Pt; out[3] = ix->Last; out[4] = (uint32_t)ix->shift;
    return 0;
}

extern "C" int bwt_index_device_bytes(const bwt_index *ix, uint64_t out[4]) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_index_device_bytes: null argument");
    out[0] = (uint64_t)ix->d.nblk * 64; out[1] = (uint64_t)ix->Last * 4; out[2] = ix->ref1_len + ix->d.rpw * 4;
    out[3] = ix->d.kt_k ? (ix->d.ktw ? kt_level_off(ix->d.kt_k) * 8 + ix->ktw_bytes : kt_entries(ix->d.kt_k) * 8) : 0;
    return 0;
}

extern "C" int bwt_ref1_read(const bwt_index *ix, uint64_t off, uint64_t len, uint8_t *outp) {
    if (!ix || (!outp && len)) return fail(BWT_EINVAL, "bwt_ref1_read: null argument");
    if (off > ix->ref1_len || len > ix->ref1_len - off) return fail(BWT_EINVAL, "bwt_ref1_read: range outside Ref1");
    HIPCHK(hipSetDevice(ix->device));
    if (len) HIPCHK(hipMemcpy(outp, ix->ref1.as<uint8_t>() + off, len, hipMemcpyDeviceToHost));
    return 0;
}

__global__ __launch_bounds__(256) static void k_ref1_gather(const uint8_t *__restrict__ ref1, const int64_t *__restrict__ woff,
                                                             const uint64_t *__restrict__ ooff, uint64_t n, uint8_t *__restrict__ out) {
    // wave per window: coalesced byte copies
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (w >= n) return;
    const uint64_t o = ooff[w], len = ooff[w + 1] - o;
    const uint8_t *src = ref1 + woff[w];
    for (uint64_t i = lane; i < len; i += 64) out[o + i] = src[i];
}

extern "C" int bwt_ref1_gather(const bwt_index *ix, const int64_t *win_off, const int32_t *win_len, uint64_t n, uint8_t *outp,
                               uint64_t out_cap, uint64_t *out_off, uint64_t *needed) {
    if (!ix || (!win_off && n) || (!win_len && n) || !out_off) return fail(BWT_EINVAL, "bwt_ref1_gather: null argument");
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (win_off[i] < 0 || win_len[i] < 0 || (uint64_t)win_off[i] + (uint64_t)win_len[i] > ix->ref1_len)
            return fail(BWT_EINVAL, "bwt_ref1_gather: window outside Ref1");
        out_off[i] = tot; tot += (uint64_t)win_len[i];
    }
    out_off[n] = tot;
    if (needed) *needed = tot;
    if (tot > out_cap || (!outp && tot)) return fail(BWT_ENOSPC, "RefSeq capacity too small");
    if (!n || !tot) return 0;
    HIPCHK(hipSetDevice(ix->device));
    DevBuf dwo, doo, dout;
    struct G { DevBuf &a, &b, &c; ~G() { a.release(); b.release(); c.release(); } } g{dwo, doo, dout};
    int rc;
    if ((rc = dwo.ensure(n * 8)) || (rc = doo.ensure((n + 1) * 8)) || (rc = dout.ensure(tot))) return rc;
    HIPCHK(hipMemcpyAsync(dwo.p, win_off, n * 8, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(doo.p, out_off, (n + 1) * 8, hipMemcpyHostToDevice, ix->stream));
    hipLaunchKernelGGL(k_ref1_gather, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, ix->stream, ix->ref1.as<uint8_t>(),
                       dwo.as<int64_t>(), doo.as<uint64_t>(), n, dout.as<uint8_t>());
    HIPCHK(hipMemcpyAsync(outp, dout.p, tot, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
struct KernelStat { double ms = 0; uint64_t launches = 0; };

struct bwt_batch {
    bwt_index *ix = nullptr;         // may be null for the index-free SW entry point
    bwt_options opt;                 // execution options (bwt_batch_set_options; a new batch takes its index's)
    bwt_batch() { bwt_options_init(&opt); }
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int occ_seed[4] = {0, 0, 0, 0};    // resident blocks per CU of the three k_seed_tasks instances and k_seed_fast (queried once per batch)
    DevBuf gq;                         // groups k_seed_fast handed back
    uint32_t occ_seed_maxT = 0;        // ... k_seed_fast's occupancy was queried for this many sub-reads per read (its LDS depends on it)
    // inputs
    uint32_t n = 0, maxT = 0, max_lq = 0;
    uint64_t total_bases = 0;
    DevBuf bases, off, rplanes;
    DevBuf pk_codes, pk_exc;           // packed upload staging (bwt_batch_set_reads_packed)
    const uint32_t *pk_zc = nullptr;   // up_zero_copy: the codes were not copied; they are still in the caller's pinned buffer (valid until its next batch)
    uint64_t pk_nexc = 0;
    bool planes_valid = false;         // rplanes already hold this batch's planes (packed upload): the search stage skips k_read_planes
    bool ascii_valid = true;           // `bases` holds every read's bytes (false after a packed upload: only the exceptions' bytes)
    // tasks
    DevBuf t_def, t_sp, t_ep, t_info, t_cnt, r_pend, r_cl, smallq;
    bool counts_done = false;          // the search stage already totalled the reads (k_seed_tasks)
    uint32_t n_small = 0;
    uint32_t wave_upto = 0;            // DP queues up to this size went to k_sw_wave in the last extend stage
    uint32_t ulen = 0;                 // common read length of the batch, 0 if the lengths differ
    bool rows_kept = true;             // every read's task-table rows were written (not so after a lean fused run)
    bool fused = false;                // ... and clustered what it could (r_K zeroed, RQ_DONE reads staged in r_cl)
    bool cand_pending = false;         // the candidate arrays are still to be made (by the prefilter, from the per-read clusters)
    bool max_w_known = false;          // ... and the largest window is known without asking the prefilter
    bool desc_partial = false;         // the fused prefilter wrote c_qoff / c_qlen / c_woff only for the candidates it queued
    // reads
    DevBuf r_P, r_qty, r_K, r_nhit, r_cigb, r_status, bigq, mid2q, mid4q, seed_off, clus_off, hit_off, cigb_off;
    DevBuf pos, scan_tmp;
    // candidates
    uint64_t n_cand = 0;
    DevBuf c_read, c_S, c_D, c_qoff, c_qlen, c_woff, c_wlen, c_status, c_score, c_ciglen, cig, fpq;
    uint32_t cig_stride = 32;
    // sw slabs
    DevBuf slab_bnd, slab_bnd_t, slab_bndv, slab_dir, slab_bnd_fp, slab_dir_fp, slab_dir_pk, tq, dq, tq2, tqw;
    DevBuf bq, bq2, bq3, slab_band;    // k_sw_band: the queues of what its tiers pass on (bq3: first tier -> third), the direction slab
    DevBuf dq_lo, bq0, bqs;            // the 12-diagonal tier: its queue (filled by the prefilter), what it passes on, what it sends past the next tier
    DevBuf hugeq;                      // reads k_cluster_big lists for k_cluster_huge
    DevBuf cq, cq2;                    // what k_sw_cert leaves of dq_lo (the 12-diagonal tier's queue when the certificate kernel runs); the queue of its wider instance
    bool band_ran = false;
    int occ_prefilter = 0;             // resident blocks per CU of k_sw_prefilter<true> (queried once per batch)
    DevBuf pfq;                        // candidates the fused prefilter pass listed for the quad-per-candidate pass
    // hits
    uint64_t n_hits = 0, cig_bytes = 0;
    uint64_t n_extra = 0, xcig_bytes = 0;
    DevBuf h_rec, h_woff, h_wlen, h_cigoff, h_cig, pairs;
    DevBuf r_nx, r_xcb, x_off, xcb_off, x_rec, x_cig, h_xidx, h_len;    // compact table: extras of the hits that are not plain
    bool expanded = false;             // h_woff / h_wlen / h_cigoff / h_cig hold the expansion of the current compact table
    // the NARROW form of the table (bwt_options.hits_narrow): reads whose answer is one plain hit are out of h_rec; their window offset
    // and score sit in n_woff / n_score (in read order), r_status says 2 for them.  Entry points that hand out the 16-byte form
    // widen the table first (ensure_v1: k_hits_widen into the w_* buffers, which then take the places of h_rec / x_rec / hit_off /
    // r_status).
    bool narrow = false;
    uint64_t n_narrow = 0;
    DevBuf r_nn, nn_off, n_woff, n_score, w_rec, w_xrec, w_hoff, w_status;
    // counters
    DevBuf ctr;
    Counters hctr{};
    int done = 0;                    // stages completed
    // for sw-only batches
    const uint8_t *sw_rbuf = nullptr;
    uint64_t sw_rbuf_len = 0;
    // profiling
    bool prof = false;
    struct Pending { const char *name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    std::map<std::string, KernelStat> stats;
    hipEvent_t ev_uploaded = nullptr;    // bwt_batch_set_reads_packed_hook: the host buffers have crossed the link
    // small device -> host values of a run (counters, scan totals) do not go through the copy engines: a batch in a pipe shares
    // those with the other slots' 0.2-0.3 GB transfers, and an 8-byte copy queued behind one of them stalls the kernels' turn for
    // milliseconds.  A one-wave kernel writes them into pinned host memory instead (mail), delivered at the next synchronisation.
    uint8_t *mbox = nullptr;             // pinned, MBOX_BYTES
    struct Mail { void *dst; uint32_t at, bytes; };
    std::vector<Mail> mail_pending;
    uint32_t mbox_used = 0;
};

static int prof_collect(bwt_batch *b) {
    for (auto &p : b->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &s = b->stats[p.name];
            s.ms += ms; s.launches++;
        }
        b->ev_pool.push_back(p.a); b->ev_pool.push_back(p.b);
    }
    b->pending.clear();
    return 0;
}

struct ProfScope {
    bwt_batch *b; const char *name; hipEvent_t a = nullptr, e = nullptr;
    ProfScope(bwt_batch *b_, const char *n) : b(b_), name(n) {
        if (!b->prof) return;
        auto get = [&]() { hipEvent_t ev = nullptr; if (!b->ev_pool.empty()) { ev = b->ev_pool.back(); b->ev_pool.pop_back(); } else (void)hipEventCreate(&ev); return ev; };
        a = get(); e = get();
        (void)hipEventRecord(a, b->stream);
    }
    ~ProfScope() {
        if (!b->prof) return;
        (void)hipEventRecord(e, b->stream);
        b->pending.push_back({name, a, e});
    }
};

#define KLAUNCH(b, name, kern, grid, block, ...)                                   \
    do {                                                                           \
        ProfScope _ps(b, name);                                                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (b)->stream, __VA_ARGS__); \
    } while (0)

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// exclusive scan in[0..n) (u32) -> out[0..n] (u64, out[n] = total)
static int dev_scan(bwt_batch *b, const uint32_t *in, uint64_t *out, uint64_t n) {
    if (n <= SCAN_TILE) {                                   // one tile: one launch
        KLAUNCH(b, "scan", k_scan_one, 1, SCAN_BLOCK, in, out, n);
        return 0;
    }
    const uint32_t nb = std::max<uint32_t>(1, cdiv(n, SCAN_TILE));
    int rc = b->scan_tmp.ensure(((size_t)nb + 1) * 8);
    if (rc) return rc;
    uint64_t *bs = b->scan_tmp.as<uint64_t>();
    KLAUNCH(b, "scan", k_scan1, nb, SCAN_BLOCK, in, bs, n);
    KLAUNCH(b, "scan", k_scan2, 1, SCAN_BLOCK, bs, nb);
    KLAUNCH(b, "scan", k_scan3, nb, SCAN_BLOCK, in, out, bs, n, nb);
    return 0;
}

// three scans of the same length (the hit, extra and CIGAR-byte offsets of a batch): one set of launches
static int dev_scan3(bwt_batch *b, const uint32_t *in0, uint64_t *out0, const uint32_t *in1, uint64_t *out1, const uint32_t *in2,
                     uint64_t *out2, uint64_t n, const uint32_t *in3, uint64_t *out3) {
    // (in3 / out3: a fourth array in the same launches, or null)
    if (n <= SCAN_TILE) {
        int rc = dev_scan(b, in0, out0, n);
        if (!rc) rc = dev_scan(b, in1, out1, n);
        if (!rc) rc = dev_scan(b, in2, out2, n);
        return rc || !in3 ? rc : dev_scan(b, in3, out3, n);
    }
    const uint32_t nb = std::max<uint32_t>(1, cdiv(n, SCAN_TILE)), na = in3 ? 4u : 3u;
    int rc = b->scan_tmp.ensure(((size_t)nb + 1) * 8 * 4);
    if (rc) return rc;
    ScanPair p;
    p.in[0] = in0; p.in[1] = in1; p.in[2] = in2; p.in[3] = in3; p.out[0] = out0; p.out[1] = out1; p.out[2] = out2; p.out[3] = out3;
    p.bsum[0] = b->scan_tmp.as<uint64_t>(); p.bsum[1] = p.bsum[0] + nb + 1; p.bsum[2] = p.bsum[1] + nb + 1; p.bsum[3] = p.bsum[2] + nb + 1;
    ProfScope ps(b, "scan");
    hipLaunchKernelGGL(k_scan1_pair, dim3(nb, na), dim3(SCAN_BLOCK), 0, b->stream, p, n);
    hipLaunchKernelGGL(k_scan2_pair, dim3(1, na), dim3(SCAN_BLOCK), 0, b->stream, p, nb);
    hipLaunchKernelGGL(k_scan3_pair, dim3(nb, na), dim3(SCAN_BLOCK), 0, b->stream, p, n, nb);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int bwt_batch_create(bwt_index *ix, bwt_batch **out) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_batch_create: null argument");
    HIPCHK(hipSetDevice(ix->device));
    bwt_batch *b = new bwt_batch();
    b->ix = ix; b->device = ix->device;
    { std::shared_lock<std::shared_mutex> lk(ix->mu); b->opt = ix->opt; }
    // its own stream: batches of different host threads on one index do not serialise (or free each other's buffers)
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete b; return fail(BWT_EHIP, "hipStreamCreate failed"); }
    b->own_stream = true;
    int rc = b->ctr.ensure(sizeof(Counters));
    if (rc) { bwt_batch_destroy(b); return rc; }
    *out = b;
    return 0;
}

static int batch_create_noindex(int device, bwt_batch **out) {
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    bwt_batch *b = new bwt_batch();
    b->device = device;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete b; return fail(BWT_EHIP, "hipStreamCreate failed"); }
    b->own_stream = true;
    int rc = b->ctr.ensure(sizeof(Counters));
    if (rc) { bwt_batch_destroy(b); return rc; }
    *out = b;
    return 0;
}

extern "C" int bwt_batch_set_options(bwt_batch *b, const bwt_options *o) {
    if (!b) return fail(BWT_EINVAL, "bwt_batch_set_options: null batch");
    return options_copy(b->opt, o, "bwt_batch_set_options");
}

extern "C" void bwt_batch_destroy(bwt_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    prof_collect(b);
    for (auto ev : b->ev_pool) (void)hipEventDestroy(ev);
    if (b->ev_uploaded) (void)hipEventDestroy(b->ev_uploaded);
    if (b->mbox) (void)hipHostFree(b->mbox);
    DevBuf *all[] = {&b->bases, &b->off, &b->rplanes, &b->pk_codes, &b->pk_exc, &b->pfq, &b->r_nx, &b->r_xcb, &b->x_off, &b->xcb_off, &b->x_rec, &b->x_cig, &b->h_xidx, &b->h_len, &b->t_def, &b->t_sp, &b->t_ep, &b->t_info, &b->t_cnt, &b->r_P, &b->r_qty,
                     &b->r_K, &b->r_nhit, &b->r_cigb, &b->r_status, &b->bigq, &b->mid2q, &b->mid4q, &b->seed_off, &b->clus_off, &b->hit_off,
                     &b->cigb_off, &b->pos, &b->scan_tmp, &b->c_read, &b->c_S, &b->c_D, &b->c_qoff, &b->c_qlen,
                     &b->c_woff, &b->c_wlen, &b->c_status, &b->c_score, &b->c_ciglen, &b->cig, &b->fpq, &b->slab_bnd,
                     &b->slab_bndv, &b->slab_dir, &b->slab_bnd_fp, &b->slab_dir_fp, &b->slab_bnd_t, &b->tq, &b->dq, &b->tq2, &b->tqw, &b->bq, &b->bq2, &b->bq3, &b->slab_band, &b->slab_dir_pk, &b->h_rec, &b->h_woff, &b->h_wlen,
                     &b->h_cigoff, &b->h_cig, &b->ctr, &b->r_pend, &b->r_cl, &b->smallq, &b->gq, &b->pairs,
                     &b->r_nn, &b->nn_off, &b->n_woff, &b->n_score, &b->w_rec, &b->w_xrec, &b->w_hoff, &b->w_status, &b->dq_lo, &b->bq0, &b->bqs, &b->cq, &b->cq2, &b->hugeq};
    for (auto d : all) d->release();
    if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

extern "C" int bwt_batch_profile(bwt_batch *b, int enable) {
    if (!b) return fail(BWT_EINVAL, "bwt_batch_profile: null batch");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    prof_collect(b);
    b->stats.clear();
    b->prof = enable != 0;
    return 0;
}

extern "C" int bwt_batch_kernel_times(bwt_batch *b, char *names, uint32_t names_cap, double *ms, uint64_t *launches,
                                      uint32_t cap) {
    if (!b) return fail(BWT_EINVAL, "bwt_batch_kernel_times: null batch");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    prof_collect(b);
    uint32_t k = 0, np = 0;
    for (auto &kv : b->stats) {
        if (k >= cap || np + kv.first.size() + 1 > names_cap) break;
        memcpy(names + np, kv.first.c_str(), kv.first.size() + 1);
        np += (uint32_t)kv.first.size() + 1;
        ms[k] = kv.second.ms; launches[k] = kv.second.launches;
        k++;
    }
    return (int)k;
}

// lengths of the reads of a batch, checked where the offsets already are: {min, max, flags (1: offsets decrease, 2: a read > 65535)}
__global__ __launch_bounds__(256) static void k_off_check(const uint64_t *__restrict__ off, uint32_t n, unsigned int *__restrict__ out) {
    uint32_t mn = 0xffffffffu, mx = 0, bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t a = off[i], b = off[i + 1];
        if (b < a) bad |= 1u;
        else {
            const uint64_t l = b - a;
            if (l > 65535) bad |= 2u;
            const uint32_t l32 = l > 0xffffffffull ? 0xffffffffu : (uint32_t)l;
            mn = l32 < mn ? l32 : mn; mx = l32 > mx ? l32 : mx;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)mn, o, 64), b = (uint32_t)__shfl_xor((int)mx, o, 64), c = (uint32_t)__shfl_xor((int)bad, o, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx; bad |= c;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], mn); atomicMax(&out[1], mx); if (bad) atomicOr(&out[2], bad); }
}

extern "C" int bwt_batch_set_reads(bwt_batch *b, const uint8_t *bases, const uint64_t *off, uint32_t n) {
    if (!b || (!bases && n) || !off) return fail(BWT_EINVAL, "bwt_batch_set_reads: null argument");
    HIPCHK(hipSetDevice(b->device));
    if (n && off[n] < off[0]) return fail(BWT_EINVAL, "bwt_batch_set_reads: offsets must be non-decreasing");
    const uint64_t total = n ? off[n] - off[0] : 0;
    int rc;
    if ((rc = b->bases.ensure(total + 16)) || (rc = b->off.ensure(((size_t)n + 1) * 8)) || (rc = b->scan_tmp.ensure(64))) return rc;
    // The offsets go up as they are when they start at 0 (with pinned caller buffers both copies are plain DMA); otherwise
    // they are rebased on the host first.  The per-read length checks run on the device, over the uploaded offsets.
    std::vector<uint64_t> o;
    const uint64_t *src = off;
    if (off[0] != 0) {
        o.resize((size_t)n + 1);
        for (uint32_t i = 0; i <= n; i++) o[i] = off[i] - off[0];
        src = o.data();
    }
    if (total) HIPCHK(hipMemcpyAsync(b->bases.p, bases + off[0], total, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->off.p, src, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, b->stream));
    unsigned int chk[3] = {0xffffffffu, 0u, 0u};
    unsigned int *dchk = b->scan_tmp.as<unsigned int>();
    HIPCHK(hipMemcpyAsync(dchk, chk, sizeof chk, hipMemcpyHostToDevice, b->stream));
    if (n) hipLaunchKernelGGL(k_off_check, dim3(std::min<uint32_t>(cdiv(n, 256), 1024)), dim3(256), 0, b->stream, b->off.as<uint64_t>(), n, dchk);
    HIPCHK(hipMemcpyAsync(chk, dchk, sizeof chk, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    b->n = 0; b->done = 0; b->n_cand = 0; b->n_hits = 0; b->cig_bytes = 0;
    if (chk[2] & 1u) return fail(BWT_EINVAL, "bwt_batch_set_reads: offsets must be non-decreasing");
    if (chk[2] & 2u) return fail(BWT_EINVAL, "bwt_batch_set_reads: read longer than 65535 bases");
    const uint32_t maxlq = n ? chk[1] : 0, minlq = n ? chk[0] : 0;
    b->n = n; b->max_lq = maxlq; b->maxT = std::max<uint32_t>(1, maxlq / BWT_MIN_LEN);
    b->ulen = (n && minlq == maxlq) ? maxlq : 0;       // every read the same length: offsets are r * ulen (no offset load in the seed kernel)
    b->total_bases = total;
    b->planes_valid = false; b->ascii_valid = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Packed upload (include/bwt_gpu.h): 2 bits per base + an exception list up, the planes made on the device.

extern "C" int bwt_batch_set_reads_packed(bwt_batch *b, const uint32_t *codes, uint64_t total, uint32_t n, uint32_t read_len,
                                          const uint64_t *off, const uint64_t *exc, uint64_t n_exc) {
    return bwt_batch_set_reads_packed_hook(b, codes, total, n, read_len, off, exc, n_exc, nullptr, nullptr);
}

// (common.h) `uploaded(arg)` is called once, as soon as the host buffers have crossed the link -- before the unpack kernels are
// waited for -- or not at all when the call fails earlier: bwt_pipe passes its upload turn on there
int bwt_batch_set_reads_packed_hook(bwt_batch *b, const uint32_t *codes, uint64_t total, uint32_t n, uint32_t read_len,
                                    const uint64_t *off, const uint64_t *exc, uint64_t n_exc, void (*uploaded)(void *), void *arg) {
    if (!b || (!codes && total) || (!exc && n_exc)) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: null argument");
    if (!read_len && !off) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: offsets are needed when the reads differ in length");
    if (read_len > 65535) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: read longer than 65535 bases");
    if (read_len && (uint64_t)n * read_len != total) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: total_bases must be n * read_len");
    if (!read_len && (off[0] != 0 || off[n] != total)) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: off[0] must be 0 and off[n] the number of bases");
    if (total >> 56) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: too many bases");
    HIPCHK(hipSetDevice(b->device));
    if (uploaded && !b->ev_uploaded) HIPCHK(hipEventCreateWithFlags(&b->ev_uploaded, hipEventDisableTiming));
    const uint64_t nwords = (total + 31) / 32;
    int rc;
    if ((rc = b->bases.ensure(nwords * 32 + 16)) || (rc = b->off.ensure(((size_t)n + 1) * 8)) || (rc = b->scan_tmp.ensure(64)) ||
        (rc = b->pk_codes.ensure(nwords * 8 + 16)) || (rc = b->pk_exc.ensure(n_exc * 8 + 16)) ||
        (rc = b->rplanes.ensure(rp_words(total) * 4)))
        return rc;
    b->n = 0; b->done = 0; b->n_cand = 0; b->n_hits = 0; b->cig_bytes = 0; b->n_extra = 0; b->xcig_bytes = 0;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
    // up_zero_copy: the codes stay where they are (pinned host memory) and k_unpack_reads reads them over the link -- one stream of
    // reads instead of a DMA into HBM followed by a kernel over the copy; the upload is then over when that kernel is
    // (only when `codes` really is pinned: a kernel reading pageable memory faults, so anything else is copied as before)
    bool zc = false;
    if (b->opt.up_zero_copy >= 1 && nwords) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, codes) == hipSuccess) zc = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    b->pk_zc = zc ? codes : nullptr;
    if (nwords && !zc) HIPCHK(hipMemcpyAsync(b->pk_codes.p, codes, nwords * 8, hipMemcpyHostToDevice, b->stream));
    if (n_exc) HIPCHK(hipMemcpyAsync(b->pk_exc.p, exc, n_exc * 8, hipMemcpyHostToDevice, b->stream));
    unsigned int chk[4] = {0xffffffffu, 0u, 0u, 0u};           // {min length, max length, offset flags, exception flags}
    unsigned int *dchk = b->scan_tmp.as<unsigned int>();
    HIPCHK(hipMemcpyAsync(dchk, chk, sizeof chk, hipMemcpyHostToDevice, b->stream));
    if (read_len) {
        if (uploaded && !zc) HIPCHK(hipEventRecord(b->ev_uploaded, b->stream));
        if (n) hipLaunchKernelGGL(k_off_uniform, dim3(std::min<uint32_t>(cdiv((uint64_t)n + 1, 256), (uint32_t)cus * 8)), dim3(256), 0, b->stream, b->off.as<uint64_t>(), n, read_len);
        else HIPCHK(hipMemsetAsync(b->off.p, 0, 8, b->stream));
    } else {
        HIPCHK(hipMemcpyAsync(b->off.p, off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, b->stream));
        if (uploaded && !zc) HIPCHK(hipEventRecord(b->ev_uploaded, b->stream));
        if (n) hipLaunchKernelGGL(k_off_check, dim3(std::min<uint32_t>(cdiv(n, 256), 1024)), dim3(256), 0, b->stream, b->off.as<uint64_t>(), n, dchk);
    }
    {
        ProfScope ps(b, "unpack_reads");
        const uint64_t words = rp_words(total), used = nwords * 4;
        HIPCHK(hipMemsetAsync(b->rplanes.as<uint32_t>() + used, 0, (words - used) * 4, b->stream));
        // the ASCII copy is only made where something reads bytes: here for the whole batch when no prefilter will run in front of
        // the DP kernels (it lists the candidates that reach them: stage_extend makes those reads' bytes then)
        const bool all_ascii = b->opt.sw_prefilter == 0;
        if (nwords)
            hipLaunchKernelGGL(k_unpack_reads, dim3((uint32_t)std::min<uint64_t>(cdiv(nwords, 256), zc ? (uint64_t)(b->opt.up_zero_copy == 1 ? 64 : b->opt.up_zero_copy) : (uint64_t)cus * 16)), dim3(256), 0, b->stream,
                               zc ? (const uint2 *)codes : (const uint2 *)b->pk_codes.as<uint2>(), nwords, total, b->rplanes.as<uint32_t>(),
                               all_ascii ? b->bases.as<uint8_t>() : (uint8_t *)nullptr);
        if (uploaded && zc) HIPCHK(hipEventRecord(b->ev_uploaded, b->stream));
        if (n_exc)
            hipLaunchKernelGGL(k_unpack_exc, dim3((uint32_t)std::min<uint64_t>(cdiv(n_exc, 256), (uint64_t)cus * 8)), dim3(256), 0, b->stream,
                               (const uint64_t *)b->pk_exc.as<uint64_t>(), n_exc, total, b->rplanes.as<uint32_t>(), b->bases.as<uint8_t>(), dchk + 3);
        b->ascii_valid = all_ascii;
    }
    HIPCHK(hipMemcpyAsync(chk, dchk, sizeof chk, hipMemcpyDeviceToHost, b->stream));
    if (uploaded) {
        const hipError_t e = hipEventSynchronize(b->ev_uploaded);
        uploaded(arg);
        HIPCHK(e);
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    if (!read_len) {
        if (chk[2] & 1u) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: offsets must be non-decreasing");
        if (chk[2] & 2u) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: read longer than 65535 bases");
    }
    if (chk[3] & 1u) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: exception position behind the last base");
    if (chk[3] & 2u) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: an exception byte is one of ACGT");
    if (chk[3] & 4u) return fail(BWT_EINVAL, "bwt_batch_set_reads_packed: exceptions must be strictly ascending by position");
    const uint32_t maxlq = !n ? 0 : read_len ? read_len : chk[1], minlq = !n ? 0 : read_len ? read_len : chk[0];
    b->n = n; b->max_lq = maxlq; b->maxT = std::max<uint32_t>(1, maxlq / BWT_MIN_LEN);
    b->ulen = (n && minlq == maxlq) ? maxlq : 0;
    b->total_bases = total;
    b->planes_valid = true; b->pk_nexc = n_exc;
    return 0;
}

// every read's bytes out of the codes (a packed batch that meets a stage which reads whole reads as bytes)
static int ascii_all(bwt_batch *b) {
    if (b->ascii_valid) return 0;
    const uint64_t nwords = (b->total_bases + 31) / 32;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
    ProfScope ps(b, "unpack_reads");
    if (nwords)
        hipLaunchKernelGGL(k_unpack_reads, dim3((uint32_t)std::min<uint64_t>(cdiv(nwords, 256), (uint64_t)cus * 16)), dim3(256), 0, b->stream,
                           b->pk_zc ? (const uint2 *)b->pk_zc : (const uint2 *)b->pk_codes.as<uint2>(), nwords, b->total_bases, (uint32_t *)nullptr, b->bases.as<uint8_t>());
    if (b->pk_nexc)            // the exceptions' bytes over the code bytes (their plane bits are set already; setting them again changes nothing)
        hipLaunchKernelGGL(k_unpack_exc, dim3((uint32_t)std::min<uint64_t>(cdiv(b->pk_nexc, 256), (uint64_t)cus * 8)), dim3(256), 0, b->stream,
                           (const uint64_t *)b->pk_exc.as<uint64_t>(), b->pk_nexc, b->total_bases, b->rplanes.as<uint32_t>(), b->bases.as<uint8_t>(),
                           b->scan_tmp.as<unsigned int>() + 3);
    HIPCHK(hipGetLastError());
    b->ascii_valid = true;
    return 0;
}

static int ensure_expanded(bwt_batch *b);
static int ensure_v1(bwt_batch *b);

// ------------------------------------------------------------------------------------------------
// Reverse strand and pairing: what the reference does not have (src/fastq.erl:43-50 is an unused helper).
// Nothing here changes the path above: a batch set with bwt_batch_set_reads_both_strands simply holds 2n reads -- read
// n + i is the reverse complement of read i, made on the device -- and every read goes through the same kernels.
__global__ __launch_bounds__(256) static void k_strand_copy(uint8_t *__restrict__ bases, uint64_t *__restrict__ off, uint32_t n, uint64_t total) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        const uint64_t a = off[r], len = off[r + 1] - a;
        for (uint64_t i = lane; i < len; i += 64) {
            const uint8_t c = bases[a + len - 1 - i];
            uint8_t d = c;                                        // IUPAC complements; anything else stays (the read fails as it would forward)
            switch (c) {
            case 'A': d = 'T'; break; case 'T': d = 'A'; break; case 'C': d = 'G'; break; case 'G': d = 'C'; break;
            case 'R': d = 'Y'; break; case 'Y': d = 'R'; break; case 'K': d = 'M'; break; case 'M': d = 'K'; break;
            case 'B': d = 'V'; break; case 'V': d = 'B'; break; case 'D': d = 'H'; break; case 'H': d = 'D'; break;
            default: break;                                       // N, S, W are their own complements
            }
            bases[total + a + i] = d;
        }
        if (lane == 0) off[n + r + 1] = total + off[r + 1];
    }
}

extern "C" int bwt_batch_set_reads_both_strands(bwt_batch *b, const uint8_t *bases, const uint64_t *off, uint32_t n) {
    if (!b || !off) return fail(BWT_EINVAL, "bwt_batch_set_reads_both_strands: null argument");
    if (n > 0x7fffffffu) return fail(BWT_EINVAL, "bwt_batch_set_reads_both_strands: too many reads");
    if (!bases && n) return fail(BWT_EINVAL, "bwt_batch_set_reads_both_strands: null argument");
    if (n && off[n] < off[0]) return fail(BWT_EINVAL, "bwt_batch_set_reads_both_strands: offsets must be non-decreasing");   // (before the sizes below are computed from them)
    HIPCHK(hipSetDevice(b->device));
    const uint64_t total = n ? off[n] - off[0] : 0;
    int rc;
    // room for both strands first (bwt_batch_set_reads keeps what is large enough), then the forward reads as usual
    if ((rc = b->bases.ensure(2 * total + 16)) || (rc = b->off.ensure(((size_t)2 * n + 1) * 8))) return rc;
    if ((rc = bwt_batch_set_reads(b, bases, off, n))) return rc;
    if (n) {
        hipLaunchKernelGGL(k_strand_copy, dim3((uint32_t)std::min<uint64_t>(cdiv((uint64_t)n * 64, 256), 4096)), dim3(256), 0, b->stream,
                           b->bases.as<uint8_t>(), b->off.as<uint64_t>(), n, total);
        HIPCHK(hipGetLastError());
    }
    b->n = 2 * n;
    b->total_bases = 2 * total;
    return 0;
}

// Proper pairs.  Mates are interleaved (reads 2p and 2p + 1 of the n = 2 n_pairs forward reads; their reverse complements are
// reads n + 2p and n + 2p + 1).  A combination is proper when one mate's forward read and the other mate's reverse
// complement both have hits, the forward one starts first, and the fragment they span -- from the forward hit's position to
// the end of the reverse hit (its position + the reference bases its CIGAR consumes) -- is min_insert..max_insert long.
// Among the proper combinations the one with the highest score sum wins; the first in enumeration order (mate 1 forward
// before mate 2 forward, then the forward mate's hits in table order, then the other's) among equals.
struct PairRec { uint32_t hit1, hit2; int32_t insert, score; uint32_t n_proper, flags; };
__device__ static int64_t cigar_ref_span(const uint8_t *cig, uint64_t a, uint64_t e) {
    int64_t span = 0, num = 0;
    for (uint64_t i = a; i < e; i++) {
        const uint8_t c = cig[i];
        if (c >= '0' && c <= '9') num = num * 10 + (c - '0');
        else { if (c == 'M' || c == 'D') span += num; num = 0; }
    }
    return span;
}
// reference bases each hit's CIGAR consumes (M + D), once per hit: the pairing loop below is quadratic in a read's hits
__global__ __launch_bounds__(256) static void k_hit_spans(uint64_t nh, const uint64_t *__restrict__ cigoff, const uint8_t *__restrict__ cig,
                                                           uint32_t *__restrict__ span) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += (uint64_t)gridDim.x * blockDim.x)
        span[h] = (uint32_t)cigar_ref_span(cig, cigoff[h], cigoff[h + 1]);
}
__global__ __launch_bounds__(256) static void k_pair_hits(uint32_t n_pairs, uint32_t n_fwd, const uint64_t *__restrict__ hit_off,
                                                           const HitRec *__restrict__ rec, const uint32_t *__restrict__ span,
                                                           int32_t min_ins, int32_t max_ins, PairRec *__restrict__ out) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
        PairRec best{0xffffffffu, 0xffffffffu, 0, 0, 0, 0};
        for (uint32_t o = 0; o < 2; o++) {                        // o = 0: mate 1 forward, mate 2 reverse; o = 1: the other way round
            const uint32_t rf = 2 * p + o, rr = n_fwd + 2 * p + (1 - o);
            for (uint64_t hf = hit_off[rf]; hf < hit_off[rf + 1]; hf++) {
                const HitRec f = rec[hf];
                for (uint64_t hr = hit_off[rr]; hr < hit_off[rr + 1]; hr++) {
                    const HitRec r = rec[hr];
                    if (r.pos < f.pos) continue;
                    const int64_t ins = r.pos + (int64_t)span[hr] - f.pos;
                    if (ins < min_ins || ins > max_ins) continue;
                    best.n_proper++;
                    const int32_t sc = f.score + r.score;
                    if (best.hit1 == 0xffffffffu || sc > best.score) {
                        best.hit1 = (uint32_t)(o == 0 ? hf : hr); best.hit2 = (uint32_t)(o == 0 ? hr : hf);
                        best.insert = (int32_t)ins; best.score = sc; best.flags = o;
                    }
                }
            }
        }
        out[p] = best;
    }
}

extern "C" int bwt_batch_pair_hits(bwt_batch *b, uint32_t n_pairs, int32_t min_insert, int32_t max_insert, bwt_pair_rec *pairs) {
    static_assert(sizeof(bwt_pair_rec) == sizeof(PairRec), "bwt_pair_rec is the device record");
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_pair_hits: extend stage not run");
    if ((uint64_t)n_pairs * 4 != b->n) return fail(BWT_EINVAL, "bwt_batch_pair_hits: the batch must hold both strands of 2 * n_pairs interleaved mates");
    if (min_insert < 0 || max_insert < min_insert) return fail(BWT_EINVAL, "bwt_batch_pair_hits: bad insert range");
    if (!pairs && n_pairs) return fail(BWT_EINVAL, "bwt_batch_pair_hits: null argument");
    HIPCHK(hipSetDevice(b->device));
    int rc;
    if ((rc = ensure_v1(b))) return rc;
    if (b->n_hits > 0xfffffffeull) return fail(BWT_EINVAL, "bwt_batch_pair_hits: more than 2^32 - 2 hits");
    if (!n_pairs) return 0;
    if ((rc = ensure_expanded(b)) || (rc = b->pairs.ensure((size_t)n_pairs * sizeof(PairRec)))) return rc;
    if ((rc = b->h_len.ensure((b->n_hits + 1) * 4))) return rc;              // (h_len is free again once the expansion is done: the spans go there)
    if (b->n_hits)
        hipLaunchKernelGGL(k_hit_spans, dim3((uint32_t)std::min<uint64_t>(cdiv(b->n_hits, 256), 4096)), dim3(256), 0, b->stream, b->n_hits,
                           (const uint64_t *)b->h_cigoff.as<uint64_t>(), (const uint8_t *)b->h_cig.as<uint8_t>(), b->h_len.as<uint32_t>());
    hipLaunchKernelGGL(k_pair_hits, dim3(std::min<uint32_t>(cdiv(n_pairs, 256), 4096)), dim3(256), 0, b->stream, n_pairs, b->n / 2,
                       b->hit_off.as<uint64_t>(), b->h_rec.as<HitRec>(), (const uint32_t *)b->h_len.as<uint32_t>(), min_insert,
                       max_insert, b->pairs.as<PairRec>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pairs, b->pairs.p, (size_t)n_pairs * sizeof(PairRec), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

// k_search is grid-stride: exactly as many blocks as are resident at once (a larger grid leaves a tail)
static uint32_t search_grid(bwt_batch *b, uint64_t n_tasks, bool u2 = true) {
    int cus = 256, per_cu = 8;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
    if (u2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_search<2>, 256, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_search<1>, 256, 0);
    per_cu = std::max(1, std::min(per_cu, 8));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cdiv(n_tasks * 4, 256), (uint64_t)cus * per_cu));
}

// reads -> bit-planes over the concatenated byte stream (k_read_planes); part of every search pass
static int read_planes(bwt_batch *b) {
    if (b->planes_valid) return 0;                          // a packed upload made them (k_unpack_reads)
    int rc = b->rplanes.ensure(rp_words(b->total_bases) * 4);
    if (rc) return rc;
    ProfScope ps(b, "read_planes");
    return launch_planes(b->bases.as<uint8_t>(), b->total_bases, b->rplanes.as<uint32_t>(), b->device, b->stream);
}

enum { MBOX_BYTES = 4096 };
static __global__ void k_mail(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t words) {
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}
// `bytes` (a multiple of 4) at device address `src` -> host `dst`, valid after the next mail_sync
static int mail(bwt_batch *b, void *dst, const void *src, uint32_t bytes) {
    if (!b->mbox) {
        if (hipHostMalloc((void **)&b->mbox, MBOX_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            b->mbox = nullptr;
            return fail(BWT_ENOMEM, "hipHostMalloc of the batch's mailbox failed");
        }
    }
    if ((bytes & 3u) || b->mbox_used + bytes > MBOX_BYTES) return fail(BWT_EHIP, "internal: mailbox overflow");
    hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, b->stream, (const uint32_t *)src, (uint32_t *)(b->mbox + b->mbox_used), bytes / 4);
    b->mail_pending.push_back({dst, b->mbox_used, bytes});
    b->mbox_used += (bytes + 15u) & ~15u;
    return 0;
}
static int mail_sync(bwt_batch *b) {
    const hipError_t e = hipStreamSynchronize(b->stream);
    for (auto &m : b->mail_pending) memcpy(m.dst, b->mbox + m.at, m.bytes);
    b->mail_pending.clear();
    b->mbox_used = 0;
    HIPCHK(e);
    return 0;
}
static int fetch_counters(bwt_batch *b) {
    int rc = mail(b, &b->hctr, b->ctr.p, (uint32_t)sizeof(Counters));
    return rc ? rc : mail_sync(b);
}
// (a function that mails into its own locals: nothing stays queued for them when it returns early)
struct MailGuard {
    bwt_batch *b;
    ~MailGuard() { b->mail_pending.clear(); b->mbox_used = 0; }
};

// The seed search.  With a suffix-interval table (and reads of at most 64 sub-reads): k_seed_tasks (sub-reads, table
// lookups, per-read totals) + k_search_pending (the LF steps the table left over) -- the per-read totals of the cluster
// stage are then already in place (b->counts_done).  Else: k_subreads + k_search over every task, from n = 1.
// raw: one task per "read", t_def already filled (bwt_find_seeds_batch).
// pre_zeroed: the caller's own kernel has already cleared r_K / r_pend / r_P (the small-batch path: a memset is a launch)
static int launch_search(bwt_batch *b, uint32_t n, uint32_t maxT, bool raw, bool lean = false, bool pre_zeroed = false) {
    int rc;
    const uint64_t T = (uint64_t)n * maxT;
    b->counts_done = false; b->fused = false; b->rows_kept = true;
    if (b->ix->d.kt_k > 0 && maxT >= 1 && maxT <= 64) {
        if ((rc = b->r_P.ensure((size_t)n * 4 + 4)) || (rc = b->r_qty.ensure((size_t)n * 4 + 4)) ||
            (rc = b->r_pend.ensure((size_t)n * 8 + 8)) || (rc = b->bigq.ensure((size_t)n * 4 + 4)) ||
            (rc = b->mid2q.ensure((size_t)n * 4 + 4)) || (rc = b->mid4q.ensure((size_t)n * 4 + 4)))
            return rc;
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
        const uint64_t want = cdiv(cdiv(n, 64 / maxT), 4);                      // wave per 64/maxT reads
#define SEED_ARGS b->ix->d, b->rplanes.as<uint32_t>(), b->off.as<uint64_t>(), n, maxT, b->t_def.as<uint32_t>(),                  \
                  b->t_sp.as<uint32_t>(), b->t_ep.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(),               \
                  b->r_pend.as<unsigned long long>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->bigq.as<uint32_t>(),       \
                  b->mid2q.as<uint32_t>(), b->mid4q.as<uint32_t>()
        // with SA values in the table's top level (and 8..64 sub-reads per read) the seed kernel clusters what it can itself
        const bool fuse = !raw && b->ix->d.ktw && maxT >= 8 && b->opt.seed_fuse != 0;
        if (fuse) {
            if ((rc = b->r_K.ensure((size_t)n * 4 + 4)) || (rc = b->r_cl.ensure((size_t)n * 8 * CL_STAGE + 16))) return rc;
            if (!pre_zeroed) HIPCHK(hipMemsetAsync(b->r_K.p, 0, (size_t)n * 4, b->stream));      // k_cluster_small only runs if reads are left for it
        }
        b->fused = fuse;
        b->rows_kept = !(fuse && lean);
        // persistent waves: as many blocks as are resident at once
        auto resident = [&](int which, const void *kern) {
            int &per_cu = b->occ_seed[which];
            if (!per_cu) {
                per_cu = 8;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 8; }
                per_cu = std::max(1, std::min(per_cu, 8));
            }
            return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)cus * per_cu));
        };
        // Batches of one read length with the full-depth wide table: k_seed_fast takes the ordinary groups of reads and hands
        // the others back as a list for k_seed_tasks (results are k_seed_tasks' either way).  Up to 21 sub-reads per read (three
        // reads per wave pass), up to 128 seeds per read.
        const bool fast = fuse && lean && b->ulen && b->ix->d.kt_k == BWT_MAX_LEN && maxT <= 21 && b->opt.seed_fast != 0;
        const uint32_t *nolist = nullptr;
        const unsigned int *nocount = nullptr;
        if (raw) KLAUNCH(b, "fm_lookup", (k_seed_tasks<true, false>), resident(0, (const void *)k_seed_tasks<true, false>), 256, SEED_ARGS, (uint32_t *)nullptr, (uint32_t *)nullptr, b->ctr.as<Counters>(), 0, 0u, nolist, nocount);
        else if (fast) {
            if ((rc = b->gq.ensure(((size_t)n + 2) * 4 + 256))) return rc;          // (the reads k_seed_fast hands back: at most all of them)
            {
                // reads of up to 14 sub-reads rarely pass 64 seeds: the one-register variant runs a wave more per SIMD
                const bool big = b->opt.seed_big >= 0 ? b->opt.seed_big == 1 : maxT > 14;
                const size_t lds = sf_lds_bytes(maxT, big);
                // the summary bit is looked at only when the index has one for the level of this batch's short last sub-read
                const int mn = (int)(b->ulen % BWT_MIN_LEN) + BWT_MIN_LEN;
                const bool sum = b->ix->d.ktsum && mn < BWT_MAX_LEN && ((b->ix->d.ktsum_levels >> mn) & 1u) && b->ix->d.ktw_off[mn - 8] == KTW_NONE;
                const void *kern = big ? (sum ? (const void *)k_seed_fast<true, true> : (const void *)k_seed_fast<true, false>)
                                       : (sum ? (const void *)k_seed_fast<false, true> : (const void *)k_seed_fast<false, false>);
                int &per_cu = b->occ_seed[3];
                if (!per_cu || b->occ_seed_maxT != (maxT | (big ? 0x100u : 0u) | (sum ? 0x200u : 0u))) {
                    per_cu = 8;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 4; }
                    per_cu = std::max(1, std::min(per_cu, 8));
                    b->occ_seed_maxT = maxT | (big ? 0x100u : 0u) | (sum ? 0x200u : 0u);
                }
                const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)cus * per_cu));
                if (!pre_zeroed) {
                    HIPCHK(hipMemsetAsync(b->r_pend.p, 0, (size_t)n * 8, b->stream));
                    HIPCHK(hipMemsetAsync(b->r_P.p, 0, (size_t)n * 4, b->stream));
                }
                ProfScope ps(b, "fm_lookup");
                {
#define SF_LAUNCH(BIGV, SUMV) hipLaunchKernelGGL((k_seed_fast<BIGV, SUMV>), dim3(grid), dim3(256), lds, b->stream, b->ix->d, b->rplanes.as<uint32_t>(), n, maxT, b->ulen, \
                                   b->r_pend.as<unsigned long long>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->r_K.as<uint32_t>(),                      \
                                   b->r_cl.as<uint32_t>(), b->gq.as<uint32_t>(), b->ctr.as<Counters>())
                    if (big) { if (sum) SF_LAUNCH(true, true); else SF_LAUNCH(true, false); }
                    else { if (sum) SF_LAUNCH(false, true); else SF_LAUNCH(false, false); }
#undef SF_LAUNCH
                }
            }
            KLAUNCH(b, "fm_lookup_rest", (k_seed_tasks<false, true>), resident(1, (const void *)k_seed_tasks<false, true>), 256, SEED_ARGS, b->r_K.as<uint32_t>(), b->r_cl.as<uint32_t>(), b->ctr.as<Counters>(), 1, b->ulen,
                    (const uint32_t *)b->gq.as<uint32_t>(), (const unsigned int *)&b->ctr.as<Counters>()->n_gq);
        } else if (fuse) KLAUNCH(b, "fm_lookup", (k_seed_tasks<false, true>), resident(1, (const void *)k_seed_tasks<false, true>), 256, SEED_ARGS, b->r_K.as<uint32_t>(), b->r_cl.as<uint32_t>(), b->ctr.as<Counters>(), lean ? 1 : 0, b->ulen, nolist, nocount);
        else KLAUNCH(b, "fm_lookup", (k_seed_tasks<false, false>), resident(2, (const void *)k_seed_tasks<false, false>), 256, SEED_ARGS, (uint32_t *)nullptr, (uint32_t *)nullptr, b->ctr.as<Counters>(), 0, b->ulen, nolist, nocount);
        {
            ProfScope ps(b, "fm_search");
            // reads per wave chunk: 64 when the batch fills the resident grid with such chunks, fewer (a power of two) when it does
            // not -- a wave works through its chunk's tasks 16 at a time, and a batch of a few hundred reads on four waves was
            // 0.12 ms of dependent LF steps with the rest of the chip idle
            const uint32_t sg = search_grid(b, 1ull << 28, false);                // (the resident grid: blocks, four waves each)
            uint32_t rpc = 64;
            while (rpc > 1 && (uint64_t)(rpc / 2) * sg * 4 >= n) rpc /= 2;
            hipLaunchKernelGGL(k_search_pending, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(sg, cdiv(cdiv(n, rpc), 4)))), dim3(256), (size_t)4 * 64 * maxT * 2, b->stream, SEED_ARGS,
                               b->ctr.as<Counters>(), rpc);
        }
#undef SEED_ARGS
        b->counts_done = true;
    } else {
        if (!raw)
            KLAUNCH(b, "subreads", k_subreads, cdiv(n, 256), 256, b->rplanes.as<uint32_t>(), b->off.as<uint64_t>(), n, maxT,
                    b->t_def.as<uint32_t>());
        const bool u2 = b->opt.search_ilp == 2;                     // tasks advanced together per quad; measured: 2 is no faster (36.1 vs 37.2 ms): VALU-bound
        KLAUNCH(b, "fm_search", (u2 ? k_search<2> : k_search<1>), search_grid(b, T, u2), 256, b->ix->d, b->rplanes.as<uint32_t>(),
                b->off.as<uint64_t>(), T, maxT, b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_ep.as<uint32_t>(),
                b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(), b->ctr.as<Counters>());
    }
    return 0;
}

static int stage_search(bwt_batch *b, bool lean) {
    const uint32_t n = b->n, maxT = b->maxT;
    const uint64_t T = (uint64_t)n * maxT;
    int rc;
    if ((rc = b->t_def.ensure(T * 4 + 4)) || (rc = b->t_sp.ensure(T * 4 + 4)) || (rc = b->t_ep.ensure(T * 4 + 4)) ||
        (rc = b->t_info.ensure(T * 4 + 4)) || (rc = b->t_cnt.ensure(T * 4 + 4)))
        return rc;
    HIPCHK(hipMemsetAsync(b->ctr.p, 0, sizeof(Counters), b->stream));
    if (n == 0) return 0;
    if ((rc = read_planes(b))) return rc;
    if ((rc = launch_search(b, n, maxT, false, lean))) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

static int stage_cluster(bwt_batch *b, bool to_extend) {
    const uint32_t n = b->n, maxT = b->maxT;
    MailGuard mail_guard{b};
    int rc;
    b->desc_partial = false;
    if ((rc = b->r_P.ensure((size_t)n * 4 + 4)) || (rc = b->r_qty.ensure((size_t)n * 4 + 4)) ||
        (rc = b->r_K.ensure((size_t)n * 4 + 4)) || (rc = b->bigq.ensure((size_t)n * 4 + 4)) ||
        (rc = b->mid2q.ensure((size_t)n * 4 + 4)) || (rc = b->mid4q.ensure((size_t)n * 4 + 4)) ||
        (rc = b->seed_off.ensure(((size_t)n + 1) * 8)) || (rc = b->clus_off.ensure(((size_t)n + 1) * 8)))
        return rc;
    if (n == 0) {
        b->n_cand = 0;
        HIPCHK(hipMemsetAsync(b->clus_off.p, 0, 8, b->stream));
        return fetch_counters(b);
    }
    if (!b->counts_done)
        KLAUNCH(b, "read_counts", k_read_counts, std::min<uint32_t>(cdiv(n, 256), 2048), 256, n, maxT, b->t_info.as<uint32_t>(),
                b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->bigq.as<uint32_t>(),
                b->mid2q.as<uint32_t>(), b->mid4q.as<uint32_t>(), b->ctr.as<Counters>());
    uint64_t total_seeds = 0, ncand = 0;
    bool all_clustered = false;
    if (b->fused) {
        // the seed kernel clustered what it could: the per-read seed slices (seed_off, pos) are only needed if reads are left.
        // Most batches leave none: the candidate offsets are scanned before the counters come back, so that such a batch needs
        // one synchronisation here, not two.
        if ((rc = dev_scan(b, b->r_K.as<uint32_t>(), b->clus_off.as<uint64_t>(), n))) return rc;
        if ((rc = mail(b, &ncand, b->clus_off.as<uint64_t>() + n, 8))) return rc;
        if ((rc = fetch_counters(b))) return rc;                               // sync #1
        all_clustered = !(b->hctr.n_small || b->hctr.n_mid2 || b->hctr.n_mid4 || b->hctr.n_big);
        if (!all_clustered) {
            if ((rc = dev_scan(b, b->r_P.as<uint32_t>(), b->seed_off.as<uint64_t>(), n))) return rc;
            if ((rc = mail(b, &total_seeds, b->seed_off.as<uint64_t>() + n, 8)) || (rc = mail_sync(b))) return rc;
        }
    } else {
        if ((rc = dev_scan(b, b->r_P.as<uint32_t>(), b->seed_off.as<uint64_t>(), n))) return rc;
        if ((rc = mail(b, &total_seeds, b->seed_off.as<uint64_t>() + n, 8))) return rc;
        if ((rc = fetch_counters(b))) return rc;                               // sync #1
    }
    if (b->hctr.err_flags & 1) return fail(BWT_EOVERFLOW, "a read produced more than 2^31 seeds");
    if (total_seeds > 0xfffffff0ull) return fail(BWT_EOVERFLOW, "batch produced more than 2^32 seeds: split it");
    if ((rc = b->pos.ensure(total_seeds * 4 + 16))) return rc;
    b->n_small = b->hctr.n_small;
    if (all_clustered) {
    } else if (b->fused && b->hctr.n_small < n / 8) {
        // few reads left for the wave-per-read kernel: list them first (a wave per read of the batch costs ~1 ms in launches alone)
        if (b->hctr.n_small) {
            if ((rc = b->smallq.ensure((size_t)b->hctr.n_small * 4 + 64))) return rc;
            unsigned int *cnt = &b->ctr.as<Counters>()->n_small;           // recounted by the listing kernel
            HIPCHK(hipMemsetAsync(cnt, 0, 4, b->stream));
            KLAUNCH(b, "cluster_small", k_collect_small, std::min<uint32_t>(cdiv(n, 256), 256), 256, n, b->r_P.as<uint32_t>(),
                    b->r_qty.as<uint32_t>(), b->smallq.as<uint32_t>(), cnt, b->hctr.n_small, b->ctr.as<Counters>());
            KLAUNCH(b, "cluster_small", k_cluster_small, cdiv((uint64_t)b->hctr.n_small * 64, 256), 256, b->ix->d, b->hctr.n_small, maxT,
                    b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(),
                    b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(), b->pos.as<uint32_t>(),
                    b->r_K.as<uint32_t>(), b->smallq.as<uint32_t>(), cnt);
        }
    } else
        KLAUNCH(b, "cluster_small", k_cluster_small, cdiv((uint64_t)n * 64, 256), 256, b->ix->d, n, maxT,
                b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(),
                b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(), b->pos.as<uint32_t>(),
                b->r_K.as<uint32_t>(), (const uint32_t *)nullptr, (const unsigned int *)nullptr);
    if (b->hctr.n_mid2 && !all_clustered)
        KLAUNCH(b, "cluster_wave2", k_cluster_wave<2>, cdiv((uint64_t)b->hctr.n_mid2 * 64, 256), 256, b->ix->d, b->hctr.n_mid2,
                b->mid2q.as<uint32_t>(), maxT, b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(),
                b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(),
                b->pos.as<uint32_t>(), b->r_K.as<uint32_t>(), (const unsigned int *)nullptr, 0u, 0xffffffffu);
    if (b->hctr.n_mid4)
        KLAUNCH(b, "cluster_wave4", k_cluster_wave<4>, cdiv((uint64_t)b->hctr.n_mid4 * 64, 256), 256, b->ix->d, b->hctr.n_mid4,
                b->mid4q.as<uint32_t>(), maxT, b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(),
                b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(),
                b->pos.as<uint32_t>(), b->r_K.as<uint32_t>(), (const unsigned int *)nullptr, 0u, 0xffffffffu);
    if (b->hctr.n_big) {
        // a block per read; a read of more seeds than its LDS holds is listed for k_cluster_huge (1024 threads, 124 KB of LDS per read)
        if ((rc = b->hugeq.ensure((size_t)b->hctr.n_big * 4 + 4))) return rc;
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
        Counters *dc = b->ctr.as<Counters>();
        if (maxT <= 64) {
            // up to 1024 seeds: a wave per read, sorted in registers (8 / 16 elements per lane)
            KLAUNCH(b, "cluster_big", k_cluster_wave<8>, cdiv((uint64_t)b->hctr.n_big * 64, 256), 256, b->ix->d, b->hctr.n_big,
                    b->bigq.as<uint32_t>(), maxT, b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(),
                    b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(),
                    b->pos.as<uint32_t>(), b->r_K.as<uint32_t>(), (const unsigned int *)nullptr, 256u, 512u);
            KLAUNCH(b, "cluster_big", k_cluster_wave<16>, cdiv((uint64_t)b->hctr.n_big * 64, 256), 256, b->ix->d, b->hctr.n_big,
                    b->bigq.as<uint32_t>(), maxT, b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(),
                    b->t_cnt.as<uint32_t>(), b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(),
                    b->pos.as<uint32_t>(), b->r_K.as<uint32_t>(), (const unsigned int *)nullptr, 512u, 1024u);
        }
        KLAUNCH(b, "cluster_big", (k_cluster_big<256, BIG_LDS>), std::min<uint32_t>(b->hctr.n_big, (uint32_t)cus * 8), 256, b->ix->d, maxT, b->bigq.as<uint32_t>(),
                b->t_def.as<uint32_t>(), b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(), b->t_cnt.as<uint32_t>(),
                b->r_P.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->seed_off.as<uint64_t>(), b->pos.as<uint32_t>(),
                b->r_K.as<uint32_t>(), (const unsigned int *)nullptr, b->hctr.n_big, b->hugeq.as<uint32_t>(), &dc->n_huge, maxT <= 64 ? 1024u : 0u, 0xffffffffu);
        const size_t huge_lds = (size_t)(HUGE_LDS + 1024) * 4;
        HIPCHK(hipFuncSetAttribute((const void *)k_cluster_huge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)huge_lds));
        {
            ProfScope ps(b, "cluster_huge");
            hipLaunchKernelGGL(k_cluster_huge, dim3(std::min<uint32_t>(b->hctr.n_big, (uint32_t)cus)), dim3(1024), huge_lds, b->stream, b->ix->d, maxT,
                               (const uint32_t *)b->hugeq.as<uint32_t>(), (const unsigned int *)&dc->n_huge, (const uint32_t *)b->t_def.as<uint32_t>(),
                               (const uint32_t *)b->t_sp.as<uint32_t>(), (const uint32_t *)b->t_info.as<uint32_t>(), (const uint32_t *)b->t_cnt.as<uint32_t>(),
                               (const uint32_t *)b->r_P.as<uint32_t>(), (const uint32_t *)b->r_qty.as<uint32_t>(), (const uint64_t *)b->seed_off.as<uint64_t>(),
                               b->pos.as<uint32_t>(), b->r_K.as<uint32_t>());
        }
    }
    if (!all_clustered) {
        if ((rc = dev_scan(b, b->r_K.as<uint32_t>(), b->clus_off.as<uint64_t>(), n))) return rc;
        if ((rc = mail(b, &ncand, b->clus_off.as<uint64_t>() + n, 8))) return rc;
        if ((rc = fetch_counters(b))) return rc;                               // sync #2
    }
    if (b->hctr.err_flags & 8) return fail(BWT_EHIP, "internal: cluster queue count mismatch");
    if (ncand > 0xfffffff0ull) return fail(BWT_EOVERFLOW, "batch produced more than 2^32 candidates: split it");
    b->n_cand = ncand;
    const size_t nc = (size_t)ncand + 1;
    if ((rc = b->c_read.ensure(nc * 4)) || (rc = b->c_S.ensure(nc * 4)) || (rc = b->c_D.ensure(nc * 4)) ||
        (rc = b->c_qoff.ensure(nc * 8)) || (rc = b->c_qlen.ensure(nc * 4)) || (rc = b->c_woff.ensure(nc * 8)) ||
        (rc = b->c_wlen.ensure(nc * 4)))
        return rc;
    if (to_extend) {                                                          // the prefilter makes the candidates itself (run_sw)
        b->cand_pending = true;
        // every cluster came from the seed kernels: the widest window is known already (Lq + the widest cluster they staged), and
        // the DP kernels can be sized without waiting for the prefilter
        b->max_w_known = all_clustered;
        b->hctr.max_w = all_clustered ? b->max_lq + b->hctr.max_D : 0;
        b->hctr.max_lq = all_clustered ? b->max_lq : 0;
        return 0;
    }
    KLAUNCH(b, "candidates", k_candidates, std::min<uint32_t>(cdiv(n, 256), 2048), 256, n, b->off.as<uint64_t>(), b->r_K.as<uint32_t>(),
            b->seed_off.as<uint64_t>(), b->clus_off.as<uint64_t>(), b->pos.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->r_cl.as<uint32_t>(),
            b->c_read.as<uint32_t>(),
            b->c_S.as<uint32_t>(), b->c_D.as<uint32_t>(), b->c_qoff.as<uint64_t>(), b->c_qlen.as<uint32_t>(),
            b->c_woff.as<int64_t>(), b->c_wlen.as<uint32_t>(), b->ctr.as<Counters>());
    HIPCHK(hipGetLastError());
    return fetch_counters(b);                                                  // sync #3 (max_w for the slabs)
}

// runs the integer kernel over all candidates, then the fp64 kernel over those that need it
// Extend stage: SW_SCORE over every candidate (boundary row in LDS, no direction matrix), then
// SW_TRACE over the candidates whose alignment is not the plain diagonal (and those too wide for the
// LDS pass), then SW_FP over the candidates that met a base outside ACGT.
static int run_sw(bwt_batch *b, const uint8_t *qbuf, const uint8_t *rbuf, uint64_t rbuf_len, uint32_t max_w,
                  uint32_t max_lq, const uint32_t *qplanes, const uint32_t *wplanes, bool first) {
    const uint64_t nc = b->n_cand;
    int rc;
    if ((rc = b->c_status.ensure(nc * 4 + 4)) || (rc = b->c_score.ensure(nc * 4 + 4)) ||
        (rc = b->c_ciglen.ensure(nc * 4 + 4)) || (rc = b->fpq.ensure(nc * 4 + 4)) || (rc = b->tq.ensure(nc * 4 + 4)) ||
        (rc = b->dq.ensure(nc * 4 + 4)) || (rc = b->tq2.ensure(nc * 4 + 4)))
        return rc;
    if (nc == 0) { b->cand_pending = false; return 0; }
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, b->device);
    if ((rc = b->cig.ensure(nc * b->cig_stride + 16))) return rc;
    // the queue counters: zero since the stage that started this run zeroed the whole block; a retry (wider CIGAR slots) resets them
    if (!first) {
        HIPCHK(hipMemsetAsync(&b->ctr.as<Counters>()->fp_count, 0, 16, b->stream));     // fp_count, tq_count, dq_count, tq2_count
        HIPCHK(hipMemsetAsync(&b->ctr.as<Counters>()->band_in, 0, 48, b->stream));      // band_*, pf_ov, band2_*, band3_*
        HIPCHK(hipMemsetAsync(&b->ctr.as<Counters>()->band0_in, 0, 32, b->stream));     // band0_*, dq_lo_count, cert_*
    }
    SwIO io;
    io.qbuf = qbuf; io.rbuf = rbuf; io.rbuf_len = rbuf_len;
    io.c_qoff = b->c_qoff.as<uint64_t>(); io.c_qlen = b->c_qlen.as<uint32_t>();
    io.c_woff = b->c_woff.as<int64_t>(); io.c_wlen = b->c_wlen.as<uint32_t>();
    io.c_status = b->c_status.as<int32_t>(); io.c_score = b->c_score.as<int32_t>();
    io.c_ciglen = b->c_ciglen.as<uint32_t>(); io.cig = b->cig.as<uint8_t>(); io.cig_stride = b->cig_stride;
    io.fpq = b->fpq.as<uint32_t>(); io.tq = b->tq.as<uint32_t>(); io.ctr = b->ctr.as<Counters>();
    io.dq = nullptr; io.tq2 = b->tq2.as<uint32_t>(); io.qplanes = qplanes; io.wplanes = wplanes; io.wave_upto = 0;
    io.tq_in = nullptr; io.tq_in_count = nullptr;
    // the banded tiers run on index-backed batches whose reads fit the kernel; the prefilter then keeps a queue of its own for the
    // candidates of the 12-diagonal tier (bwt_options.sw_band12 = 0: no such tier, everything starts at 20 diagonals)
    const uint32_t lq_bound = b->ix ? b->max_lq : max_lq;
    const bool band_on = qplanes && wplanes && b->opt.sw_prefilter != 0 && lq_bound >= 2 && lq_bound <= SWB_MAX_LQ && b->opt.sw_band != 0;
    const bool band12 = band_on && b->ix && b->opt.sw_band12 != 0;
    // ... and in front of that tier the certificate kernels (bwt_options.sw_cert = 0: none; 2: the 11-diagonal instance only, which takes
    // up to six mismatches; else also the 19-diagonal one for seven to ten, whose candidates the prefilter then queues with the others)
    const bool cert_on = band12 && b->opt.sw_cert != 0 && lq_bound <= 160;
    const bool cert_wide = cert_on && b->opt.sw_cert != 2;
    if (band12) {
        if ((rc = b->dq_lo.ensure(nc * 4 + 4)) || (rc = b->bq0.ensure(nc * 4 + 4)) || (rc = b->bqs.ensure(nc * 4 + 4))) return rc;
        io.dq_lo = b->dq_lo.as<uint32_t>();
        io.lo_max_m = cert_wide ? 10u : 6u;
    }
    if (qplanes && wplanes && b->opt.sw_prefilter != 0) {
        io.dq = b->dq.as<uint32_t>();
        CandSrc src{};
        if (b->cand_pending) {
            // quad per READ: the window arithmetic of k_candidates happens on the way, the candidate arrays are written once
            src.off = b->off.as<uint64_t>(); src.r_K = b->r_K.as<uint32_t>(); src.seed_off = b->seed_off.as<uint64_t>();
            src.clus_off = b->clus_off.as<uint64_t>(); src.pos = b->pos.as<uint32_t>(); src.r_qty = b->r_qty.as<uint32_t>();
            src.r_cl = b->r_cl.as<uint32_t>(); src.c_S = b->c_S.as<uint32_t>(); src.c_D = b->c_D.as<uint32_t>(); src.c_qoff = b->c_qoff.as<uint64_t>();
            src.c_qlen = b->c_qlen.as<uint32_t>(); src.c_woff = b->c_woff.as<int64_t>(); src.c_wlen = b->c_wlen.as<uint32_t>();
            // grid = what is resident at once (the passes are uniform: a second, partly filled round of blocks would cost a quarter)
            if (!b->occ_prefilter) {
                int per_cu = 5;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_sw_prefilter<true>, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 5; }
                b->occ_prefilter = std::max(1, std::min(per_cu, 8));
                if (b->opt.pf_blocks_per_cu > 0) b->occ_prefilter = std::max(1, std::min((int)b->opt.pf_blocks_per_cu, b->occ_prefilter));   // experiments
            }
            // the candidates past the first PF_KCAP of a read are listed by the fused pass and tested by a second one, quad per candidate
            if ((rc = b->pfq.ensure(nc * 4 + 256))) return rc;
            Counters *dc = b->ctr.as<Counters>();
            src.ovq = b->pfq.as<uint32_t>(); src.ov_count = &dc->pf_ov;
            src.ulen = b->ulen;
            if (b->opt.pf_kcap >= 0) src.kcap = (uint32_t)b->opt.pf_kcap + 1u;
            const uint32_t pf_grid = (uint32_t)std::min<uint64_t>(cdiv((uint64_t)b->n * 4, 256), (uint64_t)dev_cus * b->occ_prefilter);
            KLAUNCH(b, "sw_prefilter", k_sw_prefilter<true>, pf_grid, 256, io, b->n, src);
            {
                CandSrc rest{};
                rest.list = b->pfq.as<uint32_t>(); rest.list_count = &dc->pf_ov;
                KLAUNCH(b, "sw_prefilter", k_sw_prefilter<false>, (uint32_t)dev_cus * (uint32_t)b->occ_prefilter, 256, io, (uint32_t)nc, rest);
            }
            b->cand_pending = false;
            b->desc_partial = true;        // read offset / length and window offset exist only for the candidates that were queued
            if (!b->max_w_known && (rc = fetch_counters(b))) return rc;    // max_w / max_lq size the DP kernels' LDS and slabs
            max_w = b->hctr.max_w; max_lq = b->hctr.max_lq;
        } else {
            if (b->desc_partial) {
                // a retry (wider CIGAR slots) after a fused first attempt: every candidate's descriptor is needed now; k_candidates
                // writes them all (its totals go to a scratch block: the first attempt counted them already)
                if ((rc = b->scan_tmp.ensure(sizeof(Counters) + 64))) return rc;
                HIPCHK(hipMemsetAsync(b->scan_tmp.p, 0, sizeof(Counters), b->stream));
                KLAUNCH(b, "candidates", k_candidates, std::min<uint32_t>(cdiv(b->n, 256), 2048), 256, b->n, b->off.as<uint64_t>(), b->r_K.as<uint32_t>(),
                        b->seed_off.as<uint64_t>(), b->clus_off.as<uint64_t>(), b->pos.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->r_cl.as<uint32_t>(),
                        b->c_read.as<uint32_t>(), b->c_S.as<uint32_t>(), b->c_D.as<uint32_t>(), b->c_qoff.as<uint64_t>(), b->c_qlen.as<uint32_t>(),
                        b->c_woff.as<int64_t>(), b->c_wlen.as<uint32_t>(), b->scan_tmp.as<Counters>());
                b->desc_partial = false;
            }
            KLAUNCH(b, "sw_prefilter", k_sw_prefilter<false>, std::min<uint32_t>(cdiv(nc * 4, 256), 2048), 256, io, (uint32_t)nc, src);   // quad per candidate
        }
    }
    // A packed batch keeps no ASCII copy of its reads.  The kernels of the generic DP path read bytes (k_sw_wave, k_sw_score,
    // k_sw<...>): each gets, right in front of it, the bytes of the candidates queued for it, made from the planes; with no
    // prefilter (no queues) the whole batch is expanded once.
    if (!b->ascii_valid && !io.dq) { if ((rc = ascii_all(b))) return rc; }
    const bool sparse_ascii = !b->ascii_valid;
    auto ascii_for = [&](const uint32_t *queue, const unsigned int *count, uint32_t limit) {
        if (!sparse_ascii) return;
        ProfScope ps(b, "ascii_reads");
        hipLaunchKernelGGL(k_ascii_reads, dim3((uint32_t)dev_cus * 8), dim3(256), 0, b->stream, queue, count, limit, (const uint64_t *)io.c_qoff,
                           (const uint32_t *)io.c_qlen, qplanes, b->bases.as<uint8_t>());
    };
    const uint32_t maxW1 = max_w + 1, strips = (max_lq + SW_R - 1) / SW_R;

    // a short DP queue (what a batch of clean reads leaves after the prefilter) goes to the wave-per-alignment kernel:
    // lane-per-alignment work would run a handful of waves for the full serial length of a DP
    const uint32_t wv_dwords = swv_dwords(max_w), wv_w1 = max_w + 1;
    const uint64_t wv_lds = ((uint64_t)wv_w1 + (uint64_t)max_lq * wv_dwords) * 4 * 4;            // 4 waves per block
    const uint32_t wv_thresh = (io.dq && wv_lds <= 60 * 1024 && b->opt.sw_wave != 0) ? 32768u : 0u;
    io.wave_upto = wv_thresh;
    b->wave_upto = wv_thresh;
    // the DP queue first goes through the banded kernel: what its certificate covers is settled there (traceback included),
    // the rest -- candidates far from any clean alignment, wide clusters -- moves on in a second queue that takes dq's place
    b->band_ran = false;
    const uint32_t band_min = std::min<uint32_t>(wv_thresh, b->opt.sw_band_min >= 0 ? (uint32_t)b->opt.sw_band_min : 16384u);   // queues up to this size are left to k_sw_wave
    if (io.dq && band_on && max_lq >= 2 && max_lq <= SWB_MAX_LQ) {
        const uint32_t rows = max_lq + 1;
        // resident grids: 4 (3 for the wide tiers) waves per SIMD; the direction slab (rows x 1 KiB per wave, 2 KiB in the third tier)
        // stays under 1 GiB
        const uint64_t cap_blocks = (1ull << 30) / ((uint64_t)4 * rows * 4 * 64 * 4);
        const uint32_t blocks1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(cdiv(nc, 512), (uint64_t)dev_cus * 4), cap_blocks));
        const uint32_t blocks2 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(cdiv(nc, 512), (uint64_t)dev_cus * 3), cap_blocks));
        const uint32_t blocks3 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks2, cap_blocks * 4 / ((SWB_B3 + 7) / 8)));
        if ((rc = b->bq.ensure(nc * 4 + 4)) || (rc = b->bq2.ensure(nc * 4 + 4)) || (rc = b->bq3.ensure(nc * 4 + 4)) ||
            (rc = b->slab_band.ensure(std::max<uint64_t>((uint64_t)std::max(blocks1, blocks2) * 4 * rows * 4 * 64 * 4, (uint64_t)blocks3 * 4 * rows * ((SWB_B3 + 7) / 8) * 64 * 4))))
            return rc;
        Counters *dc = b->ctr.as<Counters>();
        // the 12-diagonal tier's queue: the prefilter's dq_lo -- after the certificate kernel has taken out of it what needs no DP at
        // all (bwt_options.sw_cert = 0: nothing)
        const uint32_t *q12 = b->dq_lo.as<uint32_t>();
        const unsigned int *n12 = &dc->dq_lo_count;
        if (cert_on) {
            if ((rc = b->cq.ensure(nc * 4 + 4)) || (rc = b->cq2.ensure(nc * 4 + 4))) return rc;
            ProfScope pc(b, "sw_cert");
            const uint32_t cblocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cdiv(nc, 256), (uint64_t)dev_cus * 8));
            uint32_t *const over = cert_wide ? b->cq2.as<uint32_t>() : nullptr;
            if (lq_bound <= 128)
                hipLaunchKernelGGL((k_sw_cert<5, 4>), dim3(cblocks), dim3(256), 0, b->stream, io, q12, n12, b->cq.as<uint32_t>(), &dc->cert_left, over,
                                   &dc->cert2_in, &dc->cert_done);
            else
                hipLaunchKernelGGL((k_sw_cert<5, 5>), dim3(cblocks), dim3(256), 0, b->stream, io, q12, n12, b->cq.as<uint32_t>(), &dc->cert_left, over,
                                   &dc->cert2_in, &dc->cert_done);
            q12 = b->cq.as<uint32_t>(); n12 = &dc->cert_left;
            // the wider instance on what had seven to ten mismatches; what it cannot settle joins the 20-diagonal tier's own queue
            if (cert_wide) {
                if (lq_bound <= 128)
                    hipLaunchKernelGGL((k_sw_cert<9, 4>), dim3(cblocks), dim3(256), 0, b->stream, io, (const uint32_t *)b->cq2.as<uint32_t>(),
                                       (const unsigned int *)&dc->cert2_in, io.dq, &dc->dq_count, (uint32_t *)nullptr, (unsigned int *)nullptr, &dc->cert_done);
                else
                    hipLaunchKernelGGL((k_sw_cert<9, 5>), dim3(cblocks), dim3(256), 0, b->stream, io, (const uint32_t *)b->cq2.as<uint32_t>(),
                                       (const unsigned int *)&dc->cert2_in, io.dq, &dc->dq_count, (uint32_t *)nullptr, (unsigned int *)nullptr, &dc->cert_done);
            }
        }
        {
            ProfScope ps(b, "sw_band");
            const uint32_t *nullq = nullptr;
            const unsigned int *nullc = nullptr;
            // the 12-diagonal tier on the prefilter's queue of square windows with <= 6 mismatches: what it does not settle goes on to
            // the 20-diagonal tier's second queue (bq0 / band0_left), what that one cannot settle either to the 32-diagonal one's (bqs)
            if (band12)
                hipLaunchKernelGGL((k_sw_band<12, 20>), dim3(blocks1), dim3(256), 0, b->stream, io, q12, n12, nullq, nullc, b->bq0.as<uint32_t>(),
                                   &dc->band0_in, b->bqs.as<uint32_t>(), &dc->band0_direct, b->slab_band.as<uint32_t>(), rows, band_min);
            // first tier: what the second cannot settle either goes straight to the third's second queue (bq3 / band_direct)
            hipLaunchKernelGGL((k_sw_band<20, 32>), dim3(blocks1), dim3(256), 0, b->stream, io, (const uint32_t *)io.dq, (const unsigned int *)&dc->dq_count,
                               band12 ? (const uint32_t *)b->bq0.as<uint32_t>() : nullq, band12 ? (const unsigned int *)&dc->band0_left : nullc,
                               b->bq.as<uint32_t>(), &dc->band_in, b->bq3.as<uint32_t>(), &dc->band_direct,
                               b->slab_band.as<uint32_t>(), rows, band_min);
            hipLaunchKernelGGL((k_sw_band<32, 0>), dim3(blocks2), dim3(256), 0, b->stream, io, (const uint32_t *)b->bq.as<uint32_t>(),
                               (const unsigned int *)&dc->band_left, band12 ? (const uint32_t *)b->bqs.as<uint32_t>() : nullq,
                               band12 ? (const unsigned int *)&dc->band0_direct : nullc, b->bq2.as<uint32_t>(), &dc->band2_in, (uint32_t *)nullptr,
                               (unsigned int *)nullptr, b->slab_band.as<uint32_t>(), rows, band_min);
            // (the first tier's output queue is free again: the third tier passes on into it; bwt_options.sw_band = 2 makes it
            // pass everything on -- the two-tier path, for comparisons)
            hipLaunchKernelGGL((k_sw_band<SWB_B3, 0>), dim3(blocks3), dim3(256), 0, b->stream, io, (const uint32_t *)b->bq2.as<uint32_t>(),
                               (const unsigned int *)&dc->band2_left, (const uint32_t *)b->bq3.as<uint32_t>(), (const unsigned int *)&dc->band_direct,
                               b->bq.as<uint32_t>(), &dc->band3_in, (uint32_t *)nullptr, (unsigned int *)nullptr, b->slab_band.as<uint32_t>(), rows,
                               b->opt.sw_band == 2 ? 0xffffffffu : band_min);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&dc->dq_count, &dc->band3_left, 4, hipMemcpyDeviceToDevice, b->stream));
        io.dq = b->bq.as<uint32_t>();
        b->band_ran = true;
    }

    const bool pk_ok_early = (max_lq == 100 || max_lq == 150) && b->opt.sw_pk != 0 && b->opt.sw_lds != 0;
    // (the packed-i16 kernels read planes; the other lane-per-alignment kernels and the wave kernel read bytes)
    if (io.dq) ascii_for(io.dq, &b->ctr.as<Counters>()->dq_count, pk_ok_early ? wv_thresh : 0xffffffffu);
    if (wv_thresh) {
        ProfScope ps(b, "sw_wave");
        hipLaunchKernelGGL(k_sw_wave, dim3((uint32_t)std::min<uint64_t>(cdiv(std::min<uint64_t>(nc, wv_thresh), 4), (uint64_t)dev_cus * 4)),
                           dim3(256), (size_t)wv_lds, b->stream, io, wv_thresh, wv_w1, wv_dwords, max_lq, (const uint32_t *)nullptr,
                           (const unsigned int *)nullptr, (uint32_t *)nullptr, (unsigned int *)nullptr);
    }

    // pass 1: bit-planes + 16-bit packed boundary row in LDS (k_sw_score); windows wider than the cap and
    // non-diagonal alignments go on to the trace pass.  LDS per wave: planes + (wcap+1) x 128 B.
    const bool fits16 = 4ull * (2ull * max_lq + max_w + 4) < 32000ull;
    const bool lds_off = b->opt.sw_lds == 0;             // experiments: the HBM boundary slab instead of LDS
    const uint32_t wcap = std::min<uint32_t>(max_w, 254), lqcap = std::min<uint32_t>(max_lq, 254);
    const uint32_t pwords = 2 * ((wcap + 31) / 32) + 2 * ((lqcap + 31) / 32);
    const uint64_t lds_wave = ((uint64_t)pwords * 64 + ((uint64_t)(wcap + 1) * 64 + 1) / 2) * 4;
    const uint64_t lds_bytes = lds_wave * 4;
    const bool pk_ok = (max_lq == 100 || max_lq == 150) && b->opt.sw_pk != 0 && !lds_off;
    if (pk_ok) {
        // two alignments per lane, whole query column in registers; LDS holds only the bit-planes
        const uint32_t qw = (max_lq + 31) / 32;
        const uint64_t lds_pk = (uint64_t)(2 * ((wcap + 31) / 32) + 2 * qw) * 64 * 4 * 2 * 4;     // 2 sets x 4 waves
        const uint32_t per_cu = 2;                                                             // VGPR-limited (181 / 247 VGPRs)
        const uint32_t *const NOQ = nullptr; const unsigned int *const NOC = nullptr;          // (k_sw_pk's queue overrides: none)
        uint32_t *const NOOQ = nullptr; unsigned int *const NOOC = nullptr;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(cdiv(nc, 512), (uint64_t)dev_cus * per_cu);
        // After the prefilter the DP queue holds hardly any plain diagonal (those with <= 2 mismatches in a square window are
        // gone): a score-only pass would hand most of it on to the traced pass anyway (measured, 150 bp reads with indels:
        // 14.7 + 17.6 ms against 26.8 ms for tracing everything), so the traced DP runs on the DP queue directly.
        const bool trace_all = io.dq && b->opt.sw_trace_all != 0;     // 0 = score pass first
        if (trace_all) {
            io.tq2 = io.dq;
            HIPCHK(hipMemcpyAsync(&b->ctr.as<Counters>()->tq2_count, &b->ctr.as<Counters>()->dq_count, 4, hipMemcpyDeviceToDevice, b->stream));
        } else {
            ProfScope ps(b, "sw_extend");
            if (max_lq == 100)
                hipLaunchKernelGGL((k_sw_pk<100, false>), dim3(blocks), dim3(256), (size_t)lds_pk, b->stream, io, (uint32_t)nc, wcap,
                                   (uint32_t *)nullptr, 0u, wv_thresh, NOQ, NOC, NOOQ, NOOC);
            else
                hipLaunchKernelGGL((k_sw_pk<150, false>), dim3(blocks), dim3(256), (size_t)lds_pk, b->stream, io, (uint32_t)nc, wcap,
                                   (uint32_t *)nullptr, 0u, wv_thresh, NOQ, NOC, NOOQ, NOOC);
        }
        // packed traceback pass over the alignments that are not the plain diagonal: per-wave direction slab
        const uint32_t ndg = (max_lq + 7) / 8, w1 = wcap + 1;
        const uint64_t per_wave_pk = (uint64_t)w1 * ndg * 64 * 4;
        uint32_t tblocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, (4ull << 30) / (per_wave_pk * 4)));
        if ((rc = b->slab_dir_pk.ensure((uint64_t)tblocks * 4 * per_wave_pk))) return rc;
        {
            ProfScope ps(b, trace_all ? "sw_extend" : "sw_trace_pk");
            if (max_lq == 100)
                hipLaunchKernelGGL((k_sw_pk<100, true>), dim3(tblocks), dim3(256), (size_t)lds_pk, b->stream, io, (uint32_t)nc, wcap,
                                   b->slab_dir_pk.as<uint32_t>(), w1, wv_thresh, NOQ, NOC, NOOQ, NOOC);
            else
                hipLaunchKernelGGL((k_sw_pk<150, true>), dim3(tblocks), dim3(256), (size_t)lds_pk, b->stream, io, (uint32_t)nc, wcap,
                                   b->slab_dir_pk.as<uint32_t>(), w1, wv_thresh, NOQ, NOC, NOOQ, NOOC);
        }
        // the WIDE pass: windows past the cap above (clusters that chain along tandem copies: several hundred bases) went to the
        // generic trace queue; they run through the same kernel once more with as much window as 160 KB of LDS holds planes for --
        // one block per CU -- and only what is wider still is left for the lane-per-alignment kernel
        if (max_w > wcap && b->opt.sw_pk != 2) {
            const uint32_t pw_max = (uint32_t)((160 * 1024) / (64 * 4 * 2 * 4) / 2) - qw;             // plane words per window that fit
            const uint32_t wcap2 = std::min<uint32_t>(max_w, pw_max * 32);
            const uint64_t lds_pk2 = (uint64_t)(2 * ((wcap2 + 31) / 32) + 2 * qw) * 64 * 4 * 2 * 4;
            const uint32_t w12 = wcap2 + 1;
            const uint64_t per_wave2 = (uint64_t)w12 * ndg * 64 * 4;
            const uint32_t wblocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(cdiv(nc, 512), (uint64_t)dev_cus), (2ull << 30) / (per_wave2 * 4)));
            if ((rc = b->slab_dir_pk.ensure((uint64_t)wblocks * 4 * per_wave2)) || (rc = b->tqw.ensure(nc * 4 + 4))) return rc;
            Counters *dc = b->ctr.as<Counters>();
            ProfScope ps(b, "sw_trace_wide");
            if (max_lq == 100) {
                HIPCHK(hipFuncSetAttribute((const void *)k_sw_pk<100, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pk2));
                hipLaunchKernelGGL((k_sw_pk<100, true>), dim3(wblocks), dim3(256), (size_t)lds_pk2, b->stream, io, (uint32_t)nc, wcap2,
                                   b->slab_dir_pk.as<uint32_t>(), w12, 0u, (const uint32_t *)io.tq, (const unsigned int *)&dc->tq_count,
                                   b->tqw.as<uint32_t>(), &dc->tqw_count);
            } else {
                HIPCHK(hipFuncSetAttribute((const void *)k_sw_pk<150, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pk2));
                hipLaunchKernelGGL((k_sw_pk<150, true>), dim3(wblocks), dim3(256), (size_t)lds_pk2, b->stream, io, (uint32_t)nc, wcap2,
                                   b->slab_dir_pk.as<uint32_t>(), w12, 0u, (const uint32_t *)io.tq, (const unsigned int *)&dc->tq_count,
                                   b->tqw.as<uint32_t>(), &dc->tqw_count);
            }
            io.tq_in = b->tqw.as<uint32_t>(); io.tq_in_count = &dc->tqw_count;
        }
    } else if (fits16 && lds_bytes <= 160 * 1024 && !lds_off) {
        uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(8, (uint32_t)((160 * 1024) / lds_bytes)));
        if (b->opt.sw_blocks_per_cu > 0) per_cu = std::min<uint32_t>(per_cu, (uint32_t)b->opt.sw_blocks_per_cu);
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(cdiv(nc, 256), (uint64_t)dev_cus * per_cu);
        if (lds_bytes > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)k_sw_score, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        ProfScope ps(b, "sw_extend");
        hipLaunchKernelGGL(k_sw_score, dim3(blocks), dim3(256), (size_t)lds_bytes, b->stream, io, (uint32_t)nc, wcap, lqcap);
    } else {
        const uint32_t wpc = b->opt.sw_waves_per_cu > 0 ? (uint32_t)b->opt.sw_waves_per_cu : 8;
        uint64_t waves = std::min<uint64_t>(cdiv(nc, 64), (uint64_t)dev_cus * wpc);
        waves = std::max<uint64_t>(4, std::min<uint64_t>(waves, (4ull << 30) / ((uint64_t)maxW1 * 256)) / 4 * 4);
        if ((rc = b->slab_bnd.ensure(waves * maxW1 * 64 * 4))) return rc;
        KLAUNCH(b, "sw_extend", (k_sw<SW_SCORE, false>), (uint32_t)(waves / 4), 256, io, (uint32_t)nc, maxW1, strips,
                b->slab_bnd.as<uint32_t>(), (double *)nullptr, (uint32_t *)nullptr);
    }
    // pass 2 / fp64: per-wave HBM slabs (boundary row + 2-bit directions); resident waves capped at ~4 GiB of slab
    const uint64_t per_wave = (uint64_t)maxW1 * 64 * 4 * (1 + strips);
    uint64_t twaves = std::min<uint64_t>(cdiv(nc, 64), (uint64_t)dev_cus * 8);
    twaves = std::max<uint64_t>(4, std::min<uint64_t>(twaves, (4ull << 30) / std::max<uint64_t>(per_wave, 1)) / 4 * 4);
    if ((rc = b->slab_bnd_t.ensure(twaves * maxW1 * 64 * 4)) || (rc = b->slab_dir.ensure(twaves * maxW1 * 64 * 4 * strips)))
        return rc;
    if (io.tq_in) ascii_for(io.tq_in, io.tq_in_count, 0xffffffffu);
    else ascii_for(io.tq, &b->ctr.as<Counters>()->tq_count, 0xffffffffu);
    KLAUNCH(b, "sw_trace", (k_sw<SW_TRACE, false>), (uint32_t)(twaves / 4), 256, io, (uint32_t)nc, maxW1, strips,
            b->slab_bnd_t.as<uint32_t>(), (double *)nullptr, b->slab_dir.as<uint32_t>());
    const uint64_t fwaves = std::max<uint64_t>(4, std::min<uint64_t>(twaves, 1024) / 4 * 4);
    if ((rc = b->slab_bnd_fp.ensure(fwaves * maxW1 * 64 * 4)) || (rc = b->slab_bndv.ensure(fwaves * maxW1 * 64 * 8)) ||
        (rc = b->slab_dir_fp.ensure(fwaves * maxW1 * 64 * 4 * strips)))
        return rc;
    ascii_for(io.fpq, &b->ctr.as<Counters>()->fp_count, 0xffffffffu);
    KLAUNCH(b, "sw_extend_fp64", (k_sw<SW_FP, false>), (uint32_t)(fwaves / 4), 256, io, (uint32_t)nc, maxW1, strips,
            b->slab_bnd_fp.as<uint32_t>(), b->slab_bndv.as<double>(), b->slab_dir_fp.as<uint32_t>());
    HIPCHK(hipGetLastError());
    return 0;
}

static int dev_scan3(bwt_batch *b, const uint32_t *in0, uint64_t *out0, const uint32_t *in1, uint64_t *out1, const uint32_t *in2,
                     uint64_t *out2, uint64_t n, const uint32_t *in3, uint64_t *out3);

static int stage_hits(bwt_batch *b, int32_t shift) {
    const uint32_t n = b->n;
    MailGuard mail_guard{b};
    int rc;
    b->expanded = false;
    if ((rc = b->r_nhit.ensure((size_t)n * 4 + 4)) || (rc = b->r_nx.ensure((size_t)n * 4 + 4)) || (rc = b->r_xcb.ensure((size_t)n * 4 + 4)) ||
        (rc = b->r_status.ensure((size_t)n + 4)) || (rc = b->hit_off.ensure(((size_t)n + 1) * 8)) ||
        (rc = b->x_off.ensure(((size_t)n + 1) * 8)) || (rc = b->xcb_off.ensure(((size_t)n + 1) * 8)))
        return rc;
    b->narrow = false; b->n_narrow = 0;
    // the narrow form: on request, and only where a score fits 16 bits (2 Lq <= 32766; a window offset always fits 32: an index has
    // fewer than 2^32 - 70000 rows)
    const bool nar = n && b->ix && b->opt.hits_narrow == 1 && b->max_lq <= 16383;
    if (nar && ((rc = b->r_nn.ensure((size_t)n * 4 + 4)) || (rc = b->nn_off.ensure(((size_t)n + 1) * 8)) ||
                (rc = b->n_woff.ensure((size_t)n * 4 + 16)) || (rc = b->n_score.ensure((size_t)n * 2 + 16))))
        return rc;
    if (n == 0) {                                   // empty batch: empty, but well-formed, hit tables
        b->n_hits = 0; b->cig_bytes = 0; b->n_extra = 0; b->xcig_bytes = 0;
        if ((rc = b->h_rec.ensure(16)) || (rc = b->x_rec.ensure(16)) || (rc = b->x_cig.ensure(16))) return rc;
        HIPCHK(hipMemsetAsync(b->hit_off.p, 0, 8, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return 0;
    }
    HIPCHK(hipMemsetAsync(&b->ctr.as<Counters>()->cig_total, 0, 8, b->stream));     // (a retry with wider CIGAR slots counts again)
    KLAUNCH(b, "hits_count", k_hits_count, std::min<uint32_t>(cdiv(n, 256), 2048), 256, n, b->clus_off.as<uint64_t>(), b->c_status.as<int32_t>(),
            b->c_ciglen.as<uint32_t>(), b->r_qty.as<uint32_t>(), b->r_nhit.as<uint32_t>(), b->r_nx.as<uint32_t>(), b->r_xcb.as<uint32_t>(),
            b->r_status.as<int8_t>(), b->ctr.as<Counters>(), nar ? b->r_nn.as<uint32_t>() : (uint32_t *)nullptr);
    if ((rc = dev_scan3(b, b->r_nhit.as<uint32_t>(), b->hit_off.as<uint64_t>(), b->r_nx.as<uint32_t>(), b->x_off.as<uint64_t>(),
                        b->r_xcb.as<uint32_t>(), b->xcb_off.as<uint64_t>(), n, nar ? (const uint32_t *)b->r_nn.as<uint32_t>() : nullptr,
                        nar ? b->nn_off.as<uint64_t>() : nullptr)))
        return rc;
    // the hit tables are sized by their bounds (a hit per candidate, a full text slot per hit), so that nothing has to come
    // back to the host before k_hits_write: the totals are read once, together with the counters, when everything is queued
    const size_t nh = (size_t)b->n_cand + 1;
    if ((rc = b->h_rec.ensure(nh * 16)) || (rc = b->x_rec.ensure(nh * 16)) || (rc = b->x_cig.ensure((size_t)b->n_cand * b->cig_stride + 16)))
        return rc;
    KLAUNCH(b, "hits_write", k_hits_write, cdiv(n, 256), 256, n, shift, b->clus_off.as<uint64_t>(),
            b->hit_off.as<uint64_t>(), b->x_off.as<uint64_t>(), b->xcb_off.as<uint64_t>(), b->r_status.as<int8_t>(), b->c_status.as<int32_t>(),
            b->c_score.as<int32_t>(), b->c_ciglen.as<uint32_t>(), b->c_S.as<uint32_t>(), b->c_wlen.as<uint32_t>(),
            b->cig.as<uint8_t>(), b->cig_stride, b->h_rec.as<HitRec>(), b->x_rec.as<HitExtra>(), b->x_cig.as<uint8_t>(),
            nar ? (const uint64_t *)b->nn_off.as<uint64_t>() : nullptr, b->n_woff.as<uint32_t>(), b->n_score.as<int16_t>());
    uint64_t tot[4] = {0, 0, 0, 0};
    if ((rc = mail(b, &tot[0], b->hit_off.as<uint64_t>() + n, 8)) || (rc = mail(b, &tot[1], b->x_off.as<uint64_t>() + n, 8)) ||
        (rc = mail(b, &tot[2], b->xcb_off.as<uint64_t>() + n, 8)) || (nar && (rc = mail(b, &tot[3], b->nn_off.as<uint64_t>() + n, 8))))
        return rc;
    if ((rc = fetch_counters(b))) return rc;                                   // sync
    HIPCHK(hipGetLastError());
    // HitExtra.hit and HitExtra.cig_off are 32-bit: a table whose hit count or extras text passes 4 GiB would wrap silently
    // (k_hits_write stores the low words of its 64-bit scan offsets).  n_cand is capped at 2^32 - 16, the text is not.
    if (tot[0] > 0xffffffffull || tot[2] > 0xffffffffull) {
        b->n_hits = 0; b->n_extra = 0; b->xcig_bytes = 0; b->cig_bytes = 0;
        return fail(BWT_EOVERFLOW, "batch produced more than 2^32-1 hits or bytes of CIGAR text in its compact table: split it");
    }
    if (tot[0] + tot[3] > 0xffffffffull) {
        b->n_hits = 0; b->n_extra = 0; b->xcig_bytes = 0; b->cig_bytes = 0;
        return fail(BWT_EOVERFLOW, "batch produced more than 2^32-1 hits: split it");
    }
    b->n_hits = tot[0]; b->n_extra = tot[1]; b->xcig_bytes = tot[2]; b->cig_bytes = b->hctr.cig_total;
    b->narrow = nar; b->n_narrow = tot[3];
    return 0;
}

// a table made in the narrow form back into the 16-byte form, in place of it (entry points that hand out records)
static int ensure_v1(bwt_batch *b) {
    if (!b->narrow) return 0;
    const uint32_t n = b->n;
    const uint64_t nh = b->n_hits + b->n_narrow;
    int rc;
    if ((rc = b->w_rec.ensure((nh + 1) * 16)) || (rc = b->w_xrec.ensure((b->n_extra + 1) * 16)) || (rc = b->w_hoff.ensure(((size_t)n + 1) * 8)) ||
        (rc = b->w_status.ensure((size_t)n + 4)))
        return rc;
    const int32_t shift = b->ix ? b->ix->shift : 0;
    KLAUNCH(b, "hits_widen", k_hits_widen, (uint32_t)std::min<uint64_t>(cdiv((uint64_t)n + 1, 256), 8192), 256, n, shift, (const int8_t *)b->r_status.as<int8_t>(),
            (const uint64_t *)b->hit_off.as<uint64_t>(), (const uint64_t *)b->nn_off.as<uint64_t>(), (const HitRec *)b->h_rec.as<HitRec>(),
            (const uint32_t *)b->n_woff.as<uint32_t>(), (const int16_t *)b->n_score.as<int16_t>(), b->w_rec.as<HitRec>(), b->w_hoff.as<uint64_t>(),
            b->w_status.as<int8_t>());
    if (b->n_extra)
        KLAUNCH(b, "hits_widen", k_extras_widen, (uint32_t)std::min<uint64_t>(cdiv(b->n_extra, 256), 8192), 256, b->n_extra, (const HitExtra *)b->x_rec.as<HitExtra>(),
                (const HitRec *)b->h_rec.as<HitRec>(), (const uint64_t *)b->nn_off.as<uint64_t>(), b->w_xrec.as<HitExtra>());
    HIPCHK(hipGetLastError());
    std::swap(b->h_rec, b->w_rec); std::swap(b->x_rec, b->w_xrec); std::swap(b->hit_off, b->w_hoff); std::swap(b->r_status, b->w_status);
    b->n_hits = nh; b->n_narrow = 0; b->narrow = false; b->expanded = false;
    return 0;
}

// the per-hit arrays of bwt_align_batch out of the compact table (kernels.hip.h: k_hits_expand), once per run, when a caller
// asks for them
static int ensure_expanded(bwt_batch *b) {
    if (b->expanded) return 0;
    int rc;
    if ((rc = ensure_v1(b))) return rc;
    const uint64_t nh = b->n_hits;
    if ((rc = b->h_woff.ensure((nh + 1) * 8)) || (rc = b->h_wlen.ensure((nh + 1) * 4)) || (rc = b->h_cigoff.ensure((nh + 2) * 8)) ||
        (rc = b->h_cig.ensure(b->cig_bytes + 16)) || (rc = b->h_xidx.ensure((nh + 1) * 4)) || (rc = b->h_len.ensure((nh + 1) * 4)))
        return rc;
    if (!nh) {
        HIPCHK(hipMemsetAsync(b->h_cigoff.p, 0, 8, b->stream));
        b->expanded = true;
        return 0;
    }
    if (nh > 0xfffffff0ull) return fail(BWT_EOVERFLOW, "more than 2^32 hits: split the batch");
    const int32_t shift = b->ix ? b->ix->shift : 0;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(nh, 256), 8192);
    HIPCHK(hipMemsetAsync(b->h_xidx.p, 0xff, nh * 4, b->stream));
    if (b->n_extra)
        KLAUNCH(b, "hits_expand", k_hits_xidx, (uint32_t)std::min<uint64_t>(cdiv(b->n_extra, 256), 8192), 256, (const HitExtra *)b->x_rec.as<HitExtra>(), b->n_extra,
                b->h_xidx.as<uint32_t>());
    KLAUNCH(b, "hits_expand", k_hits_len, grid, 256, nh, (const HitRec *)b->h_rec.as<HitRec>(), (const uint32_t *)b->h_xidx.as<uint32_t>(),
            (const HitExtra *)b->x_rec.as<HitExtra>(), (const uint64_t *)b->off.as<uint64_t>(), b->ulen, b->h_len.as<uint32_t>());
    if ((rc = dev_scan(b, b->h_len.as<uint32_t>(), b->h_cigoff.as<uint64_t>(), nh))) return rc;
    KLAUNCH(b, "hits_expand", k_hits_expand, grid, 256, nh, shift, (const HitRec *)b->h_rec.as<HitRec>(), (const uint32_t *)b->h_xidx.as<uint32_t>(),
            (const HitExtra *)b->x_rec.as<HitExtra>(), (const uint8_t *)b->x_cig.as<uint8_t>(), (const uint64_t *)b->off.as<uint64_t>(), b->ulen,
            (const uint64_t *)b->h_cigoff.as<uint64_t>(), b->h_woff.as<int64_t>(), b->h_wlen.as<int32_t>(), b->h_cig.as<uint8_t>());
    HIPCHK(hipGetLastError());
    b->expanded = true;
    return 0;
}

static int stage_extend(bwt_batch *b, const uint8_t *qbuf, const uint8_t *rbuf, uint64_t rbuf_len, uint32_t max_w,
                        uint32_t max_lq, int32_t shift, const uint32_t *qplanes = nullptr,
                        const uint32_t *wplanes = nullptr) {
    for (int attempt = 0; attempt < 8; attempt++) {
        if (attempt && b->ix) { max_w = b->hctr.max_w; max_lq = b->hctr.max_lq; }    // (known since the first attempt's prefilter)
        int rc = run_sw(b, qbuf, rbuf, rbuf_len, max_w, max_lq, qplanes, wplanes, attempt == 0);
        if (rc) return rc;
        // clear the sticky CIGAR-overflow flag before counting
        Counters *dc = b->ctr.as<Counters>();
        HIPCHK(hipMemsetAsync(&dc->err_flags, 0, 8, b->stream));
        if ((rc = stage_hits(b, shift))) return rc;
        if (b->hctr.err_flags & 4) return fail(BWT_EHIP, "internal: candidate left pending by the extend stage");
        if (b->hctr.err_flags & 16) return fail(BWT_EHIP, "internal: a DP slab was sized smaller than a candidate's window");
        if (!(b->hctr.err_flags & 2)) return 0;
        b->cig_stride *= 4;                                   // a CIGAR did not fit its slot: widen and redo
    }
    return fail(BWT_EHIP, "internal: CIGAR slot overflow persists");
}

extern "C" int bwt_batch_run(bwt_batch *b, int stages, bwt_batch_counts *counts) {
    if (!b || !b->ix) return fail(BWT_EINVAL, "bwt_batch_run: batch has no index");
    std::shared_lock<std::shared_mutex> lk(b->ix->mu);     // the index arrays stay put while the kernels run
    HIPCHK(hipSetDevice(b->device));
    int rc;
    if (stages & (BWT_STAGE_SEARCH | BWT_STAGE_CLUSTER | BWT_STAGE_EXTEND)) {
        // a run that goes straight through to the extend stage does not keep the task rows of reads the seed kernel clusters itself
        if ((rc = stage_search(b, (stages & BWT_STAGE_EXTEND) != 0))) return rc;
        b->done = BWT_STAGE_SEARCH;
    }
    if (stages & (BWT_STAGE_CLUSTER | BWT_STAGE_EXTEND)) {
        {
            const bool to_extend = (stages & BWT_STAGE_EXTEND) && b->opt.sw_prefilter != 0 && b->opt.sw_fuse_cand != 0;   // 0 = separate k_candidates
            if ((rc = stage_cluster(b, to_extend))) return rc;
        }
        b->done |= BWT_STAGE_CLUSTER;
    }
    if (stages & BWT_STAGE_EXTEND) {
        if ((rc = stage_extend(b, b->bases.as<uint8_t>(), b->ix->d.ref1, b->ix->d.ref1_len, b->hctr.max_w, b->hctr.max_lq,
                               b->ix->shift, b->rplanes.as<uint32_t>(), b->ix->d.rplanes)))
            return rc;
        b->done |= BWT_STAGE_EXTEND;
    }
    if (!(stages & BWT_STAGE_EXTEND) && (rc = fetch_counters(b))) return rc;     // (the extend stage ends with the counters on the host)
    if (b->opt.debug > 0 && b->hctr.n_gq)
        fprintf(stderr, "[libbwtgpu] k_seed_fast handed back %u reads: non-ACGT %u, table overflow %u, more seeds than its row holds %u, >%d clusters %u\n",
                b->hctr.n_gq, b->hctr.why[0], b->hctr.why[1], b->hctr.why[4], CL_STAGE, b->hctr.why[5]);
    if (b->opt.debug > 0)
        fprintf(stderr, "[libbwtgpu] cluster kernels: %u reads of 1..64 seeds, %u of 65..128, %u of 129..256, %u of more (%u of them past k_cluster_big's LDS); certificate kernels: %u settled, %u "
                        "passed to their wider instance, %u to the 12-diagonal tier\n", b->n_small, b->hctr.n_mid2, b->hctr.n_mid4, b->hctr.n_big, b->hctr.n_huge,
                b->hctr.cert_done, b->hctr.cert2_in, b->hctr.cert_left);
    if (counts) {
        counts->n_reads = b->n; counts->n_tasks = b->hctr.n_tasks; counts->lf_steps = b->hctr.lf_steps;
        counts->sa_reads = b->hctr.sa_reads; counts->n_clusters = b->n_cand; counts->sw_cells = b->hctr.sw_cells;
        counts->n_hits = b->n_hits + b->n_narrow; counts->cigar_bytes = b->cig_bytes; counts->window_bytes = b->hctr.window_bytes;
        counts->n_float = b->hctr.fp_count; counts->n_traced = (uint64_t)b->hctr.tq_count + b->hctr.tq2_count;
        // (a short DP queue taken by k_sw_wave is traced as a whole; in score-first mode it never reaches tq2)
        if (b->wave_upto && b->hctr.dq_count <= b->wave_upto && !b->hctr.tq2_count) counts->n_traced += b->hctr.dq_count;
        // (with the 12-diagonal tier: band_in = the 20-diagonal tier's own queue + what the 12-diagonal tier passed on to it)
        const uint64_t n_dp_band = (uint64_t)b->hctr.band_in + b->hctr.band0_in - b->hctr.band0_left;
        counts->n_dp = b->band_ran ? n_dp_band : b->hctr.dq_count;
        counts->n_banded = b->band_ran ? n_dp_band - b->hctr.band3_left : 0;
        counts->n_traced += counts->n_banded;
        counts->n_lookups = b->hctr.kt_lookups; counts->n_small = b->n_small;
        counts->n_handed_back = b->hctr.n_gq;
        counts->n_certified = b->hctr.cert_done;
    }
    return 0;
}

template <typename T> static int d2h(bwt_batch *b, T *dst, const DevBuf &src, size_t count) {
    if (!count || !dst) return 0;
    HIPCHK(hipMemcpyAsync(dst, src.p, count * sizeof(T), hipMemcpyDeviceToHost, b->stream));
    return 0;
}

extern "C" int bwt_batch_fetch_tasks(bwt_batch *b, int32_t *status, uint32_t *sp, uint32_t *ep, uint32_t *nmatch,
                                     uint32_t *task_read, uint32_t *task_end, uint64_t cap, uint64_t *needed) {
    if (!b || !(b->done & BWT_STAGE_SEARCH)) return fail(BWT_EINVAL, "bwt_batch_fetch_tasks: search stage not run");
    if (!b->rows_kept) return fail(BWT_EINVAL, "bwt_batch_fetch_tasks: the task table was not kept (the run included the extend stage); run BWT_STAGE_SEARCH alone first");
    HIPCHK(hipSetDevice(b->device));
    const uint64_t T = (uint64_t)b->n * b->maxT;
    std::vector<uint32_t> def(T), info(T), vsp(T), vep(T);
    int rc;
    if ((rc = d2h(b, def.data(), b->t_def, T)) || (rc = d2h(b, info.data(), b->t_info, T)) ||
        (rc = d2h(b, vsp.data(), b->t_sp, T)) || (rc = d2h(b, vep.data(), b->t_ep, T)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    uint64_t k = 0;
    for (uint64_t t = 0; t < T; t++) {
        if ((def[t] & 0xffff) == 0) continue;
        if (k < cap) {
            const uint32_t st = info[t] & 0xff;
            if (status) status[k] = st == TS_CRASH ? -1 : (int32_t)st;
            if (sp) sp[k] = vsp[t];
            if (ep) ep[k] = vep[t];
            if (nmatch) nmatch[k] = (info[t] >> 8) & 0xff;
            if (task_read) task_read[k] = (uint32_t)(t / b->maxT);
            if (task_end) task_end[k] = def[t] & 0xffff;
        }
        k++;
    }
    if (needed) *needed = k;
    return k > cap ? fail(BWT_ENOSPC, "task capacity too small") : 0;
}

extern "C" int bwt_batch_fetch_clusters(bwt_batch *b, uint64_t *clus_off, int64_t *S, int32_t *D, uint64_t cap,
                                        uint64_t *needed, int32_t *read_status) {
    if (!b || !(b->done & BWT_STAGE_CLUSTER)) return fail(BWT_EINVAL, "bwt_batch_fetch_clusters: cluster stage not run");
    HIPCHK(hipSetDevice(b->device));
    if (needed) *needed = b->n_cand;
    int rc;
    if (clus_off && (rc = d2h(b, clus_off, b->clus_off, (size_t)b->n + 1))) return rc;
    std::vector<uint32_t> qty(b->n);
    if (read_status && (rc = d2h(b, qty.data(), b->r_qty, b->n))) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (read_status) for (uint32_t i = 0; i < b->n; i++) read_status[i] = (qty[i] >> 31) ? -1 : 0;
    if (b->n_cand > cap) return fail(BWT_ENOSPC, "cluster capacity too small");
    std::vector<uint32_t> s(b->n_cand), d(b->n_cand);
    if ((rc = d2h(b, s.data(), b->c_S, b->n_cand)) || (rc = d2h(b, d.data(), b->c_D, b->n_cand))) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint64_t i = 0; i < b->n_cand; i++) { if (S) S[i] = s[i]; if (D) D[i] = (int32_t)d[i]; }
    return 0;
}

// read_status: one signed byte per read on the device, int32 in the per-hit ABI
static int fetch_status32(bwt_batch *b, int32_t *read_status) {
    if (!read_status || !b->n) return 0;
    std::vector<int8_t> st(b->n);
    HIPCHK(hipMemcpyAsync(st.data(), b->r_status.p, b->n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint32_t i = 0; i < b->n; i++) read_status[i] = st[i];
    return 0;
}

extern "C" int bwt_batch_fetch_hits(bwt_batch *b, int32_t *read_status, uint64_t *hit_off, int32_t *score, int64_t *pos,
                                    int64_t *win_off, int32_t *win_len, uint64_t *cig_off, char *cigar, uint64_t hit_cap,
                                    uint64_t cig_cap, uint64_t needed[2]) {
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_fetch_hits: extend stage not run");
    HIPCHK(hipSetDevice(b->device));
    int rc;
    if ((rc = ensure_v1(b))) return rc;
    if (needed) { needed[0] = b->n_hits; needed[1] = b->cig_bytes; }
    if ((rc = fetch_status32(b, read_status)) || (rc = d2h(b, hit_off, b->hit_off, (size_t)b->n + 1))) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->n_hits > hit_cap || b->cig_bytes > cig_cap) return fail(BWT_ENOSPC, "hit / cigar capacity too small");
    if ((rc = ensure_expanded(b))) return rc;
    std::vector<HitRec> rec(b->n_hits);
    if ((rc = d2h(b, rec.data(), b->h_rec, b->n_hits)) || (rc = d2h(b, win_off, b->h_woff, b->n_hits)) ||
        (rc = d2h(b, win_len, b->h_wlen, b->n_hits)) || (rc = d2h(b, cig_off, b->h_cigoff, b->n_hits + 1)) ||
        (rc = d2h(b, cigar, b->h_cig, b->cig_bytes)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint64_t i = 0; i < b->n_hits; i++) { if (score) score[i] = rec[i].score; if (pos) pos[i] = rec[i].pos; }
    return 0;
}

extern "C" int bwt_batch_fetch_hits_raw(bwt_batch *b, int32_t *read_status, uint64_t *hit_off, bwt_hit_rec *hits, int64_t *win_off,
                                        int32_t *win_len, uint64_t *cig_off, char *cigar, uint64_t hit_cap, uint64_t cig_cap,
                                        uint64_t needed[2]) {
    static_assert(sizeof(bwt_hit_rec) == sizeof(HitRec), "bwt_hit_rec is the device record");
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_fetch_hits_raw: extend stage not run");
    HIPCHK(hipSetDevice(b->device));
    { int rc0 = ensure_v1(b); if (rc0) return rc0; }
    if (needed) { needed[0] = b->n_hits; needed[1] = b->cig_bytes; }
    if ((hits || win_off || win_len || cig_off) && b->n_hits > hit_cap) return fail(BWT_ENOSPC, "hit capacity too small");
    if (cigar && b->cig_bytes > cig_cap) return fail(BWT_ENOSPC, "cigar capacity too small");
    if (!read_status && !hit_off && !hits && !win_off && !win_len && !cig_off && !cigar) return b->n_hits || b->cig_bytes ? BWT_ENOSPC : 0;
    int rc;
    if ((win_off || win_len || cig_off || cigar) && (rc = ensure_expanded(b))) return rc;
    if ((rc = fetch_status32(b, read_status)) || (rc = d2h(b, hit_off, b->hit_off, (size_t)b->n + 1)) ||
        (rc = d2h(b, (HitRec *)hits, b->h_rec, b->n_hits)) || (rc = d2h(b, win_off, b->h_woff, b->n_hits)) ||
        (rc = d2h(b, win_len, b->h_wlen, b->n_hits)) || (rc = d2h(b, cig_off, b->h_cigoff, b->n_hits + 1)) ||
        (rc = d2h(b, cigar, b->h_cig, b->cig_bytes)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int bwt_batch_fetch_hits_compact(bwt_batch *b, int8_t *read_status, bwt_hit_rec *hits, bwt_hit_extra *extra, char *cigar,
                                            uint64_t hit_cap, uint64_t extra_cap, uint64_t cig_cap, uint64_t needed[3]) {
    static_assert(sizeof(bwt_hit_extra) == sizeof(HitExtra), "bwt_hit_extra is the device record");
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_fetch_hits_compact: extend stage not run");
    HIPCHK(hipSetDevice(b->device));
    { int rc0 = ensure_v1(b); if (rc0) return rc0; }
    if (needed) { needed[0] = b->n_hits; needed[1] = b->n_extra; needed[2] = b->xcig_bytes; }
    if ((hits && b->n_hits > hit_cap) || (extra && b->n_extra > extra_cap) || (cigar && b->xcig_bytes > cig_cap))
        return fail(BWT_ENOSPC, "hit / extra / cigar capacity too small");
    if (!read_status && !hits && !extra && !cigar) return (b->n_hits || b->n_extra || b->xcig_bytes) ? BWT_ENOSPC : 0;
    int rc;
    if ((rc = d2h(b, read_status, b->r_status, b->n)) || (rc = d2h(b, (HitRec *)hits, b->h_rec, b->n_hits)) ||
        (rc = d2h(b, (HitExtra *)extra, b->x_rec, b->n_extra)) || (rc = d2h(b, cigar, b->x_cig, b->xcig_bytes)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

// the table as the device made it under bwt_options.hits_narrow (a batch run without the option, or one whose reads are too long
// for 16-bit scores, simply has no narrow reads: everything is in the record stream)
extern "C" int bwt_batch_fetch_hits_narrow(bwt_batch *b, int8_t *read_status, uint32_t *narrow_off, int16_t *narrow_score, bwt_hit_rec *hits,
                                           bwt_hit_extra *extra, char *cigar, uint64_t narrow_cap, uint64_t hit_cap, uint64_t extra_cap, uint64_t cig_cap,
                                           uint64_t needed[4]) {
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_fetch_hits_narrow: extend stage not run");
    HIPCHK(hipSetDevice(b->device));
    if (needed) { needed[0] = b->n_narrow; needed[1] = b->n_hits; needed[2] = b->n_extra; needed[3] = b->xcig_bytes; }
    if (((narrow_off || narrow_score) && b->n_narrow > narrow_cap) || (hits && b->n_hits > hit_cap) || (extra && b->n_extra > extra_cap) ||
        (cigar && b->xcig_bytes > cig_cap))
        return fail(BWT_ENOSPC, "narrow / hit / extra / cigar capacity too small");
    if (!read_status && !narrow_off && !narrow_score && !hits && !extra && !cigar) return (b->n_narrow || b->n_hits || b->n_extra || b->xcig_bytes) ? BWT_ENOSPC : 0;
    int rc;
    if ((rc = d2h(b, read_status, b->r_status, b->n)) || (rc = d2h(b, narrow_off, b->n_woff, b->n_narrow)) || (rc = d2h(b, narrow_score, b->n_score, b->n_narrow)) ||
        (rc = d2h(b, (HitRec *)hits, b->h_rec, b->n_hits)) || (rc = d2h(b, (HitExtra *)extra, b->x_rec, b->n_extra)) || (rc = d2h(b, cigar, b->x_cig, b->xcig_bytes)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int bwt_batch_hits_to_device(bwt_batch *b, void *dev_dst, uint64_t cap_hits) {
    if (!b || !(b->done & BWT_STAGE_EXTEND) || !dev_dst) return fail(BWT_EINVAL, "bwt_batch_hits_to_device: not ready");
    HIPCHK(hipSetDevice(b->device));
    { int rc0 = ensure_v1(b); if (rc0) return rc0; }
    if (b->n_hits > cap_hits) return fail(BWT_ENOSPC, "hit capacity too small");
    if (b->n_hits) HIPCHK(hipMemcpyAsync(dev_dst, b->h_rec.p, b->n_hits * 16, hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

__global__ __launch_bounds__(256) static void k_hits_global(const HitRec *__restrict__ in, uint64_t n, uint32_t base, HitRec *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        HitRec h = in[i];
        h.read += base;
        out[i] = h;
    }
}
extern "C" int bwt_batch_hits_to_device_global(bwt_batch *b, void *dev_dst, uint64_t cap_hits, uint32_t read_base) {
    if (!b || !(b->done & BWT_STAGE_EXTEND) || !dev_dst) return fail(BWT_EINVAL, "bwt_batch_hits_to_device_global: not ready");
    HIPCHK(hipSetDevice(b->device));
    { int rc0 = ensure_v1(b); if (rc0) return rc0; }
    if (b->n_hits > cap_hits) return fail(BWT_ENOSPC, "hit capacity too small");
    if (b->n_hits)
        hipLaunchKernelGGL(k_hits_global, dim3((uint32_t)std::min<uint64_t>(cdiv(b->n_hits, 256), 4096)), dim3(256), 0, b->stream,
                           b->h_rec.as<HitRec>(), b->n_hits, read_base, (HitRec *)dev_dst);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// the whole compact table, device to device, into caller-owned device memory (the send buffer of the RCCL gather): hit records
// with global read ids, extras and their text as they are (their hit / text offsets stay slab-local: the receiver adds the bases)
extern "C" int bwt_batch_compact_to_device(bwt_batch *b, void *dst_hits, uint64_t cap_hits, void *dst_extra, uint64_t cap_extra,
                                           void *dst_cigar, uint64_t cap_cigar, uint32_t read_base) {
    if (!b || !(b->done & BWT_STAGE_EXTEND)) return fail(BWT_EINVAL, "bwt_batch_compact_to_device: extend stage not run");
    HIPCHK(hipSetDevice(b->device));
    { int rc0 = ensure_v1(b); if (rc0) return rc0; }
    if ((!dst_hits && b->n_hits) || (!dst_extra && b->n_extra) || (!dst_cigar && b->xcig_bytes)) return fail(BWT_EINVAL, "bwt_batch_compact_to_device: null destination");
    if (b->n_hits > cap_hits || b->n_extra > cap_extra || b->xcig_bytes > cap_cigar) return fail(BWT_ENOSPC, "bwt_batch_compact_to_device: capacity too small");
    if (b->n_hits)
        hipLaunchKernelGGL(k_hits_global, dim3((uint32_t)std::min<uint64_t>(cdiv(b->n_hits, 256), 4096)), dim3(256), 0, b->stream,
                           b->h_rec.as<HitRec>(), b->n_hits, read_base, (HitRec *)dst_hits);
    if (b->n_extra) HIPCHK(hipMemcpyAsync(dst_extra, b->x_rec.p, b->n_extra * 16, hipMemcpyDeviceToDevice, b->stream));
    if (b->xcig_bytes) HIPCHK(hipMemcpyAsync(dst_cigar, b->x_cig.p, b->xcig_bytes, hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fastq:reverse_complement/1 (src/fastq.erl:43-50): wave per read, coalesced reads from the right end
__global__ __launch_bounds__(256) static void k_revcomp(const uint8_t *__restrict__ in, const uint64_t *__restrict__ off, uint32_t n,
                                                         uint8_t *__restrict__ out, int32_t *__restrict__ status) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        const uint64_t a = off[r], len = off[r + 1] - a;
        bool bad = false;
        for (uint64_t i = lane; i < len; i += 64) {
            const uint8_t c = in[a + len - 1 - i];
            const uint8_t d = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'N' ? 'N' : 0;
            bad = bad || d == 0;
            out[a + i] = d;
        }
        const bool any = __ballot(bad) != 0;
        if (lane == 0) status[r] = any ? -1 : 0;
    }
}

extern "C" int bwt_reverse_complement_batch(int device, const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *out, int32_t *status) {
    if (!off || (!bases && n) || (!out && n) || (!status && n)) return fail(BWT_EINVAL, "bwt_reverse_complement_batch: null argument");
    for (uint32_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return fail(BWT_EINVAL, "bwt_reverse_complement_batch: offsets must be non-decreasing");
    if (!n) return 0;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "bwt_reverse_complement_batch: no such HIP device");
    HIPCHK(hipSetDevice(device));
    const uint64_t lo = off[0], total = off[n] - lo;
    DevBuf din, dout, doff, dst;
    struct G { DevBuf *b[4]; ~G() { for (auto x : b) x->release(); } } g{{&din, &dout, &doff, &dst}};
    int rc;
    if ((rc = din.ensure(total + 16)) || (rc = dout.ensure(total + 16)) || (rc = doff.ensure(((size_t)n + 1) * 8)) || (rc = dst.ensure((size_t)n * 4)))
        return rc;
    std::vector<uint64_t> o((size_t)n + 1);
    for (uint32_t i = 0; i <= n; i++) o[i] = off[i] - lo;
    HIPCHK(hipMemcpy(din.p, bases + lo, total, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doff.p, o.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_revcomp, dim3((uint32_t)std::min<uint64_t>(cdiv((uint64_t)n * 64, 256), 4096)), dim3(256), 0, 0, din.as<uint8_t>(),
                       doff.as<uint64_t>(), n, dout.as<uint8_t>(), dst.as<int32_t>());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out + lo, dout.p, total, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host-buffer entry points (what a NIF binds)
extern "C" int bwt_find_seeds_batch(bwt_index *ix, const uint8_t *bases, const uint64_t *off, uint32_t n, int32_t *status,
                                    uint32_t *sp, uint32_t *ep, uint32_t *nmatch, uint64_t *seed_off, int64_t *seeds,
                                    uint64_t seeds_cap, uint64_t *seeds_needed) {
    if (!ix || !off) return fail(BWT_EINVAL, "bwt_find_seeds_batch: null argument");
    for (uint32_t i = 0; i < n; i++)
        if (off[i + 1] <= off[i]) return fail(BWT_EINVAL, "bwt_find_seeds_batch: empty sub-read (the reference crashes on [])");
    bwt_batch *b = nullptr;
    int rc = bwt_batch_create(ix, &b);
    if (rc) return rc;
    struct Guard { bwt_batch *b; ~Guard() { bwt_batch_destroy(b); } } g{b};
    if ((rc = bwt_batch_set_reads(b, bases, off, n))) return rc;
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    // every sub-read is its own task here: one task slot per "read", defined directly
    b->maxT = 1;
    const uint64_t T = n;
    if ((rc = b->t_def.ensure(T * 4 + 4)) || (rc = b->t_sp.ensure(T * 4 + 4)) || (rc = b->t_ep.ensure(T * 4 + 4)) ||
        (rc = b->t_info.ensure(T * 4 + 4)) || (rc = b->t_cnt.ensure(T * 4 + 4)) || (rc = b->seed_off.ensure((T + 1) * 8)))
        return rc;
    std::vector<uint32_t> def(T);
    for (uint32_t i = 0; i < n; i++) def[i] = (uint32_t)(off[i + 1] - off[i]);
    HIPCHK(hipMemsetAsync(b->ctr.p, 0, sizeof(Counters), b->stream));
    if (T) {
        HIPCHK(hipMemcpyAsync(b->t_def.p, def.data(), T * 4, hipMemcpyHostToDevice, b->stream));
        if ((rc = read_planes(b))) return rc;
        if ((rc = launch_search(b, n, 1u, true))) return rc;
    }
    if ((rc = dev_scan(b, b->t_cnt.as<uint32_t>(), b->seed_off.as<uint64_t>(), T))) return rc;
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, b->seed_off.as<uint64_t>() + T, 8, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (seeds_needed) *seeds_needed = total;
    std::vector<uint32_t> info(T);
    if ((rc = d2h(b, info.data(), b->t_info, T)) || (rc = d2h(b, sp, b->t_sp, T)) || (rc = d2h(b, ep, b->t_ep, T)) ||
        (rc = d2h(b, seed_off, b->seed_off, T + 1)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint64_t i = 0; i < T; i++) {
        const uint32_t st = info[i] & 0xff;
        if (status) status[i] = st == TS_CRASH ? -1 : (int32_t)st;
        if (nmatch) nmatch[i] = (info[i] >> 8) & 0xff;
    }
    if (total > seeds_cap) return fail(BWT_ENOSPC, "seed capacity too small");
    if (total && seeds) {
        DevBuf dseeds;
        if ((rc = dseeds.ensure(total * 8))) return rc;
        KLAUNCH(b, "seed_list", k_seed_list, cdiv(T * 64, 256), 256, ix->d, T, b->t_sp.as<uint32_t>(), b->t_info.as<uint32_t>(),
                b->t_cnt.as<uint32_t>(), b->seed_off.as<uint64_t>(), dseeds.as<int64_t>());
        hipError_t e = hipMemcpyAsync(seeds, dseeds.p, total * 8, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        dseeds.release();
        if (e != hipSuccess) return fail(BWT_EHIP, std::string("seed list: ") + hipGetErrorString(e));
    }
    return 0;
}

extern "C" int bwt_sga_batch(bwt_index *ix, const uint8_t *bases, const uint64_t *off, uint32_t n, uint64_t *clus_off,
                             int64_t *S, int32_t *D, uint64_t cap, uint64_t *needed, int32_t *read_status) {
    if (!ix || !off) return fail(BWT_EINVAL, "bwt_sga_batch: null argument");
    bwt_batch *b = nullptr;
    int rc = bwt_batch_create(ix, &b);
    if (rc) return rc;
    struct Guard { bwt_batch *b; ~Guard() { bwt_batch_destroy(b); } } g{b};
    if ((rc = bwt_batch_set_reads(b, bases, off, n))) return rc;
    if ((rc = bwt_batch_run(b, BWT_STAGE_CLUSTER, nullptr))) return rc;
    return bwt_batch_fetch_clusters(b, clus_off, S, D, cap, needed, read_status);
}

extern "C" int bwt_align_batch(bwt_index *ix, const uint8_t *bases, const uint64_t *off, uint32_t n, int32_t *read_status,
                               uint64_t *hit_off, int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len,
                               uint64_t *cig_off, char *cigar, uint64_t hit_cap, uint64_t cig_cap, uint64_t needed[2]) {
    if (!ix || !off) return fail(BWT_EINVAL, "bwt_align_batch: null argument");
    bwt_batch *b = nullptr;
    int rc = bwt_batch_create(ix, &b);
    if (rc) return rc;
    struct Guard { bwt_batch *b; ~Guard() { bwt_batch_destroy(b); } } g{b};
    if ((rc = bwt_batch_set_reads(b, bases, off, n))) return rc;
    if ((rc = bwt_batch_run(b, BWT_STAGE_EXTEND, nullptr))) return rc;
    return bwt_batch_fetch_hits(b, read_status, hit_off, score, pos, win_off, win_len, cig_off, cigar, hit_cap, cig_cap,
                                needed);
}

extern "C" int bwt_sw_batch(int device, const uint8_t *q, const uint64_t *qoff, const uint8_t *r, const uint64_t *roff,
                            uint32_t n, int32_t *status, int32_t *score, uint64_t *cig_off, char *cigar, uint64_t cig_cap,
                            uint64_t *cig_needed) {
    if ((!q && n) || !qoff || (!r && n) || !roff) return fail(BWT_EINVAL, "bwt_sw_batch: null argument");
    uint32_t max_lq = 0, max_w = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (qoff[i + 1] <= qoff[i] || roff[i + 1] <= roff[i])
            return fail(BWT_EINVAL, "bwt_sw_batch: empty sequence (sw:sw/2 needs two non-empty lists)");
        if (qoff[i + 1] - qoff[i] > 65535 || roff[i + 1] - roff[i] > (1u << 24))
            return fail(BWT_EINVAL, "bwt_sw_batch: sequence too long");
        max_lq = std::max<uint32_t>(max_lq, (uint32_t)(qoff[i + 1] - qoff[i]));
        max_w = std::max<uint32_t>(max_w, (uint32_t)(roff[i + 1] - roff[i]));
    }
    bwt_batch *b = nullptr;
    int rc = batch_create_noindex(device, &b);
    if (rc) return rc;
    struct Guard { bwt_batch *b; ~Guard() { bwt_batch_destroy(b); } } g{b};
    const uint64_t qtot = n ? qoff[n] - qoff[0] : 0, rtot = n ? roff[n] - roff[0] : 0;
    DevBuf dq, dr;
    struct G2 { DevBuf &a, &b; ~G2() { a.release(); b.release(); } } g2{dq, dr};
    const size_t nc = (size_t)n + 1;
    if ((rc = dq.ensure(qtot + 16)) || (rc = dr.ensure(rtot + 16)) || (rc = b->c_qoff.ensure(nc * 8)) ||
        (rc = b->c_qlen.ensure(nc * 4)) || (rc = b->c_woff.ensure(nc * 8)) || (rc = b->c_wlen.ensure(nc * 4)) ||
        (rc = b->clus_off.ensure((nc + 1) * 8)) || (rc = b->r_qty.ensure(nc * 4)) || (rc = b->c_S.ensure(nc * 4)))
        return rc;
    std::vector<uint64_t> hq(n), co((size_t)n + 1);
    std::vector<int64_t> hw(n);
    std::vector<uint32_t> lq(n), lw(n);
    for (uint32_t i = 0; i < n; i++) {
        hq[i] = qoff[i] - qoff[0]; hw[i] = (int64_t)(roff[i] - roff[0]);
        lq[i] = (uint32_t)(qoff[i + 1] - qoff[i]); lw[i] = (uint32_t)(roff[i + 1] - roff[i]);
        co[i] = i;
    }
    co[n] = n;
    HIPCHK(hipMemsetAsync(b->ctr.p, 0, sizeof(Counters), b->stream));
    if (n) {
        HIPCHK(hipMemcpyAsync(dq.p, q + qoff[0], qtot, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(dr.p, r + roff[0], rtot, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->c_qoff.p, hq.data(), (size_t)n * 8, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->c_woff.p, hw.data(), (size_t)n * 8, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->c_qlen.p, lq.data(), (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->c_wlen.p, lw.data(), (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemsetAsync(b->r_qty.p, 0, (size_t)n * 4, b->stream));
        HIPCHK(hipMemsetAsync(b->c_S.p, 0, (size_t)n * 4, b->stream));
    }
    HIPCHK(hipMemcpyAsync(b->clus_off.p, co.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, b->stream));
    // one "read" per pair, one candidate each; the queries' offsets are the batch's read offsets (the hit expansion wants Lq)
    std::vector<uint64_t> hq1((size_t)n + 1);
    for (uint32_t i = 0; i <= n; i++) hq1[i] = qoff[i] - qoff[0];
    if ((rc = b->off.ensure(((size_t)n + 1) * 8))) return rc;
    HIPCHK(hipMemcpyAsync(b->off.p, hq1.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->n = n; b->n_cand = n; b->ulen = 0;
    // bit-planes of both byte streams so that the prefilter applies here as well
    DevBuf pq, pr;
    struct G3 { DevBuf &a, &b; ~G3() { a.release(); b.release(); } } g3{pq, pr};
    if ((rc = pq.ensure(rp_words(qtot) * 4)) || (rc = pr.ensure(rp_words(rtot) * 4))) return rc;
    {
        ProfScope ps(b, "read_planes");
        if ((rc = launch_planes(dq.as<uint8_t>(), qtot, pq.as<uint32_t>(), b->device, b->stream)) ||
            (rc = launch_planes(dr.as<uint8_t>(), rtot, pr.as<uint32_t>(), b->device, b->stream)))
            return rc;
    }
    if ((rc = stage_extend(b, dq.as<uint8_t>(), dr.as<uint8_t>(), rtot, max_w, max_lq, 0, pq.as<uint32_t>(), pr.as<uint32_t>())))
        return rc;
    if (cig_needed) *cig_needed = b->cig_bytes;
    std::vector<int32_t> cst(n), csc(n);
    std::vector<uint32_t> cl(n);
    if ((rc = d2h(b, cst.data(), b->c_status, n)) || (rc = d2h(b, csc.data(), b->c_score, n)) ||
        (rc = d2h(b, cl.data(), b->c_ciglen, n)))
        return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    uint64_t o = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (cig_off) cig_off[i] = o;
        const bool ok = cst[i] == CS_OK;
        if (status) status[i] = ok ? 0 : 1;
        if (score) score[i] = ok ? csc[i] : 0;
        if (ok) o += cl[i] & ~CIG_DIAG;
    }
    if (cig_off) cig_off[n] = o;
    if (o > cig_cap) return fail(BWT_ENOSPC, "cigar capacity too small");
    if (o && cigar) {
        if ((rc = ensure_expanded(b))) return rc;
        if (o != b->cig_bytes) return fail(BWT_EHIP, "internal: CIGAR text size mismatch");
        HIPCHK(hipMemcpyAsync(cigar, b->h_cig.p, o, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    return 0;
}

#include "small_batch.hip.h"
