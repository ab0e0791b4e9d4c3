// libbwtgpu: seeds.erl on the GPU -- seeds:generate_fs/3 (src/seeds.erl:11-85) and the module `fs`
// it writes (matches/Len clauses + the sliding find_seeds/1).
//
// The generated clauses are every way of spending EXACTLY `Defects` edits (mutation, insertion, deletion; edited
// window positions become `_`) on the first Len+Defects query bases while producing Len pattern items
// (gen_pattern/5, :51-62).  A window matches some clause iff a small NFA accepts it:
//     state (i window items consumed, j query items consumed, d edits spent), start (0,0,0);
//     match     w[i] == q[j]        -> (i+1, j+1, d)
//     mutation  any w[i]            -> (i+1, j+1, d+1)        `_`, the query base is consumed      (:58)
//     insertion any w[i]            -> (i+1, j,   d+1)        `_`, the query base stays            (:59)
//     deletion  (only while i<Len)  -> (i,   j+1, d+1)        the query base is skipped            (:60, guard :57)
//     accept    i == Len and d == Defects   (after the last edit only matches follow: the clause of :51-55)
// j - i stays within [-Defects, +Defects], so the state set is (Defects+1) x (2 Defects+1) bits: one u64 per lane,
// level d in bits [10d, 10d+9).  tests/test_fs_seeds.py checks the NFA against the explicit clause list of the oracle.
//
// fs:find_seeds/1 (:24-41) is a greedy left-to-right scan: a hit at N is recorded and the scan continues at N+Len, a
// miss continues at N+1.  Here: k_fs_match marks every matching window start (lane per start, reference tile in LDS,
// waves stop once no lane has a live state), the marks are compacted to an ascending list, and one wave replays the
// greedy scan over that list (`first mark >= last hit + Len`), 64 marks per step; the result is written newest first.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define FS_MAX_DEFECTS 4
#define FS_MAX_LEN 1024
#define FS_TILE 1024                       // window starts per block pass (256 lanes x 4)

__device__ __forceinline__ unsigned long long fs_rep(int D) {      // bit 0 of every level 0..D
    unsigned long long r = 0;
    for (int d = 0; d <= D; d++) r |= 1ull << (10 * d);
    return r;
}

// sub: the query prefix as u16, padded with D never-matching items (0xffff) in front and D+1 behind:
// sub[D + j] = q[j] for 0 <= j < Len+D.  marks: one bit per window start (u64 words, start p at bit p & 63 of word p >> 6).
__global__ __launch_bounds__(256) void k_fs_match(const uint8_t *__restrict__ ref, uint64_t n, const uint16_t *__restrict__ sub,
                                                   int Len, int D, unsigned long long *__restrict__ marks, uint64_t nwords) {
    extern __shared__ uint8_t tile[];                       // FS_TILE + Len bytes
    const uint64_t nstarts = n >= (uint64_t)Len ? n - (uint64_t)Len + 1 : 0;
    const unsigned long long REP = fs_rep(D), VALID = REP * 0x1ffull;       // 9 bits per level (D <= 4 uses 2D+1 <= 9)
    const unsigned long long START = 1ull << D;             // level 0, j - i = 0
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t base = (uint64_t)blockIdx.x * FS_TILE; base < nwords * 64; base += (uint64_t)gridDim.x * FS_TILE) {
        __syncthreads();
        const uint32_t need = FS_TILE + (uint32_t)Len - 1;
        for (uint32_t i = threadIdx.x; i < need; i += 256) tile[i] = base + i < n ? ref[base + i] : (uint8_t)0;
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < FS_TILE / 256; k++) {
            const uint32_t t = (uint32_t)k * 256 + wv * 64 + lane;          // my start within the tile
            const uint64_t p = base + t;
            unsigned long long S = p < nstarts ? START : 0ull;
            // deletions before the first window item (i = 0 < Len)
            for (int d = 0; d < D; d++) S |= (((S << 1) & VALID) << 10) & VALID;
            for (int i = 0; i < Len; i++) {
                if (!__any(S != 0)) break;                                  // wave-uniform: nobody is alive
                const uint32_t w = tile[t + (uint32_t)i];
                uint32_t E = 0;                                             // bit b: w == q[i + b - D]
                for (int b = 0; b <= 2 * D; b++) E |= (uint32_t)(sub[i + b] == w) << b;   // uniform address: scalar loads
                S = (S & ((unsigned long long)E * REP)) | (((S | ((S >> 1) & VALID)) << 10) & VALID);
                if (i + 1 < Len)
                    for (int d = 0; d < D; d++) S |= (((S << 1) & VALID) << 10) & VALID;
            }
            const bool hit = ((S >> (10 * D)) & 0x1ffull) != 0;
            const unsigned long long m = __ballot(hit);
            if (lane == 0 && (p >> 6) < nwords) marks[p >> 6] = m;
        }
    }
}

// ascending list of the marked starts: per-block counts, a one-block scan of the block sums, then the list
#define FS_LWORDS 256                       // mark words per list block
__global__ __launch_bounds__(256) void k_fs_count(const unsigned long long *__restrict__ marks, uint64_t nwords,
                                                   unsigned long long *__restrict__ bsum) {
    __shared__ uint32_t red[4];
    const uint64_t w = (uint64_t)blockIdx.x * FS_LWORDS + threadIdx.x;
    uint32_t c = w < nwords ? (uint32_t)__popcll(marks[w]) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (unsigned long long)red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_fs_scan(unsigned long long *bsum, uint64_t nb, unsigned long long *total) {
    __shared__ unsigned long long part[256], carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t c0 = 0; c0 < nb; c0 += 256 * 16) {
        unsigned long long v[16], s = 0;
        for (int k = 0; k < 16; k++) { const uint64_t i = c0 + (uint64_t)threadIdx.x * 16 + k; v[k] = i < nb ? bsum[i] : 0ull; s += v[k]; }
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long run = carry; for (int t = 0; t < 256; t++) { const unsigned long long x = part[t]; part[t] = run; run += x; } carry = run; }
        __syncthreads();
        unsigned long long run = part[threadIdx.x];
        for (int k = 0; k < 16; k++) { const uint64_t i = c0 + (uint64_t)threadIdx.x * 16 + k; if (i < nb) bsum[i] = run; run += v[k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void k_fs_list(const unsigned long long *__restrict__ marks, uint64_t nwords,
                                                  const unsigned long long *__restrict__ bsum, uint32_t *__restrict__ list) {
    __shared__ uint32_t wsum[4];
    const uint64_t w = (uint64_t)blockIdx.x * FS_LWORDS + threadIdx.x;
    unsigned long long m = w < nwords ? marks[w] : 0ull;
    const uint32_t c = (uint32_t)__popcll(m), lane = threadIdx.x & 63;
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) before += wsum[k];
    unsigned long long o = bsum[blockIdx.x] + before + incl - c;
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        list[o++] = (uint32_t)(w * 64 + (uint64_t)b);
    }
}

// one wave replays fs:find_seeds/3 over the ascending mark list: the next hit is the first mark >= thresh
__global__ __launch_bounds__(64) void k_fs_select(const uint32_t *__restrict__ list, uint64_t m, uint32_t Len,
                                                   uint32_t *__restrict__ sel, unsigned long long *__restrict__ nsel) {
    const uint32_t lane = threadIdx.x;
    uint64_t thresh = 0, count = 0;
    uint32_t keep = 0;                                       // lane k holds hit number (count & ~63) + k until 64 are full
    for (uint64_t c0 = 0; c0 < m; c0 += 64) {
        const bool valid = c0 + lane < m;
        const uint32_t p = valid ? list[c0 + lane] : 0u;
        for (;;) {
            const unsigned long long mask = __ballot(valid && (uint64_t)p >= thresh);
            if (!mask) break;
            const uint32_t ps = (uint32_t)__shfl((int)p, __ffsll((long long)mask) - 1, 64);
            if (lane == (uint32_t)(count & 63)) keep = ps;
            count++;
            thresh = (uint64_t)ps + Len;
            if ((count & 63) == 0) sel[count - 64 + lane] = keep;
        }
    }
    if (lane < (count & 63)) sel[(count & ~63ull) + lane] = keep;
    if (lane == 0) *nsel = count;
}
// Acc is built by prepending (seeds.erl:29,35): newest first, positions 1-based
__global__ __launch_bounds__(256) void k_fs_out(const uint32_t *__restrict__ sel, uint64_t count, int64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (int64_t)sel[count - 1 - i] + 1;
}

// ------------------------------------------------------------------------------------------------
static int fs_check_args(const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects, const char *who) {
    if (!qsec && qlen) return fail(BWT_EINVAL, std::string(who) + ": null query");
    if (len < 2) return fail(BWT_EINVAL, std::string(who) + ": Len must be >= 2 (the generator crashes below: [_|Res] = \"\" for 0, src/seeds.erl:81-85; "
                                                  "[_,_,_|T] = \"X1\" for 1, :32)");
    if (len > FS_MAX_LEN || defects > FS_MAX_DEFECTS)
        return fail(BWT_EINVAL, std::string(who) + ": supported up to Len " + std::to_string(FS_MAX_LEN) + ", Defects " + std::to_string(FS_MAX_DEFECTS));
    if ((uint64_t)qlen < (uint64_t)len + defects)
        return fail(BWT_EINVAL, std::string(who) + ": the query must hold Len+Defects bases (a shorter one makes gen_pattern print a "
                                                   "clause of smaller arity: the generated fs.erl does not compile)");
    return 0;
}

static int fs_run(int device, const uint8_t *qsec, uint32_t len, uint32_t defects, const uint8_t *dref, uint64_t n,
                  int64_t *pos, uint64_t cap, uint64_t *needed) {
    if (n >= 0xffffff00ull) return fail(BWT_EINVAL, "fs:find_seeds: reference longer than 2^32-257 bases");
    if (needed) *needed = 0;
    if (n < len) return 0;                                   // find_seeds(Acc,_,_) -> Acc at once
    const int D = (int)defects, Len = (int)len;
    std::vector<uint16_t> sub((size_t)Len + 3 * D + 2, 0xffffu);
    for (int j = 0; j < Len + D; j++) sub[(size_t)D + j] = qsec[j];
    const uint64_t nstarts = n - len + 1, nwords = (nstarts + 63) / 64, nb = (nwords + FS_LWORDS - 1) / FS_LWORDS;
    DevBuf dsub, dmarks, dbsum, dtot, dlist, dsel, dout;
    struct G { DevBuf *b[7]; ~G() { for (auto x : b) x->release(); } } g{{&dsub, &dmarks, &dbsum, &dtot, &dlist, &dsel, &dout}};
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct SG { hipStream_t s; ~SG() { (void)hipStreamDestroy(s); } } sg{st};
    int rc;
    if ((rc = dsub.ensure(sub.size() * 2)) || (rc = dmarks.ensure(nwords * 8)) || (rc = dbsum.ensure((nb + 1) * 8)) || (rc = dtot.ensure(16)))
        return rc;
    HIPCHK(hipMemcpyAsync(dsub.p, sub.data(), sub.size() * 2, hipMemcpyHostToDevice, st));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nwords * 64 + FS_TILE - 1) / FS_TILE, (uint64_t)cus * 8));
    hipLaunchKernelGGL(k_fs_match, dim3(grid), dim3(256), (size_t)(FS_TILE + Len), st, dref, n, dsub.as<uint16_t>(), Len, D,
                       dmarks.as<unsigned long long>(), nwords);
    hipLaunchKernelGGL(k_fs_count, dim3((uint32_t)nb), dim3(256), 0, st, dmarks.as<unsigned long long>(), nwords, dbsum.as<unsigned long long>());
    hipLaunchKernelGGL(k_fs_scan, dim3(1), dim3(256), 0, st, dbsum.as<unsigned long long>(), nb, dtot.as<unsigned long long>());
    unsigned long long m = 0;
    HIPCHK(hipMemcpyAsync(&m, dtot.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (!m) return 0;
    if ((rc = dlist.ensure(m * 4)) || (rc = dsel.ensure((m + 64) * 4))) return rc;
    hipLaunchKernelGGL(k_fs_list, dim3((uint32_t)nb), dim3(256), 0, st, dmarks.as<unsigned long long>(), nwords, dbsum.as<unsigned long long>(),
                       dlist.as<uint32_t>());
    hipLaunchKernelGGL(k_fs_select, dim3(1), dim3(64), 0, st, dlist.as<uint32_t>(), (uint64_t)m, len, dsel.as<uint32_t>(),
                       dtot.as<unsigned long long>() + 1);
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, dtot.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (needed) *needed = cnt;
    if (cnt > cap || (!pos && cnt)) return fail(BWT_ENOSPC, "fs:find_seeds: position capacity too small");
    if ((rc = dout.ensure(cnt * 8))) return rc;
    hipLaunchKernelGGL(k_fs_out, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, st, dsel.as<uint32_t>(), (uint64_t)cnt, dout.as<int64_t>());
    HIPCHK(hipMemcpyAsync(pos, dout.p, cnt * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int bwt_fs_find_seeds(int device, const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects,
                                 const uint8_t *ref, uint64_t ref_len, int64_t *pos, uint64_t cap, uint64_t *needed) {
    int rc = fs_check_args(qsec, qlen, len, defects, "bwt_fs_find_seeds");
    if (rc) return rc;
    if (!ref && ref_len) return fail(BWT_EINVAL, "bwt_fs_find_seeds: null reference");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "bwt_fs_find_seeds: no such HIP device");
    HIPCHK(hipSetDevice(device));
    DevBuf dref;
    struct G { DevBuf &a; ~G() { a.release(); } } g{dref};
    if ((rc = dref.ensure(ref_len + 16))) return rc;
    if (ref_len) HIPCHK(hipMemcpy(dref.p, ref, ref_len, hipMemcpyHostToDevice));
    return fs_run(device, qsec, len, defects, dref.as<uint8_t>(), ref_len, pos, cap, needed);
}

extern "C" int bwt_fs_find_seeds_index(const bwt_index *ix, const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects,
                                       int64_t *pos, uint64_t cap, uint64_t *needed) {
    if (!ix) return fail(BWT_EINVAL, "bwt_fs_find_seeds_index: null index");
    int rc = fs_check_args(qsec, qlen, len, defects, "bwt_fs_find_seeds_index");
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    // the chunk's own bases: Ref1 without its two 200-N halos
    return fs_run(ix->device, qsec, len, defects, ix->ref1.as<uint8_t>() + BWT_REF_EXT, ix->ref1_len - 2 * BWT_REF_EXT, pos, cap, needed);
}

// ------------------------------------------------------------------------------------------------
// seeds:generate_fs/3: the text of fs.erl (host only; the same depth-first recursion as src/seeds.erl:51-62)
namespace {
struct FsGen {
    int len;
    std::string out;
    uint64_t bytes = 0;
    bool count_only;
    std::vector<int> acc;                      // items in reading order; -1 = `_`
    void emit(const uint8_t *rest) {           // print_pattern/2 (:65-74)
        const int ntail = len - (int)acc.size();
        uint64_t sz = 8;                                                  // "matches("
        sz += (uint64_t)len - 1;                                          // commas
        for (int v : acc) sz += v < 0 ? 1 : 2;
        sz += 2ull * ntail + 11;                                          // ") -> true;\n"
        bytes += sz;
        if (count_only) return;
        out += "matches(";
        bool first = true;
        auto item = [&](int v) { if (!first) out += ','; first = false; if (v < 0) out += '_'; else { out += '$'; out += (char)v; } };
        for (int v : acc) item(v);
        for (int i = 0; i < ntail; i++) item(rest[i]);
        out += ") -> true;\n";
    }
    void gen(const uint8_t *rest, int nrest, int defects) {
        if (defects == 0) { emit(rest); return; }
        if (nrest > 0 && (int)acc.size() < len) {
            acc.push_back(-1); gen(rest + 1, nrest - 1, defects - 1); acc.pop_back();          // mutation
            acc.push_back(-1); gen(rest, nrest, defects - 1); acc.pop_back();                  // insertion
            gen(rest + 1, nrest - 1, defects - 1);                                             // deletion
            acc.push_back(rest[0]); gen(rest + 1, nrest - 1, defects); acc.pop_back();         // match
        }
    }
};
}  // namespace

extern "C" int bwt_seeds_generate_fs(const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects, char *text,
                                     uint64_t cap, uint64_t *needed) {
    int rc = fs_check_args(qsec, qlen, len, defects, "bwt_seeds_generate_fs");
    if (rc) return rc;
    std::string args, args1;
    for (uint32_t j = 1; j <= len; j++) { if (j > 1) args += ','; args += "X" + std::to_string(j); }
    args1 = args.substr(3);                                              // [_,_,_|Args_str1] = Args_str (:32)
    std::string head = "% Seeds finding algo\n%\n% Cloudozer(c), 2014\n\n\n-module(fs).\n-export([matches/" + std::to_string(len) +
                       ", find_seeds/1]).\n";
    head += "\n\nfind_seeds(Ref_Seq) ->\n\tfind_seeds([],1, Ref_Seq).\n\nfind_seeds(Acc,N,[" + args + "]) ->\n\tcase matches(" + args +
            ") of\n\t\ttrue -> [N|Acc];\n\t\t_ -> Acc\n\tend;\n";
    head += "\nfind_seeds(Acc,N,[" + args + "|Ref_Seq]) ->\n\tcase matches(" + args + ") of\n\t\ttrue -> find_seeds([N|Acc],N+length([" + args +
            "]),Ref_Seq);\n\t\t_ -> find_seeds(Acc,N+1,[" + args1 + "|Ref_Seq])\n\tend;\n\nfind_seeds(Acc,_,_) -> Acc.\n\n\n";
    std::string tail = "matches(";
    for (uint32_t j = 0; j < len; j++) { if (j) tail += ','; tail += '_'; }
    tail += ") -> false.\n";
    FsGen g{(int)len, std::string(), 0, true, {}};
    g.gen(qsec, (int)(len + defects), (int)defects);
    const uint64_t total = head.size() + g.bytes + tail.size();
    if (needed) *needed = total;
    if (total > cap || !text) return fail(BWT_ENOSPC, "bwt_seeds_generate_fs: text capacity too small");
    g.count_only = false; g.bytes = 0;
    g.out.reserve((size_t)(total - head.size()));
    g.gen(qsec, (int)(len + defects), (int)defects);
    memcpy(text, head.data(), head.size());
    memcpy(text + head.size(), g.out.data(), g.out.size());
    memcpy(text + head.size() + g.out.size(), tail.data(), tail.size());
    return 0;
}
