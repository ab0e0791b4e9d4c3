// libbwtgpu: a compact native index file next to the reference's .fm / .ref.
//
// The reference's on-disk index is the Erlang external term format of a tuple of {F,L,I,SA} rows -- about 16 bytes per row,
// and every small integer a tagged term: a monolithic GRCh38 index would be a ~50 GB term that binary_to_term cannot even
// hold.  The .fm / .ref pair stays the interchange format for the reference's chunk sizes (bwt_index_open / bwt_index_save,
// byte-exact); an index of any size is additionally kept exactly as it sits in HBM:
//
//   header (128 bytes, little endian)
//     0   magic "BWTGPUIX"        8   version = 1 (u32)      12  flags = 0 (u32)
//     16  rows (u32)              20  Pc   24 Pg   28 Pt   32 Last (u32 each)    36 shift (i32)   40 dollar row (u32)
//     44  rank blocks (u32)       48  .ref bytes (u64)
//     56  FNV-1a-64 of the rank-block section   64 ... of the SA section   72 ... of the .ref section
//     80  reserved (zero)
//   rank blocks   nblk x 64 bytes  (layout: kernels_index.h)
//   SA            rows x u32
//   .ref          the chunk's bases (before substitution, as in the reference's .ref)
//
// 5.4 bytes per row; opening it is three sequential reads streamed to the GPU through two pinned buffers, then the two
// derived structures are made on the device as for any other index (Ref1 bit-planes, suffix-interval table).
#include "common.h"

#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr char MAGIC[8] = {'B', 'W', 'T', 'G', 'P', 'U', 'I', 'X'};
constexpr size_t CHUNK = 64u << 20;

struct Header {
    char magic[8];
    uint32_t version, flags;
    uint32_t rows, Pc, Pg, Pt, Last;
    int32_t shift;
    uint32_t dollar, nblk;
    uint64_t ref_len;
    uint64_t h_blocks, h_sa, h_ref;
    uint8_t reserved[48];
};
static_assert(sizeof(Header) == 128, "native index header is 128 bytes");

inline uint64_t fnv1a(uint64_t h, const uint8_t *p, size_t n) {
    // FNV-1a over 8-byte words (the tail byte-wise): a checksum against truncation and bit rot, not a MAC
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
constexpr uint64_t FNV0 = 0xcbf29ce484222325ull;

int header_ok(const Header &h, uint64_t file_size, std::string &err) {
    if (memcmp(h.magic, MAGIC, 8) != 0) { err = "not a native index file (magic)"; return BWT_EFORMAT; }
    if (h.version != 1) { err = "native index: unknown version " + std::to_string(h.version); return BWT_EFORMAT; }
    if (h.rows < 8 || h.Last != h.rows || h.ref_len + 1 != h.rows || h.nblk != h.rows / BWT_BLOCK_ROWS + 1 || h.dollar < 1 || h.dollar > h.rows ||
        !(1 < h.Pc && h.Pc < h.Pg && h.Pg < h.Pt && h.Pt <= h.Last) ||
        (uint64_t)h.rows + 70000ull >= 0xffffffffull) { err = "native index: inconsistent header"; return BWT_EFORMAT; }   // (the invariants of index_build)
    const uint64_t want = sizeof(Header) + (uint64_t)h.nblk * 64 + (uint64_t)h.rows * 4 + h.ref_len;
    if (file_size != want) { err = "native index: file is " + std::to_string(file_size) + " bytes, header says " + std::to_string(want); return BWT_EFORMAT; }
    return 0;
}

// device -> file, in chunks through one pinned buffer
int dump_section(FILE *f, const uint8_t *dev, uint64_t bytes, uint8_t *pin, uint64_t &hash) {
    hash = FNV0;
    for (uint64_t o = 0; o < bytes; o += CHUNK) {
        const size_t n = (size_t)std::min<uint64_t>(CHUNK, bytes - o);
        HIPCHK(hipMemcpy(pin, dev + o, n, hipMemcpyDeviceToHost));
        hash = fnv1a(hash, pin, n);
        if (fwrite(pin, 1, n, f) != n) return fail(BWT_EIO, "native index: short write");
    }
    return 0;
}
// file -> device, double-buffered: the read of chunk k+1 overlaps the upload of chunk k (an event per pinned buffer says when
// its last upload is done)
int load_section(FILE *f, uint8_t *dev, uint64_t bytes, uint8_t *pin[2], hipStream_t st, uint64_t want_hash, const char *what) {
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct EG { hipEvent_t *e; ~EG() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } eg{ev};
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    uint64_t hash = FNV0, k = 0;
    for (uint64_t o = 0; o < bytes; o += CHUNK, k++) {
        const int b = (int)(k & 1);
        const size_t n = (size_t)std::min<uint64_t>(CHUNK, bytes - o);
        if (k >= 2) HIPCHK(hipEventSynchronize(ev[b]));
        if (fread(pin[b], 1, n, f) != n) return fail(BWT_EIO, std::string("native index: short read in ") + what);
        hash = fnv1a(hash, pin[b], n);
        HIPCHK(hipMemcpyAsync(dev + o, pin[b], n, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(ev[b], st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (hash != want_hash) return fail(BWT_EFORMAT, std::string("native index: checksum mismatch in ") + what);
    return 0;
}
}  // namespace

// What the checksums cannot say (they are not keyed: a crafted or stale file can carry consistent ones): the CONTENTS obey the
// invariants the kernels rely on -- every SA value is a row's suffix offset (< rows: it becomes a window offset into Ref1), every
// rank block's counts are the previous block's plus that block's rows, and the totals are the header's pointers.
__global__ __launch_bounds__(256) static void k_native_verify(const uint32_t *__restrict__ blocks, uint32_t nblk, const uint32_t *__restrict__ sa,
                                                               uint32_t rows, uint32_t dollar, uint32_t Pc, uint32_t Pg, uint32_t Pt,
                                                               unsigned int *__restrict__ err) {
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x, t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int bad = 0;
    for (uint64_t r = t0; r < rows; r += nthreads) if (sa[r] >= rows) bad |= 1u;
    for (uint64_t b = t0; b < nblk; b += nthreads) {
        const uint32_t *w = blocks + b * 16;
        uint32_t c[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < BWT_BLOCK_ROWS; i++) {
            const uint64_t row0 = b * BWT_BLOCK_ROWS + i;
            if (row0 >= rows) { if (blk_get_code(w, i)) bad |= 2u; continue; }       // nothing behind the last row
            if (row0 + 1 == dollar) { if (blk_get_code(w, i)) bad |= 2u; continue; } // the $ row is stored as code 0 and not counted
            c[blk_get_code(w, i)]++;
        }
        for (int q = 0; q < 4; q++) {
            const uint32_t have = w[4 * q];
            if (b == 0 && have != 0) bad |= 4u;
            const uint32_t next = b + 1 < nblk ? w[16 + 4 * q] : (q == 0 ? Pc - 2 : q == 1 ? Pg - Pc : q == 2 ? Pt - Pg : rows + 1 - Pt);
            if (have + c[q] != next) bad |= 4u;
        }
    }
    if (bad) atomicOr(err, bad);
}

extern "C" int bwt_index_save_native(const bwt_index *ix, const char *path) {
    if (!ix || !path) return fail(BWT_EINVAL, "bwt_index_save_native: null argument");
    HIPCHK(hipSetDevice(ix->device));
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, MAGIC, 8);
    h.version = 1; h.rows = ix->Last; h.Pc = ix->Pc; h.Pg = ix->Pg; h.Pt = ix->Pt; h.Last = ix->Last; h.shift = ix->shift;
    h.dollar = ix->d.dollar; h.nblk = ix->d.nblk; h.ref_len = ix->ref1_len - 2 * BWT_REF_EXT;
    // written next to its final name and renamed when complete: a crash leaves no half-written index under `path`
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(BWT_EIO, std::string("cannot create ") + tmp);
    struct FG { FILE *f; std::string t; ~FG() { if (f) { fclose(f); remove(t.c_str()); } } } fg{f, tmp};
    uint8_t *pin = nullptr;
    if (hipHostMalloc((void **)&pin, CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(BWT_ENOMEM, "native index: no pinned buffer"); }
    struct PG { uint8_t *p; ~PG() { (void)hipHostFree(p); } } pg{pin};
    if (fwrite(&h, 1, sizeof h, f) != sizeof h) return fail(BWT_EIO, "native index: short write");
    int rc;
    if ((rc = dump_section(f, (const uint8_t *)ix->blocks.p, (uint64_t)h.nblk * 64, pin, h.h_blocks)) ||
        (rc = dump_section(f, (const uint8_t *)ix->sa.p, (uint64_t)h.rows * 4, pin, h.h_sa)) ||
        (rc = dump_section(f, ix->ref1.as<uint8_t>() + BWT_REF_EXT, h.ref_len, pin, h.h_ref)))
        return rc;
    if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&h, 1, sizeof h, f) != sizeof h) return fail(BWT_EIO, "native index: cannot finish the header");
    fg.f = nullptr;
    if (fclose(f) != 0) { remove(tmp.c_str()); return fail(BWT_EIO, "native index: close failed"); }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return fail(BWT_EIO, std::string("native index: cannot rename to ") + path); }
    return 0;
}

static int read_header(const char *path, Header &h, FILE **fout) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(BWT_EIO, std::string("cannot open ") + path);
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail(BWT_EIO, "native index: cannot seek"); }
    const long long size = ftello(f);
    fseek(f, 0, SEEK_SET);
    if (size < (long long)sizeof h || fread(&h, 1, sizeof h, f) != sizeof h) { fclose(f); return fail(BWT_EFORMAT, "native index: no header"); }
    std::string err;
    const int rc = header_ok(h, (uint64_t)size, err);
    if (rc) { fclose(f); return fail(rc, err); }
    *fout = f;
    return 0;
}

// host only: header consistency and the three section checksums -> {Pc, Pg, Pt, Last, Shift}
extern "C" int bwt_native_check(const char *path, uint32_t out[5]) {
    if (!path || !out) return fail(BWT_EINVAL, "bwt_native_check: null argument");
    Header h;
    FILE *f = nullptr;
    int rc = read_header(path, h, &f);
    if (rc) return rc;
    struct FG { FILE *f; ~FG() { fclose(f); } } fg{f};
    std::vector<uint8_t> buf(CHUNK);
    const uint64_t sizes[3] = {(uint64_t)h.nblk * 64, (uint64_t)h.rows * 4, h.ref_len}, want[3] = {h.h_blocks, h.h_sa, h.h_ref};
    const char *names[3] = {"rank blocks", "SA", ".ref"};
    for (int s = 0; s < 3; s++) {
        uint64_t hash = FNV0;
        for (uint64_t o = 0; o < sizes[s]; o += CHUNK) {
            const size_t n = (size_t)std::min<uint64_t>(CHUNK, sizes[s] - o);
            if (fread(buf.data(), 1, n, f) != n) return fail(BWT_EIO, std::string("native index: short read in ") + names[s]);
            hash = fnv1a(hash, buf.data(), n);
        }
        if (hash != want[s]) return fail(BWT_EFORMAT, std::string("native index: checksum mismatch in ") + names[s]);
    }
    out[0] = h.Pc; out[1] = h.Pg; out[2] = h.Pt; out[3] = h.Last; out[4] = (uint32_t)h.shift;
    return 0;
}

extern "C" int bwt_index_open_native(const char *path, int device, bwt_index **out) {
    if (!path || !out) return fail(BWT_EINVAL, "bwt_index_open_native: null argument");
    Header h;
    FILE *f = nullptr;
    int rc = read_header(path, h, &f);
    if (rc) return rc;
    struct FG { FILE *f; ~FG() { fclose(f); } } fg{f};
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BWT_EINVAL, "bwt_index_open_native: no such HIP device");
    HIPCHK(hipSetDevice(device));
    uint8_t *pin[2] = {nullptr, nullptr};
    struct PG { uint8_t **p; ~PG() { for (int i = 0; i < 2; i++) if (p[i]) (void)hipHostFree(p[i]); } } pg{pin};
    for (int i = 0; i < 2; i++)
        if (hipHostMalloc((void **)&pin[i], CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(BWT_ENOMEM, "native index: no pinned buffer"); }
    bwt_index *ix = new bwt_index();
    ix->device = device;
    ix->Pc = h.Pc; ix->Pg = h.Pg; ix->Pt = h.Pt; ix->Last = h.Last; ix->shift = h.shift;
    ix->ref1_len = h.ref_len + 2 * BWT_REF_EXT;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ix; return fail(BWT_EHIP, "hipStreamCreate failed"); }
    if ((rc = ix->blocks.exact((size_t)h.nblk * 64)) || (rc = ix->sa.exact((size_t)h.rows * 4)) || (rc = ix->ref1.exact(ix->ref1_len))) { bwt_index_close(ix); return rc; }
    e = hipMemsetAsync(ix->ref1.p, 'N', ix->ref1_len, ix->stream);                 // the 200-N halo of worker_bwt.erl:54-55
    if (e != hipSuccess) { bwt_index_close(ix); return fail(BWT_EHIP, "native index: memset failed"); }
    if ((rc = load_section(f, (uint8_t *)ix->blocks.p, (uint64_t)h.nblk * 64, pin, ix->stream, h.h_blocks, "rank blocks")) ||
        (rc = load_section(f, (uint8_t *)ix->sa.p, (uint64_t)h.rows * 4, pin, ix->stream, h.h_sa, "SA")) ||
        (rc = load_section(f, ix->ref1.as<uint8_t>() + BWT_REF_EXT, h.ref_len, pin, ix->stream, h.h_ref, ".ref"))) { bwt_index_close(ix); return rc; }
    {
        unsigned int *derr = nullptr, herr = 0;
        if (hipMalloc((void **)&derr, 4) != hipSuccess) { (void)hipGetLastError(); bwt_index_close(ix); return fail(BWT_ENOMEM, "native index: no memory"); }
        (void)hipMemsetAsync(derr, 0, 4, ix->stream);
        hipLaunchKernelGGL(k_native_verify, dim3(2048), dim3(256), 0, ix->stream, (const uint32_t *)ix->blocks.as<uint32_t>(), h.nblk,
                           (const uint32_t *)ix->sa.as<uint32_t>(), h.rows, h.dollar, h.Pc, h.Pg, h.Pt, derr);
        e = hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        (void)hipFree(derr);
        if (e != hipSuccess) { bwt_index_close(ix); return fail(BWT_EHIP, std::string("native index: verification failed to run: ") + hipGetErrorString(e)); }
        if (herr) {
            bwt_index_close(ix);
            return fail(BWT_EFORMAT, std::string("native index: contents break the index invariants (") + ((herr & 1) ? "SA value >= rows; " : "") +
                                         ((herr & 2) ? "code bits outside the rows; " : "") + ((herr & 4) ? "rank-block counts disagree with the rows / pointers" : "") + ")");
        }
    }
    ix->d.blocks = ix->blocks.as<uint4>(); ix->d.sa = ix->sa.as<uint32_t>(); ix->d.ref1 = ix->ref1.as<uint8_t>();
    ix->d.ref1_len = ix->ref1_len;
    ix->d.Pc = h.Pc; ix->d.Pg = h.Pg; ix->d.Pt = h.Pt; ix->d.Last = h.Last; ix->d.dollar = h.dollar; ix->d.nblk = h.nblk;
    ix->d.c0[0] = 1; ix->d.c0[1] = h.Pc - 1; ix->d.c0[2] = h.Pg - 1; ix->d.c0[3] = h.Pt - 1;
    if ((rc = bwt_index_make_refplanes(ix, ix->stream)) || (rc = bwt_index_set_ktable(ix, -1, -1))) { bwt_index_close(ix); return rc; }
    *out = ix;
    return 0;
}
