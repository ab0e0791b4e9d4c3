// TEST INFRASTRUCTURE ONLY: a CPU stand-in for the handful of libbwtgpu entry points c_src/bwt_nif.c calls, so that the shim can be
// run under AddressSanitizer / UBSan in a container without a GPU (sanitizers are not available on the GPU boxes).  It computes
// NOTHING of the reference's path -- its "alignments" are a fixed function of the read's text, chosen to exercise every branch of
// the shim's marshalling (no hit, several hits, hits with extras, both error statuses, empty reads).  bwt_hits_expand is the real
// one (bwt_amd/csrc/pack.cpp is linked in).  The product never links this file.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/bwt_gpu.h"

static thread_local std::string g_err;
int bwt_fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *bwt_last_error(void) { return g_err.c_str(); }
extern "C" void bwt_constants(bwt_constants_t *o) {
    o->tolerance = 20; o->min_len = 8; o->max_range = 45; o->max_len = 15; o->match = 2; o->mismatch = -1; o->gap_penalty = -2;
    o->gap_ext_penalty = -1; o->unknown = 1.6; o->ref_extension_len = 200;
}
struct bwt_index { int dummy; };
struct bwt_small { std::vector<int8_t> st; std::vector<bwt_hit_rec> hits; std::vector<bwt_hit_extra> extra; std::string text; };
static long g_open;
extern "C" long stub_open_indexes(void) { return g_open; }
extern "C" int bwt_index_open(const char *fm, const char *ref, int device, bwt_index **out) {
    (void)ref; (void)device;
    if (strstr(fm, "nonexistent")) return bwt_fail(BWT_EIO, std::string("cannot open ") + fm);
    *out = new bwt_index();
    g_open++;
    return 0;
}
extern "C" void bwt_index_close(bwt_index *ix) { if (ix) { delete ix; g_open--; } }
extern "C" int bwt_index_meta(const bwt_index *, uint32_t out[5]) { out[0] = 10; out[1] = 20; out[2] = 30; out[3] = 40; out[4] = 1; return 0; }
extern "C" int bwt_small_create(bwt_index *, bwt_small **out) { *out = new bwt_small(); return 0; }
extern "C" void bwt_small_destroy(bwt_small *s) { delete s; }
extern "C" int bwt_small_align(bwt_small *s, const uint8_t *bases, const uint64_t *off, uint32_t n, bwt_small_result *res) {
    s->st.assign((size_t)n + 1, 0); s->hits.clear(); s->extra.clear(); s->text.clear();
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t a = off[i], len = off[i + 1] - off[i];
        const uint8_t *q = bases + a;
        if (len < 8) { s->st[i] = 1; continue; }                        // "sga returned []"
        if (memchr(q, 'X', len)) { s->st[i] = -1; continue; }           // "the reference would crash"
        if (q[0] == 'Z') { s->st[i] = -2; continue; }
        const unsigned k = (unsigned)(q[0] % 4);                        // 0..3 hits; every second one with an extra
        for (unsigned h = 0; h < k; h++) {
            uint64_t sum = 0;
            for (uint64_t t = 0; t < len; t++) sum += q[t];                 // (a function of the read's text alone: the same in any batch)
            bwt_hit_rec r; r.read = i; r.score = (int32_t)(2 * len) - (int32_t)h; r.pos = (int64_t)(7 * sum + h);
            if (h & 1) {
                bwt_hit_extra x; x.hit = (uint32_t)s->hits.size(); x.win_len = (uint32_t)len + 3; x.cig_off = (uint32_t)s->text.size();
                const std::string c = std::to_string(len / 2) + "M3D" + std::to_string(len - len / 2) + "M";
                x.cig_len = (uint32_t)c.size();
                s->text += c; s->extra.push_back(x);
            }
            s->hits.push_back(r);
        }
    }
    memset(res, 0, sizeof *res);
    res->n_reads = n; res->path = 1; res->n_hits = s->hits.size(); res->n_extra = s->extra.size(); res->cigar_bytes = s->text.size();
    res->read_status = s->st.data(); res->hits = s->hits.data(); res->extra = s->extra.data(); res->cigar = s->text.data();
    return 0;
}
extern "C" int bwt_ref1_gather(const bwt_index *, const int64_t *win_off, const int32_t *win_len, uint64_t n, uint8_t *out, uint64_t cap,
                               uint64_t *out_off, uint64_t *needed) {
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) { out_off[i] = tot; tot += (uint64_t)win_len[i]; }
    out_off[n] = tot;
    if (needed) *needed = tot;
    if (tot > cap || (!out && tot)) return bwt_fail(BWT_ENOSPC, "RefSeq capacity too small");
    for (uint64_t i = 0; i < n; i++) memset(out + out_off[i], "ACGT"[(uint64_t)win_off[i] & 3], (size_t)win_len[i]);
    return 0;
}
