// libbwtgpu: the host side of the boundary's wire formats (include/bwt_gpu.h).  Host only: no HIP in this file (the sanitizer
// harness tests/fuzz/host_fuzz.cpp builds it with g++).
//   bwt_pack_reads   ASCII bases -> two bits per base as a low and a high bit-plane word per 32 bases, plus an exception list for
//                    everything outside ACGT;
//   bwt_hits_expand  a compact hit table -> the per-hit arrays of bwt_align_batch (what k_hits_expand does on the device).
//
// Eight bases at a time: with x the 8 ASCII bytes, bit 1 and bit 2 of every byte give the code (A 41 -> 00, C 43 -> 01,
// G 47 -> 10, T 54 -> 11: hi = bit 2, lo = bit 1 ^ bit 2); the byte the code stands for is rebuilt arithmetically
// (41 + 2 lo + 6 hi + 0B (lo & hi)) and compared with x -- equal for all eight bytes is the common case; otherwise the
// group is redone byte by byte and its exceptions listed.  The eight flag bits of a group are gathered by one multiply.
#include "../../include/bwt_gpu.h"
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

int bwt_fail(int code, const std::string &msg);

namespace {
constexpr uint64_t ONES = 0x0101010101010101ull;
inline uint32_t gather8(uint64_t v) { return (uint32_t)((v * 0x0102040810204080ull) >> 56); }   // bit 0 of byte i -> bit i

// bases [a, e) of the stream, a a multiple of 32: the code words of that range; exceptions appended to ex
void pack_range(const uint8_t *bases, uint64_t a, uint64_t e, uint32_t *codes, std::vector<uint64_t> &ex) {
    for (uint64_t w = a / 32; w * 32 < e; w++) {
        uint32_t lo = 0, hi = 0;
        const uint64_t p0 = w * 32;
        for (uint32_t g = 0; g < 4; g++) {
            const uint64_t p = p0 + 8 * g;
            if (p >= e) break;
            uint64_t x;
            if (p + 8 <= e) memcpy(&x, bases + p, 8);
            else {                                            // the stream's tail: 'A' behind the last base (code 0, no exception)
                x = 0x4141414141414141ull;
                memcpy(&x, bases + p, (size_t)(e - p));
            }
            const uint64_t b1 = (x >> 1) & ONES, b2 = (x >> 2) & ONES;
            uint64_t l = b1 ^ b2, h = b2;
            const uint64_t expect = 0x41 * ONES + 2 * l + 6 * h + 0x0B * (l & h);
            if (x != expect) {
                // some byte of the group is not one of ACGT: it gets code 0 and an exception record
                const uint64_t d = x ^ expect;
                for (uint32_t i = 0; i < 8; i++)
                    if ((d >> (8 * i)) & 0xff) {
                        l &= ~(1ull << (8 * i)); h &= ~(1ull << (8 * i));
                        ex.push_back((p + i) | ((x >> (8 * i)) & 0xff) << 56);
                    }
            }
            lo |= gather8(l) << (8 * g); hi |= gather8(h) << (8 * g);
        }
        codes[2 * w] = lo; codes[2 * w + 1] = hi;
    }
}
}  // namespace

extern "C" uint64_t bwt_packed_words(uint64_t total_bases) { return 2 * ((total_bases + 31) / 32) + 2; }

extern "C" int bwt_pack_reads(const uint8_t *bases, uint64_t total, uint32_t *codes, uint64_t *exc, uint64_t exc_cap,
                              uint64_t *n_exc, uint32_t threads) {
    if ((!bases && total) || !codes || (!exc && exc_cap)) return bwt_fail(BWT_EINVAL, "bwt_pack_reads: null argument");
    if (total >> 56) return bwt_fail(BWT_EINVAL, "bwt_pack_reads: too many bases");
    const uint64_t nwords = (total + 31) / 32;
    unsigned T = threads ? threads : std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, nwords / 32768 + 1));        // a thread is worth >= 1 MB of bases
    std::vector<std::vector<uint64_t>> ex(T);
    const uint64_t per = ((nwords + T - 1) / T) * 32;
    if (T == 1) pack_range(bases, 0, total, codes, ex[0]);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const uint64_t a = std::min<uint64_t>(total, t * per), e = std::min<uint64_t>(total, a + per);
                if (a < e) pack_range(bases, a, e, codes, ex[t]);
            });
        for (auto &x : th) x.join();
    }
    codes[2 * nwords] = 0; codes[2 * nwords + 1] = 0;         // (bwt_packed_words leaves room for one empty word behind the last)
    uint64_t k = 0;
    for (auto &v : ex) k += v.size();
    if (n_exc) *n_exc = k;
    if (k > exc_cap) return bwt_fail(BWT_ENOSPC, "bwt_pack_reads: exception capacity too small");
    uint64_t o = 0;
    for (auto &v : ex) { if (!v.empty()) memcpy(exc + o, v.data(), v.size() * 8); o += v.size(); }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The same expansion as k_hits_expand, on the host, for a compact table that has left the device (pipe results, gathered tables).
// Every index and offset of the table is checked against its array: a table from elsewhere (another rank, a file) cannot make this
// read or write out of bounds.
// (one decoder for both forms of the table: the narrow form's reads with status 2 take their hit from the narrow arrays, in read
// order; without them -- read_status null or no 2 in it -- this is the compact table's expansion)
extern "C" int bwt_hits_expand_narrow(uint32_t n, uint32_t read_len, const uint64_t *off, int32_t shift, const int8_t *read_status,
                                      const uint32_t *narrow_off, const int16_t *narrow_score, uint64_t n_narrow, const bwt_hit_rec *hits, uint64_t n_hits,
                                      const bwt_hit_extra *extra, uint64_t n_extra, const char *xcigar, uint64_t xcigar_len, uint64_t *hit_off,
                                      int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len, uint64_t *cig_off, char *cigar,
                                      uint64_t cig_cap, uint64_t *cig_needed) {
    enum { REF_EXT = 200 };                                   // ?REF_EXTENSION_LEN, src/worker_bwt.erl:17 (bwt_constants().ref_extension_len)
    if ((!hits && n_hits) || (!extra && n_extra) || (!xcigar && xcigar_len) || (!read_len && !off && (n_hits || n_narrow)) ||
        (n_narrow && (!read_status || !narrow_off || !narrow_score)))
        return bwt_fail(BWT_EINVAL, "bwt_hits_expand: null argument");
    auto lq_of = [&](uint32_t r) { return read_len ? read_len : (uint32_t)(off[r + 1] - off[r]); };
    auto digits = [](uint32_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; };
    auto plain_text = [&](uint64_t co, uint32_t cl, uint32_t lq) {
        cigar[co + cl - 1] = 'M';
        uint32_t v = lq;
        for (uint32_t x = cl - 1; x-- > 0;) { cigar[co + x] = (char)('0' + v % 10); v /= 10; }
    };
    uint64_t xi = 0, co = 0, g = 0, k = 0, h = 0;             // g: next record of hits[], k: next narrow record, h: next output hit
    for (uint32_t r = 0; r < n; r++) {
        if (g < n_hits && hits[g].read < r) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: hits must be ascending by read");
        if (hit_off) hit_off[r] = h;
        if (n_narrow && read_status[r] == 2) {
            if (k >= n_narrow) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: more reads with status 2 than narrow records");
            if (g < n_hits && hits[g].read == r) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: a read with a narrow record also has records in hits[]");
            const uint32_t lq = lq_of(r), cl = digits(lq) + 1;
            if (score) score[h] = narrow_score[k];
            if (pos) pos[h] = (int64_t)narrow_off[k] - REF_EXT + shift;
            if (win_off) win_off[h] = narrow_off[k];
            if (win_len) win_len[h] = (int32_t)lq;
            if (cig_off) cig_off[h] = co;
            if (cigar && co + cl <= cig_cap) plain_text(co, cl, lq);
            co += cl; k++; h++;
            continue;
        }
        for (; g < n_hits && hits[g].read == r; g++, h++) {
            const bwt_hit_rec &hr = hits[g];
            const bool ex = xi < n_extra && extra[xi].hit == g;
            if (ex && ((uint64_t)extra[xi].cig_off + extra[xi].cig_len > xcigar_len))
                return bwt_fail(BWT_EINVAL, "bwt_hits_expand: an extra's text lies outside the text buffer");
            const uint32_t lq = lq_of(hr.read), cl = ex ? extra[xi].cig_len : digits(lq) + 1;
            if (score) score[h] = hr.score;
            if (pos) pos[h] = hr.pos;
            if (win_off) win_off[h] = hr.pos - shift + REF_EXT;
            if (win_len) win_len[h] = (int32_t)(ex ? extra[xi].win_len : lq);
            if (cig_off) cig_off[h] = co;
            if (cigar && co + cl <= cig_cap) {
                if (ex) memcpy(cigar + co, xcigar + extra[xi].cig_off, cl);
                else plain_text(co, cl, lq);
            }
            co += cl;
            if (ex) xi++;
        }
    }
    if (g != n_hits) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: hits must be ascending by read and name reads of the batch");
    if (k != n_narrow) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: fewer reads with status 2 than narrow records");
    if (xi != n_extra) return bwt_fail(BWT_EINVAL, "bwt_hits_expand: extras must be ascending by hit and name existing hits");
    if (hit_off) hit_off[n] = h;
    if (cig_off) cig_off[h] = co;
    if (cig_needed) *cig_needed = co;
    if (cigar && co > cig_cap) return bwt_fail(BWT_ENOSPC, "cigar capacity too small");
    return 0;
}
extern "C" int bwt_hits_expand(uint32_t n, uint32_t read_len, const uint64_t *off, int32_t shift, const bwt_hit_rec *hits, uint64_t n_hits,
                               const bwt_hit_extra *extra, uint64_t n_extra, const char *xcigar, uint64_t xcigar_len, uint64_t *hit_off,
                               int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len, uint64_t *cig_off, char *cigar,
                               uint64_t cig_cap, uint64_t *cig_needed) {
    return bwt_hits_expand_narrow(n, read_len, off, shift, nullptr, nullptr, nullptr, 0, hits, n_hits, extra, n_extra, xcigar, xcigar_len, hit_off, score,
                                  pos, win_off, win_len, cig_off, cigar, cig_cap, cig_needed);
}
