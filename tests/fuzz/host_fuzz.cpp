// TEST INFRASTRUCTURE: AddressSanitizer / UBSan harness for the host-side parsers of libbwtgpu that read files a user hands in:
// the .fm reader (Erlang external term format, bwt_amd/csrc/etf_fm.cpp) and the FASTQ reader (bwt_amd/csrc/fastq.cpp, both the
// sequential and the threaded path).  Built by tests/test_host_fuzz.py with g++ -fsanitize=address,undefined from the library's own
// sources (no GPU, no HIP): a seeded stream of mutated inputs (bit flips, truncations, splices) must be either parsed or refused --
// never read out of bounds.
//   host_fuzz <fm file> <rounds> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "../../bwt_amd/csrc/etf_fm.h"
#include "../../include/bwt_gpu.h"

static std::string g_err;
int bwt_fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *bwt_last_error(void) { return g_err.c_str(); }

static uint64_t rng_state;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}
static void spit(const char *p, const std::vector<uint8_t> &v) {
    FILE *f = fopen(p, "wb");
    if (v.size()) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
}
static void mutate(std::vector<uint8_t> &v) {
    const int kind = (int)(rnd() % 5);
    if (v.empty()) return;
    if (kind == 0) v.resize((size_t)(rnd() % v.size()));                                   // truncate
    else if (kind == 1) for (int i = 0; i < 1 + (int)(rnd() % 8); i++) v[rnd() % v.size()] ^= (uint8_t)(1u << (rnd() % 8));
    else if (kind == 2) { size_t a = rnd() % v.size(), n = rnd() % 64; for (size_t i = a; i < v.size() && i < a + n; i++) v[i] = (uint8_t)rnd(); }
    else if (kind == 3) { size_t a = rnd() % v.size(), b = rnd() % v.size(), n = rnd() % 4096; if (a + n < v.size() && b + n < v.size()) memmove(&v[a], &v[b], n); }
    else { size_t a = rnd() % v.size(); v.erase(v.begin() + (long)a, v.begin() + (long)std::min(v.size(), a + 1 + rnd() % 200)); }
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int rounds = atoi(argv[2]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[3]);
    const std::vector<uint8_t> fm0 = slurp(argv[1]);
    if (fm0.empty()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    char tmp[256];
    snprintf(tmp, sizeof tmp, "/tmp/host_fuzz_%d.bin", (int)rnd() & 0xffffff);
    int ok_fm = 0, bad_fm = 0, ok_fq = 0, bad_fq = 0;
    // ---- .fm: a small prefix of the fixture keeps the rounds fast (the row loop is the same code for every row)
    for (int r = 0; r < rounds; r++) {
        std::vector<uint8_t> v = fm0;
        if (r % 4) v.resize(std::min<size_t>(v.size(), 4096 + rnd() % 60000));
        for (int k = 0; k < 1 + (int)(rnd() % 3); k++) mutate(v);
        spit(tmp, v);
        FmHost fm; std::string err;
        (fm_read_etf(tmp, fm, err) == 0 ? ok_fm : bad_fm)++;
    }
    {   // the untouched fixture parses
        FmHost fm; std::string err;
        if (fm_read_etf(argv[1], fm, err) != 0) { fprintf(stderr, "fixture refused: %s\n", err.c_str()); return 1; }
    }
    // ---- FASTQ: well-formed text, then mutated; sequential and threaded readers, ragged batch sizes
    std::vector<uint8_t> fq0;
    for (int i = 0; i < 3000; i++) {
        char line[512];
        const int L = 20 + (int)(rnd() % 120);
        int n = snprintf(line, sizeof line, "@read.%d extra\n", i);
        fq0.insert(fq0.end(), line, line + n);
        for (int k = 0; k < L; k++) fq0.push_back((uint8_t)"ACGTN"[rnd() % 5]);
        fq0.push_back('\n'); fq0.push_back('+'); fq0.push_back('\n');
        for (int k = 0; k < L; k++) fq0.push_back((uint8_t)"@IJ#+"[rnd() % 5]);
        fq0.push_back('\n');
    }
    for (int r = 0; r < rounds; r++) {
        std::vector<uint8_t> v = fq0;
        if (r) for (int k = 0; k < 1 + (int)(rnd() % 3); k++) mutate(v);
        spit(tmp, v);
        for (int threads = 1; threads <= 7; threads += 3) {
            bwt_fastq *f = nullptr;
            if (bwt_fastq_open(tmp, &f)) continue;
            bwt_fastq_set_threads(f, (uint32_t)threads, threads == 1 ? ~0ull : 1);
            for (int call = 0; call < 12; call++) {
                const uint32_t n = 1 + (uint32_t)(rnd() % 900);
                std::vector<uint8_t> bases(n * 160 + 64), names(n * 64 + 64);
                std::vector<uint64_t> off(n + 1), noff(n + 1);
                uint32_t k = 0; int st = 0; uint64_t need[2] = {0, 0};
                int rc = bwt_fastq_read_seqs(f, n, bases.data(), bases.size(), off.data(), (char *)names.data(), names.size(), noff.data(), &k, &st, need);
                if (rc == BWT_ENOSPC) {
                    bases.resize(need[0] + 16); names.resize(need[1] + 16);
                    rc = bwt_fastq_read_seqs(f, n, bases.data(), bases.size(), off.data(), (char *)names.data(), names.size(), noff.data(), &k, &st, need);
                }
                if (rc) { bad_fq++; break; }
                ok_fq++;
                if (k > n || off[k] > bases.size() || noff[k] > names.size()) { fprintf(stderr, "reader overran its outputs\n"); return 1; }
                if (st != 0) break;
            }
            bwt_fastq_close(f);
        }
    }
    remove(tmp);
    // ---- the wire formats (bwt_amd/csrc/pack.cpp): the packer on arbitrary bytes with EXACTLY sized buffers (any overrun is an ASan
    // report), its output against a byte-by-byte restatement; the table expansion on tables with every field mutated
    int ok_pk = 0, ok_ex = 0, bad_ex = 0;
    for (int r = 0; r < rounds * 2; r++) {
        const size_t total = r < 70 ? (size_t)r : (size_t)(rnd() % 30000);
        std::vector<uint8_t> b(total);
        for (auto &c : b) c = (rnd() % 13) ? (uint8_t)"ACGT"[rnd() % 4] : (uint8_t)rnd();
        std::vector<uint32_t> codes((size_t)bwt_packed_words(total), 0xdeadbeefu);
        uint64_t nexc = 0;
        std::vector<uint64_t> exc(4);
        int rc = bwt_pack_reads(b.data(), total, codes.data(), exc.data(), exc.size(), &nexc, 1 + (uint32_t)(rnd() % 5));
        if (rc == BWT_ENOSPC) {
            exc.assign((size_t)nexc, 0);
            rc = bwt_pack_reads(b.data(), total, codes.data(), exc.data(), exc.size(), &nexc, 1 + (uint32_t)(rnd() % 5));
        }
        if (rc) { fprintf(stderr, "bwt_pack_reads failed: %s\n", g_err.c_str()); return 1; }
        size_t xi = 0;
        for (size_t i = 0; i < total; i++) {
            const uint8_t c = b[i];
            const int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
            const uint32_t lo = (codes[2 * (i / 32)] >> (i % 32)) & 1u, hi = (codes[2 * (i / 32) + 1] >> (i % 32)) & 1u;
            if (code < 0) {
                if (xi >= nexc || exc[xi] != ((uint64_t)i | (uint64_t)c << 56) || lo || hi) { fprintf(stderr, "packer: exception %zu wrong\n", i); return 1; }
                xi++;
            } else if ((int)(lo | hi << 1) != code) { fprintf(stderr, "packer: code %zu wrong\n", i); return 1; }
        }
        if (xi != nexc) { fprintf(stderr, "packer: exception count\n"); return 1; }
        ok_pk++;
    }
    for (int r = 0; r < rounds * 20; r++) {
        // a well-formed table of nh hits over n reads, some with extras; then mutate any field
        const uint32_t n = 1 + (uint32_t)(rnd() % 40);
        const uint64_t nh = rnd() % 60;
        std::vector<uint64_t> off(n + 1, 0);
        for (uint32_t i = 0; i < n; i++) off[i + 1] = off[i] + 20 + rnd() % 300;
        std::vector<bwt_hit_rec> hits((size_t)nh);
        std::vector<bwt_hit_extra> extra;
        std::string text;
        uint32_t rd = 0;
        for (uint64_t h = 0; h < nh; h++) {
            rd = std::min<uint32_t>(n - 1, rd + (uint32_t)(rnd() % 3));
            hits[(size_t)h] = {rd, (int32_t)(rnd() % 400), (int64_t)(rnd() % 100000) - 300};
            if (rnd() % 3 == 0) { const std::string t = std::to_string(1 + rnd() % 90) + "M2D" + std::to_string(1 + rnd() % 90) + "M"; extra.push_back({(uint32_t)h, (uint32_t)(100 + rnd() % 30), (uint32_t)text.size(), (uint32_t)t.size()}); text += t; }
        }
        const bool mutated = r % 3 != 0;
        if (mutated) {
            const int what = (int)(rnd() % 6);
            if (what == 0 && nh) hits[(size_t)(rnd() % nh)].read = (uint32_t)rnd();
            else if (what == 1 && !extra.empty()) extra[rnd() % extra.size()].hit = (uint32_t)rnd();
            else if (what == 2 && !extra.empty()) extra[rnd() % extra.size()].cig_off = (uint32_t)rnd();
            else if (what == 3 && !extra.empty()) extra[rnd() % extra.size()].cig_len = (uint32_t)rnd();
            else if (what == 4 && nh > 1) std::swap(hits[0], hits[(size_t)nh - 1]);
            else if (!extra.empty()) extra.pop_back(), extra.insert(extra.begin(), {(uint32_t)nh + 5, 100, 0, 1});
        }
        std::vector<uint64_t> hoff(n + 1), coff((size_t)nh + 1);
        std::vector<int32_t> sc((size_t)nh), wl((size_t)nh);
        std::vector<int64_t> ps((size_t)nh), wo((size_t)nh);
        uint64_t need = 0;
        int rc = bwt_hits_expand(n, 0, off.data(), 1, hits.data(), nh, extra.data(), extra.size(), text.data(), text.size(), hoff.data(), sc.data(),
                                 ps.data(), wo.data(), wl.data(), coff.data(), nullptr, 0, &need);
        if (rc == 0) {
            std::vector<char> cg((size_t)need);                    // exactly sized: an overrun is an ASan report
            rc = bwt_hits_expand(n, 0, off.data(), 1, hits.data(), nh, extra.data(), extra.size(), text.data(), text.size(), hoff.data(), sc.data(),
                                 ps.data(), wo.data(), wl.data(), coff.data(), cg.data(), need, &need);
            if (rc) { fprintf(stderr, "bwt_hits_expand: second call failed\n"); return 1; }
            ok_ex++;
        } else bad_ex++;
        if (!mutated && rc) { fprintf(stderr, "bwt_hits_expand refused a well-formed table: %s\n", g_err.c_str()); return 1; }
    }
    printf("fm: %d parsed, %d refused; fastq calls: %d ok, %d refused; packer: %d streams ok; table expansion: %d ok, %d refused\n", ok_fm, bad_fm, ok_fq, bad_fq,
           ok_pk, ok_ex, bad_ex);
    return 0;
}
