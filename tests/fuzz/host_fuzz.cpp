// TEST INFRASTRUCTURE: AddressSanitizer / UBSan harness for the host-side parsers of libbwtgpu that read files a user hands in:
// the .fm reader (Erlang external term format, bwt_amd/csrc/etf_fm.cpp) and the FASTQ reader (bwt_amd/csrc/fastq.cpp, both the
// sequential and the threaded path).  Built by tests/test_host_fuzz.py with g++ -fsanitize=address,undefined from the library's own
// sources (no GPU, no HIP): a seeded stream of mutated inputs (bit flips, truncations, splices) must be either parsed or refused --
// never read out of bounds.
//   host_fuzz <fm file> <rounds> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
    printf("fm: %d parsed, %d refused; fastq calls: %d ok, %d refused\n", ok_fm, bad_fm, ok_fq, bad_fq);
    return 0;
}
