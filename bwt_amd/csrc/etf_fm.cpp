#include "etf_fm.h"
#include "../../include/bwt_gpu.h"
#include <cstdio>
#include <cstring>

namespace {
struct Cur {
    const uint8_t *p, *end;
    std::string *err;
    bool fail(const char *what) {
        *err = std::string("fm etf: ") + what;
        return false;
    }
    bool need(size_t n) { return (size_t)(end - p) >= n; }
};

inline uint32_t be32(const uint8_t *p) {
    return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
}

bool rd_tuple(Cur &c, uint32_t &arity) {
    if (!c.need(2)) return c.fail("truncated tuple header");
    if (c.p[0] == 104) { arity = c.p[1]; c.p += 2; return true; }
    if (c.p[0] == 105) {
        if (!c.need(5)) return c.fail("truncated large tuple header");
        arity = be32(c.p + 1); c.p += 5; return true;
    }
    return c.fail("tuple expected");
}

bool rd_atom(Cur &c, std::string &a) {
    if (!c.need(2)) return c.fail("truncated atom");
    uint8_t t = c.p[0];
    size_t n, h;
    if (t == 100 || t == 118) {
        if (!c.need(3)) return c.fail("truncated atom");
        n = (size_t)c.p[1] << 8 | c.p[2]; h = 3;
    } else if (t == 115 || t == 119) {
        n = c.p[1]; h = 2;
    } else return c.fail("atom expected");
    if (!c.need(h + n)) return c.fail("truncated atom text");
    a.assign((const char *)c.p + h, n);
    c.p += h + n;
    return true;
}

bool rd_int(Cur &c, int64_t &v) {
    if (!c.need(2)) return c.fail("truncated integer");
    uint8_t t = c.p[0];
    if (t == 97) { v = c.p[1]; c.p += 2; return true; }
    if (t == 98) {
        if (!c.need(5)) return c.fail("truncated integer");
        v = (int32_t)be32(c.p + 1); c.p += 5; return true;
    }
    if (t == 110) {  // SMALL_BIG_EXT
        if (!c.need(3)) return c.fail("truncated bignum");
        size_t n = c.p[1];
        if (n > 8 || !c.need(3 + n)) return c.fail("bignum too wide");
        uint64_t u = 0;
        for (size_t i = 0; i < n; i++) u |= (uint64_t)c.p[3 + i] << (8 * i);
        v = c.p[2] ? -(int64_t)u : (int64_t)u;
        c.p += 3 + n;
        return true;
    }
    return c.fail("integer expected");
}
}  // namespace

int fm_read_etf(const char *path, FmHost &out, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return BWT_EIO; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)sz);
    if (sz > 0 && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f); err = std::string("short read on ") + path; return BWT_EIO;
    }
    fclose(f);
    Cur c{buf.data(), buf.data() + buf.size(), &err};
    if (!c.need(1) || c.p[0] != 131) { err = "fm etf: version byte 131 expected"; return BWT_EFORMAT; }
    c.p++;
    uint32_t ar;
    if (!rd_tuple(c, ar)) return BWT_EFORMAT;
    if (ar != 2) { err = "fm etf: {Meta,FM} expected"; return BWT_EFORMAT; }
    if (!c.need(5) || c.p[0] != 108) { err = "fm etf: Meta list expected"; return BWT_EFORMAT; }
    uint32_t nmeta = be32(c.p + 1);
    c.p += 5;
    bool have_ptr = false, have_shift = false;
    for (uint32_t m = 0; m < nmeta; m++) {
        if (!rd_tuple(c, ar) || ar != 2) { if (err.empty()) err = "fm etf: meta entry"; return BWT_EFORMAT; }
        std::string key;
        if (!rd_atom(c, key)) return BWT_EFORMAT;
        if (key == "pointers") {
            if (!rd_tuple(c, ar) || ar != 4) { err = "fm etf: pointers tuple"; return BWT_EFORMAT; }
            int64_t v[4];
            for (int k = 0; k < 4; k++) if (!rd_int(c, v[k])) return BWT_EFORMAT;
            out.Pc = (uint32_t)v[0]; out.Pg = (uint32_t)v[1]; out.Pt = (uint32_t)v[2]; out.Last = (uint32_t)v[3];
            have_ptr = true;
        } else if (key == "shift") {
            int64_t v;
            if (!rd_int(c, v)) return BWT_EFORMAT;
            out.shift = (int32_t)v;
            have_shift = true;
        } else { err = "fm etf: unknown meta key " + key; return BWT_EFORMAT; }
    }
    if (!c.need(1) || c.p[0] != 106) { err = "fm etf: Meta list tail"; return BWT_EFORMAT; }
    c.p++;
    if (!have_ptr || !have_shift) { err = "fm etf: pointers/shift missing"; return BWT_EFORMAT; }
    uint32_t rows;
    if (!rd_tuple(c, rows)) return BWT_EFORMAT;
    if (rows != out.Last) { err = "fm etf: tuple size != Last"; return BWT_EFORMAT; }
    out.L.resize(rows);
    out.SA.resize(rows);
    // I[r] (0-based LF target) must equal C0[L] + occ(L, r) - 1 with C0 = {$:0, A:1, C:Pc-1, G:Pg-1, T:Pt-1}
    uint64_t occ[256] = {0};
    uint64_t c0[256] = {0};
    c0['A'] = 1; c0['C'] = out.Pc - 1; c0['G'] = out.Pg - 1; c0['T'] = out.Pt - 1;
    uint8_t prevF = 0;
    for (uint32_t r = 0; r < rows; r++) {
        if (!c.need(6) || c.p[0] != 104 || c.p[1] != 4 || c.p[2] != 97 || c.p[4] != 97) {
            err = "fm etf: row is not {F,L,I,SA}"; return BWT_EFORMAT;
        }
        uint8_t F = c.p[3], L = c.p[5];
        c.p += 6;
        int64_t I, SA;
        if (!rd_int(c, I) || !rd_int(c, SA)) return BWT_EFORMAT;
        if (L != '$' && L != 'A' && L != 'C' && L != 'G' && L != 'T') { err = "fm etf: L outside $ACGT"; return BWT_EFORMAT; }
        // '$' (36) < 'A' < 'C' < 'G' < 'T' in ASCII, so F must be non-decreasing
        if (F < prevF) { err = "fm etf: F column not sorted"; return BWT_EFORMAT; }
        prevF = F;
        occ[L]++;
        if ((uint64_t)I != c0[L] + occ[L] - 1) { err = "fm etf: I column violates the LF identity"; return BWT_EFORMAT; }
        if (SA < 0 || SA >= rows) { err = "fm etf: SA out of range"; return BWT_EFORMAT; }
        out.L[r] = L;
        out.SA[r] = (uint32_t)SA;
    }
    if (c.p != c.end) { err = "fm etf: trailing bytes"; return BWT_EFORMAT; }
    if (occ['$'] != 1) { err = "fm etf: L must hold exactly one $"; return BWT_EFORMAT; }
    return 0;
}

void fm_pointers_from_L(FmHost &fm) {
    uint64_t n[256] = {0};
    for (uint8_t c : fm.L) n[c]++;
    fm.Pc = (uint32_t)(2 + n['A']);
    fm.Pg = (uint32_t)(fm.Pc + n['C']);
    fm.Pt = (uint32_t)(fm.Pg + n['G']);
    fm.Last = (uint32_t)fm.L.size();
}

namespace {
void put_int(std::vector<uint8_t> &o, int64_t v) {
    if (v >= 0 && v <= 255) { o.push_back(97); o.push_back((uint8_t)v); return; }
    if (v >= INT32_MIN && v <= INT32_MAX) {
        uint32_t u = (uint32_t)(int32_t)v;
        o.push_back(98); o.push_back(u >> 24); o.push_back(u >> 16); o.push_back(u >> 8); o.push_back(u);
        return;
    }
    uint64_t u = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
    uint8_t d[8]; int n = 0;
    while (u) { d[n++] = (uint8_t)u; u >>= 8; }
    o.push_back(110); o.push_back((uint8_t)n); o.push_back(v < 0);
    for (int i = 0; i < n; i++) o.push_back(d[i]);
}
void put_tuple(std::vector<uint8_t> &o, uint32_t n) {
    if (n <= 255) { o.push_back(104); o.push_back((uint8_t)n); return; }
    o.push_back(105); o.push_back(n >> 24); o.push_back(n >> 16); o.push_back(n >> 8); o.push_back(n);
}
void put_atom(std::vector<uint8_t> &o, const char *a) {
    size_t n = strlen(a);
    o.push_back(100); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n);
    o.insert(o.end(), a, a + n);
}
}  // namespace

int fm_write_etf(const char *path, const FmHost &fm, std::string &err) {
    const uint32_t n = (uint32_t)fm.L.size();
    std::vector<uint8_t> o;
    o.reserve((size_t)n * 16 + 64);
    o.push_back(131);
    put_tuple(o, 2);
    o.push_back(108); o.push_back(0); o.push_back(0); o.push_back(0); o.push_back(2);
    put_tuple(o, 2); put_atom(o, "pointers"); put_tuple(o, 4);
    put_int(o, fm.Pc); put_int(o, fm.Pg); put_int(o, fm.Pt); put_int(o, fm.Last);
    put_tuple(o, 2); put_atom(o, "shift"); put_int(o, fm.shift);
    o.push_back(106);
    put_tuple(o, n);
    // F column: row 1 = '$', then A.., C.., G.., T..   I column from the LF identity
    uint64_t occ[256] = {0}, c0[256] = {0};
    c0['A'] = 1; c0['C'] = fm.Pc - 1; c0['G'] = fm.Pg - 1; c0['T'] = fm.Pt - 1;
    for (uint32_t r = 1; r <= n; r++) {
        uint8_t F = r == 1 ? '$' : r < fm.Pc ? 'A' : r < fm.Pg ? 'C' : r < fm.Pt ? 'G' : 'T';
        uint8_t L = fm.L[r - 1];
        occ[L]++;
        put_tuple(o, 4);
        o.push_back(97); o.push_back(F);
        o.push_back(97); o.push_back(L);
        put_int(o, (int64_t)(c0[L] + occ[L] - 1));
        put_int(o, fm.SA[r - 1]);
    }
    FILE *f = fopen(path, "wb");
    if (!f) { err = std::string("cannot create ") + path; return BWT_EIO; }
    size_t w = fwrite(o.data(), 1, o.size(), f);
    fclose(f);
    if (w != o.size()) { err = std::string("short write on ") + path; return BWT_EIO; }
    return 0;
}
