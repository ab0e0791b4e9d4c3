// Native FASTQ ingest: what fastq:read_seq/1 and read_seqs/2 do
// (/root/reference/src/fastq.erl:5-40), on a memory-mapped file instead of file:read_line/1 on charlists,
// producing the packed batch layout the GPU path consumes (concatenated bases + offsets).
//   read_seq: skip lines until one starts with '@'; Name = that line minus '@' and minus its LAST character
//   (lists:droplast/1 -- the newline, or a real character when the file does not end in one); the next line
//   minus its last character is the sequence; two more lines (delimiter, quality) are consumed unseen.
//   read_seqs(Dev, N): up to N records, PREPENDED to the accumulator => returned in reverse file order.
#include "../../include/bwt_gpu.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

int bwt_fail(int code, const std::string &msg);

struct bwt_fastq {
    int fd = -1;
    const char *p = nullptr;
    size_t len = 0, pos = 0;
    unsigned threads = 0;            // 0 = by batch size and core count; 1 = always the sequential reader
    size_t par_min = 65536;          // records from which a call is worth spreading over threads
};

extern "C" int bwt_fastq_open(const char *path, bwt_fastq **out) {
    if (!path || !out) return bwt_fail(BWT_EINVAL, "bwt_fastq_open: null argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return bwt_fail(BWT_EIO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st)) { close(fd); return bwt_fail(BWT_EIO, "fstat failed"); }
    bwt_fastq *f = new bwt_fastq();
    f->fd = fd; f->len = (size_t)st.st_size;
    if (f->len) {
        void *m = mmap(nullptr, f->len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); delete f; return bwt_fail(BWT_EIO, "mmap failed"); }
        f->p = (const char *)m;
    }
    *out = f;
    return 0;
}

extern "C" void bwt_fastq_close(bwt_fastq *f) {
    if (!f) return;
    if (f->p) munmap((void *)f->p, f->len);
    if (f->fd >= 0) close(f->fd);
    delete f;
}

// file:read_line/1: the line including its '\n' (the last line may lack one); false at eof
static bool read_line(bwt_fastq *f, const char *&s, size_t &n) {
    if (f->pos >= f->len) return false;
    const char *b = f->p + f->pos;
    const char *e = (const char *)memchr(b, '\n', f->len - f->pos);
    n = e ? (size_t)(e - b) + 1 : f->len - f->pos;
    s = b;
    f->pos += n;
    return true;
}

struct Rec { const char *name; size_t nl; const char *seq; size_t sl; };

// The same result from many threads, for the files every FASTQ writer produces: each record four lines, the first starting
// with '@'.  Then the k-th record of a call starts at line 4k from the current position, and the work splits cleanly:
//   1. the bytes ahead are cut into slices; every thread counts the newlines of its slice;
//   2. from the prefix sums: where line 4n (the end of the batch) lies, and for every slice the number of its first line;
//   3. every thread takes the records whose header line STARTS in its slice: four line ends per record by memchr, the header's
//      '@' checked, lengths summed; after a prefix sum over the threads' totals each thread copies its records to their final
//      place (reverse file order, as the accumulator of read_seqs/3 leaves them).
// Anything else -- a line where a header is due that does not start with '@' (the reference then skips lines until one does),
// a file that ends inside a record -- returns false and the sequential reader decides.
static bool read_seqs_parallel(bwt_fastq *f, uint32_t n, unsigned T, std::vector<std::vector<Rec>> &per, size_t &end_pos, bool &eof) {
    const char *base = f->p + f->pos;
    const size_t avail = f->len - f->pos;
    if (!avail) return false;
    const auto t_0 = std::chrono::steady_clock::now();
    // a generous bound of the bytes n records take: look at the first records
    size_t look = std::min<size_t>(avail, 1u << 16), nl = 0;
    for (size_t i = 0; i < look; i++) nl += base[i] == '\n';
    if (nl < 8) return false;
    const double per_rec = (double)look / ((double)nl / 4.0);
    size_t span = (size_t)std::min<double>((double)avail, per_rec * 1.25 * (double)n + (1u << 16));
    for (int attempt = 0; attempt < 8; attempt++) {
        const size_t slice = (span + T - 1) / T;
        std::vector<size_t> cnt(T, 0);
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    const size_t a = std::min(span, t * slice), b = std::min(span, a + slice);
                    size_t c = 0;
                    const char *p = base + a, *e = base + b;
                    for (; p < e; p++) c += *p == '\n';                        // (vectorised: lines are too short for a memchr per line)
                    cnt[t] = c;
                });
            for (auto &x : th) x.join();
        }
        size_t lines = 0;
        for (auto c : cnt) lines += c;
        const auto t_1 = std::chrono::steady_clock::now();
        const bool whole = span == avail;
        // lines ahead (a last line without '\n' counts as one)
        const size_t total_lines = lines + ((whole && avail && base[avail - 1] != '\n') ? 1 : 0);
        if (total_lines < 4ull * n && !whole) { span = std::min(avail, span * 2); continue; }     // the slice estimate was short
        const size_t nrec = std::min<size_t>(n, total_lines / 4);
        if (whole && total_lines % 4 != 0 && total_lines / 4 < n) return false;                    // trailing partial record: sequential rules
        eof = nrec < n;
        // line number at the start of every slice
        std::vector<size_t> first_line(T + 1, 0);
        for (unsigned t = 0; t < T; t++) first_line[t + 1] = first_line[t] + cnt[t];
        per.assign(T, {});
        std::vector<char> bad(T, 0);
        std::vector<size_t> endp(T, 0);
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    const size_t a = std::min(span, t * slice), b = std::min(span, a + slice);
                    // the first line that STARTS in [a, b): line first_line[t] if a is a line start, else the one after
                    const char *p = base + a;
                    size_t ln = first_line[t];
                    if (a > 0 && base[a - 1] != '\n') {
                        const char *q = (const char *)memchr(p, '\n', avail - a);
                        if (!q) return;
                        p = q + 1; ln++;
                    }
                    // advance to the next header line (line number a multiple of 4)
                    const char *lim = base + avail;
                    while (ln % 4 != 0) {
                        if (p >= base + b) return;
                        const char *q = (const char *)memchr(p, '\n', (size_t)(lim - p));
                        if (!q) return;
                        p = q + 1; ln++;
                    }
                    auto &out = per[t];
                    while (p < base + b && ln / 4 < nrec) {
                        const char *l[4]; size_t ll[4];
                        const char *c = p;
                        for (int k = 0; k < 4; k++) {
                            if (c >= lim) { bad[t] = 1; return; }
                            const char *q = (const char *)memchr(c, '\n', (size_t)(lim - c));
                            l[k] = c; ll[k] = q ? (size_t)(q - c) + 1 : (size_t)(lim - c);
                            c += ll[k];
                            if (!q && k < 3) { bad[t] = 1; return; }
                        }
                        if (l[0][0] != '@') { bad[t] = 1; return; }
                        out.push_back({l[0] + 1, ll[0] - 1 ? ll[0] - 2 : 0, l[1], ll[1] ? ll[1] - 1 : 0});
                        if (ln / 4 + 1 == nrec) endp[t] = (size_t)(c - base);
                        p = c; ln += 4;
                    }
                });
            for (auto &x : th) x.join();
        }
        size_t got = 0;
        for (unsigned t = 0; t < T; t++) { if (bad[t]) { if (getenv("BWT_FASTQ_DEBUG")) fprintf(stderr, "fastq par: thread %u bad\n", t); return false; } got += per[t].size(); end_pos = std::max(end_pos, endp[t]); }
        if (got != nrec) { if (getenv("BWT_FASTQ_DEBUG")) fprintf(stderr, "fastq par: got %zu of %zu records, span %zu avail %zu lines %zu\n", got, nrec, span, avail, total_lines); return false; }
        if (nrec == 0) return false;
        if (getenv("BWT_FASTQ_DEBUG")) fprintf(stderr, "fastq par: count %.2f ms, parse %.2f ms (span %zu MB, T %u)\n", std::chrono::duration<double, std::milli>(t_1 - t_0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_1).count(), span >> 20, T);
        return true;
    }
    return false;
}

// Up to n records.  Records are written in the reference's list order (reverse file order).
// *state: 0 = {ok, Acc} (n records), 1 = eof with no record, 2 = {eof, Acc} (fewer than n, file exhausted).
// Returns BWT_ENOSPC (nothing consumed, needed[] filled) when a capacity is too small.
extern "C" int bwt_fastq_read_seqs(bwt_fastq *f, uint32_t n, uint8_t *bases, uint64_t bases_cap, uint64_t *off,
                                   char *names, uint64_t names_cap, uint64_t *name_off, uint32_t *n_out, int *state,
                                   uint64_t needed[2]) {
    if (!f || !off || !name_off || !n_out || !state) return bwt_fail(BWT_EINVAL, "bwt_fastq_read_seqs: null argument");
    const size_t start = f->pos;
    // big batches: many threads (the sequential reader below handles ~10 M records/s; a GPU takes them 40 times faster)
    unsigned T = f->threads ? f->threads : std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    if (!f->threads) T = (unsigned)std::max<size_t>(1, std::min<size_t>(T, n / 16384));      // a thread is worth >= 16 k records
    if (T > 1 && n >= f->par_min) {
        std::vector<std::vector<Rec>> per;
        size_t end_pos = 0;
        bool eof_p = false;
        if (read_seqs_parallel(f, n, T, per, end_pos, eof_p)) {
            std::vector<uint64_t> tb(T + 1, 0), tn(T + 1, 0), tc(T + 1, 0);
            for (unsigned t = 0; t < T; t++) {
                uint64_t b = 0, m = 0;
                for (auto &r : per[t]) { b += r.sl; m += r.nl; }
                tb[t + 1] = tb[t] + b; tn[t + 1] = tn[t] + m; tc[t + 1] = tc[t] + per[t].size();
            }
            const uint64_t nb = tb[T], nn = tn[T], k = tc[T];
            if (needed) { needed[0] = nb; needed[1] = nn; }
            if (nb > bases_cap || nn > names_cap || (!bases && nb) || (!names && nn)) return bwt_fail(BWT_ENOSPC, "FASTQ batch capacity too small");
            // reverse file order: record i (file order) lands at index k-1-i; its bytes end where the bytes of the records after
            // it ... begin: offsets counted from the END of the file-order prefix sums
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    uint64_t bo = nb - tb[t], no = nn - tn[t];                 // end of this thread's first record in the output
                    uint64_t idx = k - tc[t];                                  // one past its output index
                    for (auto &r : per[t]) {
                        bo -= r.sl; no -= r.nl; idx--;
                        off[idx] = bo; name_off[idx] = no;
                        memcpy(bases + bo, r.seq, r.sl);
                        memcpy(names + no, r.name, r.nl);
                    }
                });
            for (auto &x : th) x.join();
            off[k] = nb; name_off[k] = nn;
            f->pos = start + end_pos;
            *n_out = (uint32_t)k;
            *state = !eof_p ? 0 : 2;
            // the sequential reader would now see eof on its next line read only if nothing but the records consumed remains
            if (eof_p) f->pos = f->len;
            return 0;
        }
        f->pos = start;
    }
    std::vector<Rec> recs;
    recs.reserve(n);
    bool eof = false;
    while (recs.size() < n) {
        const char *s; size_t l;
        if (!read_line(f, s, l)) { eof = true; break; }
        if (l == 0 || s[0] != '@') continue;                     // fastq.erl:14-15: not a header, keep looking
        const char *q; size_t ql;
        if (!read_line(f, q, ql)) {                              // {ok, SeqData} = eof: the reference badmatches
            f->pos = start;
            return bwt_fail(BWT_EFORMAT, "FASTQ ends inside a record (header without a sequence line)");
        }
        const char *d; size_t dl;
        read_line(f, d, dl);                                     // delimiter and quality: results ignored (fastq.erl:9-10)
        read_line(f, d, dl);
        recs.push_back({s + 1, l - 1 ? l - 2 : 0, q, ql ? ql - 1 : 0});   // droplast on both
    }
    uint64_t nb = 0, nn = 0;
    for (auto &r : recs) { nb += r.sl; nn += r.nl; }
    if (needed) { needed[0] = nb; needed[1] = nn; }
    if (nb > bases_cap || nn > names_cap || (!bases && nb) || (!names && nn)) {
        f->pos = start;
        return bwt_fail(BWT_ENOSPC, "FASTQ batch capacity too small");
    }
    uint64_t bo = 0, no = 0;
    uint32_t k = 0;
    for (size_t i = recs.size(); i-- > 0; k++) {                 // prepend accumulation => reverse order
        off[k] = bo; name_off[k] = no;
        memcpy(bases + bo, recs[i].seq, recs[i].sl); bo += recs[i].sl;
        memcpy(names + no, recs[i].name, recs[i].nl); no += recs[i].nl;
    }
    off[k] = bo; name_off[k] = no;
    *n_out = k;
    *state = !eof ? 0 : (k == 0 ? 1 : 2);
    return 0;
}

// threads: 0 = automatic, 1 = sequential reader only; min_records: batch size from which the threaded reader is used
extern "C" int bwt_fastq_set_threads(bwt_fastq *f, uint32_t threads, uint64_t min_records) {
    if (!f) return bwt_fail(BWT_EINVAL, "bwt_fastq_set_threads: null handle");
    f->threads = threads; f->par_min = (size_t)min_records;
    return 0;
}
