// Native FASTQ ingest: what fastq:read_seq/1 and read_seqs/2 do
// (/root/reference/src/fastq.erl:5-40), on a memory-mapped file instead of file:read_line/1 on charlists,
// producing the packed batch layout the GPU path consumes (concatenated bases + offsets).
//   read_seq: skip lines until one starts with '@'; Name = that line minus '@' and minus its LAST character
//   (lists:droplast/1 -- the newline, or a real character when the file does not end in one); the next line
//   minus its last character is the sequence; two more lines (delimiter, quality) are consumed unseen.
//   read_seqs(Dev, N): up to N records, PREPENDED to the accumulator => returned in reverse file order.
#include "../../include/bwt_gpu.h"
#include <cstdio>
#include <cstring>
#include <string>
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

// Up to n records.  Records are written in the reference's list order (reverse file order).
// *state: 0 = {ok, Acc} (n records), 1 = eof with no record, 2 = {eof, Acc} (fewer than n, file exhausted).
// Returns BWT_ENOSPC (nothing consumed, needed[] filled) when a capacity is too small.
extern "C" int bwt_fastq_read_seqs(bwt_fastq *f, uint32_t n, uint8_t *bases, uint64_t bases_cap, uint64_t *off,
                                   char *names, uint64_t names_cap, uint64_t *name_off, uint32_t *n_out, int *state,
                                   uint64_t needed[2]) {
    if (!f || !off || !name_off || !n_out || !state) return bwt_fail(BWT_EINVAL, "bwt_fastq_read_seqs: null argument");
    const size_t start = f->pos;
    struct Rec { const char *name; size_t nl; const char *seq; size_t sl; };
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
