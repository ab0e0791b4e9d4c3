// libbwtgpu: bwt_run_fastq -- the whole job the reference's master and one worker do for a FASTQ file, natively:
//   master:produce_workload (src/master.erl:162-178: lists of 200 reads, ten lists per workload), the worker's `running` clause
//   for every list (src/worker_bwt.erl:73-112: sga + sw per read, then the pick among several alignments by lists:sort/2 with the
//   comparator R1 > R2, :99-106), and the master's result sink (src/master.erl:149-155: a line per cast whose score reaches the
//   threshold) -- with the casts in the order the reference sends them: workloads in file order, inside a workload the lists
//   newest first (produce_workload prepends), inside a list the reads in file order (read_seqs prepends, the worker's fold
//   prepends again).
// FASTQ records are parsed by the threaded reader into a pipe slot's staging buffer, packed to two bits per base by the slot's own
// thread (bwt_pipe_submit_ascii) and uploaded; the compact hit tables that come back (bwt_pipe: upload, kernels and download
// of consecutive batches overlap) are picked and printed on the host.  Transport (navel) and the gen_server shell stay out of scope.
#include "common.h"
#include "otp_sort.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct SlotHost { std::vector<char> names; std::vector<uint64_t> name_off; const uint64_t *off = nullptr; uint32_t n = 0; int state = 0; };

inline char *put_int(char *p, long long v) {
    char tmp[24];
    int k = 0;
    unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (k) *p++ = tmp[--k];
    return p;
}
}  // namespace

extern "C" int bwt_otp_pick(const int32_t *scores, uint32_t n) {
    // index of the head of lists:sort(fun(A, B) -> score(A) > score(B) end, L) for L = the candidates in list order
    if (!scores || !n) return -1;
    if (n == 1) return 0;
    otp::L l(n);
    for (uint32_t i = 0; i < n; i++) l[i] = (int)i;
    auto fun = [&](int a, int b) { return scores[a] > scores[b]; };
    return otp::sort(fun, l)[0];
}

extern "C" int bwt_run_fastq(bwt_index *ix, const char *fastq_path, const char *chromosome, int32_t threshold, uint32_t batch_reads,
                             const char *out_path, bwt_run_stats *stats) {
    if (!ix || !fastq_path || !chromosome) return fail(BWT_EINVAL, "bwt_run_fastq: null argument");
    const uint32_t LIST = 200, LISTS = 10, WL = LIST * LISTS;              // master.erl:59, worker_bwt.erl:16
    if (!batch_reads) batch_reads = 2000000;
    batch_reads = std::max<uint32_t>(WL, batch_reads / WL * WL);           // whole workloads per batch: the cast order does not straddle batches
    const auto t0 = std::chrono::steady_clock::now();
    bwt_run_stats st;
    memset(&st, 0, sizeof st);
    bwt_fastq *fq = nullptr;
    int rc = bwt_fastq_open(fastq_path, &fq);
    if (rc) return rc;
    struct FQG { bwt_fastq *f; ~FQG() { bwt_fastq_close(f); } } fqg{fq};
    FILE *out = nullptr;
    if (out_path && !(out = fopen(out_path, "wb"))) return fail(BWT_EIO, std::string("cannot create ") + out_path);
    struct OG { FILE *f; ~OG() { if (f) fclose(f); } } og{out};
    const uint32_t depth = 3;
    uint32_t meta[5];
    if ((rc = bwt_index_meta(ix, meta))) return rc;
    bwt_pipe *pipe = nullptr;
    if ((rc = bwt_pipe_create(ix, depth, &pipe))) return rc;
    struct PG { bwt_pipe *p; ~PG() { bwt_pipe_destroy(p); } } pg{pipe};
    std::vector<SlotHost> host(depth);
    const size_t clen = strlen(chromosome);
    uint64_t bases_cap = (uint64_t)batch_reads * 160 + 4096;
    bool eof = false;
    uint32_t inflight = 0;
    std::vector<char> text;
    std::vector<uint32_t> sel;            // per read of the batch: the picked hit, or ~0u
    std::vector<int64_t> gwoff; std::vector<int32_t> gwlen; std::vector<uint64_t> goff; std::vector<uint8_t> gwin;
    std::vector<int32_t> sc;
    std::vector<uint64_t> hoff;           // per read of the batch: where its hits start in the compact table
    std::vector<uint32_t> xof;            // per printed hit: its extra record, or ~0u (plain: "<Lq>M" on a window of Lq bases)

    double t_parse = 0, t_wait = 0, t_pick = 0, t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto consume = [&]() -> int {
        bwt_pipe_result r;
        const auto tw0 = now();
        int rc2 = bwt_pipe_wait(pipe, &r);
        const auto tw1 = now();
        t_wait += secs(tw0, tw1);
        struct PickTimer { double &acc; std::chrono::steady_clock::time_point a; ~PickTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); } } pt{t_pick, tw1};
        if (rc2) return rc2;
        SlotHost &h = host[r.slot];
        const uint32_t n = r.n_reads;
        // the pick (worker_bwt.erl:99-106) per read; array index i is file-order record n-1-i of the batch
        sel.assign(n, ~0u);
        uint64_t nsel = 0;
        // the compact table is ascending by read: one pass gives every read's slice
        hoff.assign((size_t)n + 1, r.n_hits);
        {
            uint64_t rn = 0;
            for (uint64_t k = 0; k < r.n_hits; k++) while (rn <= r.hits[k].read) hoff[(size_t)rn++] = k;
        }
        for (uint32_t i = 0; i < n; i++) {
            const int32_t rs = r.read_status[i];
            if (rs == 1) continue;                                          // sga returned []: nothing is cast (:77)
            if (rs < 0) { st.crashes++; continue; }
            st.with_clusters++;
            const uint64_t a = hoff[i], b = hoff[(size_t)i + 1];
            if (a == b) { st.no_cigar++; continue; }
            uint32_t pick = (uint32_t)a;
            if (b - a > 1) {
                sc.resize((size_t)(b - a));
                for (uint64_t k = a; k < b; k++) sc[(size_t)(k - a)] = r.hits[k].score;
                pick = (uint32_t)(a + (uint64_t)bwt_otp_pick(sc.data(), (uint32_t)(b - a)));
            }
            st.cigars++;
            if (r.hits[pick].score >= threshold) { sel[i] = pick; nsel++; }
        }
        if (nsel && out) {
            // RefSeq of the printed hits in one device gather, in array order
            gwoff.resize((size_t)nsel); gwlen.resize((size_t)nsel); goff.resize((size_t)nsel + 1);
            std::vector<uint32_t> gidx(n, 0);
            uint64_t k = 0, wbytes = 0;
            // window and CIGAR of a picked hit: from its extra record if it has one (the extras are ascending by hit, and so are
            // the picked hits in array order), else the plain form
            xof.assign((size_t)nsel, ~0u);
            uint64_t xi = 0;
            const int32_t shift = (int32_t)meta[4];
            for (uint32_t i = 0; i < n; i++) if (sel[i] != ~0u) {
                const uint32_t hsel = sel[i];
                while (xi < r.n_extra && r.extra[xi].hit < hsel) xi++;
                const bool ex = xi < r.n_extra && r.extra[xi].hit == hsel;
                const uint32_t lq = r.read_len ? r.read_len : (uint32_t)(h.off[i + 1] - h.off[i]);
                const int32_t wl = (int32_t)(ex ? r.extra[xi].win_len : lq);
                if (ex) xof[(size_t)k] = (uint32_t)xi;
                gidx[i] = (uint32_t)k; gwoff[(size_t)k] = r.hits[hsel].pos - shift + 200; gwlen[(size_t)k] = wl; wbytes += (uint64_t)wl; k++;
            }
            gwin.resize((size_t)wbytes + 16);
            uint64_t need = 0;
            if ((rc2 = bwt_ref1_gather(ix, gwoff.data(), gwlen.data(), nsel, gwin.data(), wbytes, goff.data(), &need))) return rc2;
            // the lines, in the reference's cast order
            text.clear();
            const uint32_t nlists = (n + LIST - 1) / LIST;
            for (uint32_t w0 = 0; w0 < nlists; w0 += LISTS) {
                const uint32_t w1 = std::min(nlists, w0 + LISTS);
                for (uint32_t li = w1; li-- > w0;) {
                    const uint32_t f1 = std::min(n, (li + 1) * LIST);
                    for (uint32_t f = li * LIST; f < f1; f++) {
                        const uint32_t i = n - 1 - f;
                        if (sel[i] == ~0u) continue;
                        const uint32_t hsel = sel[i];
                        const uint64_t g = gidx[i], wl = goff[(size_t)g + 1] - goff[(size_t)g];
                        char plain[16];
                        const char *ctext; uint64_t cl;
                        if (xof[(size_t)g] != ~0u) { ctext = r.cigar + r.extra[xof[(size_t)g]].cig_off; cl = r.extra[xof[(size_t)g]].cig_len; }
                        else { char *e = put_int(plain, (long long)wl); *e++ = 'M'; ctext = plain; cl = (uint64_t)(e - plain); }
                        const uint64_t nl = h.name_off[i + 1] - h.name_off[i];
                        const size_t at = text.size();
                        text.resize(at + nl + clen + cl + wl + 96);
                        char *p = text.data() + at;
                        *p++ = '\n';
                        memcpy(p, h.names.data() + h.name_off[i], nl); p += nl;
                        memcpy(p, "      ", 6); p += 6; memcpy(p, chromosome, clen); p += clen;
                        memcpy(p, "      ", 6); p += 6; p = put_int(p, r.hits[hsel].pos);
                        memcpy(p, "      ", 6); p += 6; memcpy(p, ctext, cl); p += cl;
                        memcpy(p, "      ", 6); p += 6; p = put_int(p, r.hits[hsel].score);
                        memcpy(p, "      ", 6); p += 6; memcpy(p, gwin.data() + goff[(size_t)g], wl); p += wl;
                        *p++ = '\n';
                        text.resize((size_t)(p - text.data()));
                        st.lines++;
                    }
                }
            }
            if (fwrite(text.data(), 1, text.size(), out) != text.size()) return fail(BWT_EIO, "bwt_run_fastq: short write");
        } else
            st.lines += nsel;
        inflight--;
        return bwt_pipe_release(pipe, r.slot);
    };

    while (!eof || inflight) {
        if (!eof && inflight < depth) {
            uint8_t *pb; uint64_t *po; uint32_t slot;
            bwt_pipe_input in;
            rc = bwt_pipe_acquire_packed(pipe, bases_cap, batch_reads, batch_reads / 8 + 4096, &in, &slot);
            if (rc == BWT_ENOSPC) { if ((rc = consume())) return rc; continue; }
            if (rc) return rc;
            pb = in.ascii; po = in.off;
            SlotHost &h = host[slot];
            h.names.resize((size_t)batch_reads * 48 + 4096);
            h.name_off.resize((size_t)batch_reads + 1);
            uint32_t k = 0; int state = 0; uint64_t need[2] = {0, 0};
            const auto tp0 = now();
            rc = bwt_fastq_read_seqs(fq, batch_reads, pb, bases_cap, po, h.names.data(), h.names.size(), h.name_off.data(), &k, &state, need);
            t_parse += secs(tp0, now());
            if (rc == BWT_ENOSPC) {                                         // longer reads / names than assumed: grow and take the slot again
                bases_cap = std::max<uint64_t>(bases_cap, need[0] + 4096);
                h.names.resize((size_t)need[1] + 4096);
                (void)bwt_pipe_release(pipe, slot);
                if ((rc = bwt_pipe_acquire_packed(pipe, bases_cap, batch_reads, batch_reads / 8 + 4096, &in, &slot))) return rc;
                pb = in.ascii; po = in.off;
                SlotHost &h2 = host[slot];
                h2.names.resize((size_t)need[1] + 4096); h2.name_off.resize((size_t)batch_reads + 1);
                rc = bwt_fastq_read_seqs(fq, batch_reads, pb, bases_cap, po, h2.names.data(), h2.names.size(), h2.name_off.data(), &k, &state, need);
            }
            if (rc) { (void)bwt_pipe_release(pipe, slot); return rc; }
            if (state != 0) eof = true;                                     // eof, with or without a last partial batch
            if (k == 0) { (void)bwt_pipe_release(pipe, slot); continue; }
            host[slot].n = k; host[slot].off = po;
            st.reads += k;
            if ((rc = bwt_pipe_submit_ascii(pipe, slot, k, 0))) return rc;
            inflight++;
        } else if ((rc = consume()))
            return rc;
    }
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("BWT_RUN_DEBUG"))
        fprintf(stderr, "[libbwtgpu] bwt_run_fastq: %.3f s = set-up %.3f + reader %.3f + waiting for the GPU %.3f + pick / sink %.3f + rest\n", st.seconds, t_setup,
                t_parse, t_wait, t_pick);
    if (stats) *stats = st;
    return 0;
}
