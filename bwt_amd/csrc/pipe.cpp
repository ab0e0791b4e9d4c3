// libbwtgpu: bwt_pipe -- the host-buffer boundary as a pipeline (include/bwt_gpu.h).
//
// bwt_align_batch is one synchronous call: pageable host buffers in, host tables out, nothing overlaps (0.07 G reads/s at
// GRCh38 scale against 2 G reads/s with the reads resident).  A pipe keeps `depth` slots, each with its own resident
// batch (own HIP stream), PINNED host buffers for the reads and for the hit tables, and a host thread that runs
//     H2D of the slot's reads  ->  seed search + clustering + extension  ->  D2H of the hit tables
// so that the upload of batch k+1, the kernels of batch k and the download of batch k-1 proceed together (separate
// streams, both DMA directions, the kernels).  What crosses the link is the packed form of the reads (2 bits per base +
// exceptions: 25 bytes per 100 bp read; round 2 sent 108) and the compact hit table (16 bytes per hit + 1 per read; round 2
// sent 52 per read), see include/bwt_gpu.h.  The caller fills the pinned input buffers directly (bwt_pipe_acquire_packed hands
// them out: a FASTQ parser / NIF packs there), submits, and takes the results in submission order; result pointers stay valid
// until the slot is released.
#include "common.h"

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(BWT_ENOMEM, "hipHostMalloc of " + std::to_string(want) + " bytes failed");
        }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

enum SlotState { FREE = 0, FILLING, QUEUED, DONE, TAKEN };
enum SlotMode { M_ASCII = 0, M_PACKED, M_PACK_ASCII };

struct Slot {
    bwt_batch *batch = nullptr;
    PinBuf bases, off, codes, exc;                 // inputs (pinned): ASCII form / packed form
    uint8_t *ascii = nullptr; size_t ascii_cap = 0; // pageable staging for bwt_pipe_submit_ascii (malloc: not zero-filled)
    PinBuf rstat, rec, extra, cig, nwoff, nscore;  // results (pinned): the compact hit table (+ the narrow arrays of its narrow form)
    SlotState state = FREE;
    SlotMode mode = M_ASCII;
    uint32_t n = 0, read_len = 0, reads_cap = 0;
    uint64_t total = 0, n_exc = 0, bases_cap = 0, exc_cap = 0;
    int rc = 0;
    std::string err;
    bwt_batch_counts counts{};
    uint64_t n_hits = 0, n_extra = 0, cig_bytes = 0, n_narrow = 0;
    double ms[4] = {0, 0, 0, 0};                   // upload, kernels, download, waiting for a turn
    uint64_t seq = 0;                              // submission number: the order of the turns below
    std::thread th;
};

// A stage of the pipeline that one batch at a time passes through, IN SUBMISSION ORDER (a plain mutex lets a later batch overtake
// an earlier one whose results the caller is waiting for).  Every submitted batch enters and leaves every stage exactly once,
// failed or not.
struct Turn {
    std::mutex m;
    std::condition_variable cv;
    uint64_t now = 0;
    void enter(uint64_t seq) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return now == seq; }); }
    void leave() { { std::lock_guard<std::mutex> lk(m); now++; } cv.notify_all(); }
};
struct InTurn {                                    // scope of a batch's turn (or up to leave_now)
    Turn &t;
    bool in = true;
    InTurn(Turn &turn, uint64_t seq) : t(turn) { t.enter(seq); }
    void leave_now() { if (in) { in = false; t.leave(); } }
    ~InTurn() { leave_now(); }
};
}  // namespace

struct bwt_pipe {
    bwt_index *ix = nullptr;
    std::vector<Slot> slots;
    std::deque<uint32_t> order;          // submitted, not yet handed to the caller (submission order)
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    // one upload, one batch's kernels and one download at a time, in submission order: slots that copy side by side share the link
    // and stay in lock step (all upload, all compute, all download -- no overlap between the phases); taking turns staggers them
    // into a real pipeline.  The kernels of a batch fill the chip by themselves (persistent grids): two batches' kernels side by side gain
    // nothing, and taking turns keeps the per-kernel event times those of undisturbed launches.
    Turn h2d_turn, gpu_turn, d2h_turn;
    uint64_t next_seq = 0;
    bool debug = false;                  // BWT_PIPE_DEBUG: a timeline line per event on stderr
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void trace(uint32_t slot, const char *what) const {
        if (debug) fprintf(stderr, "[pipe %9.3f ms] slot %u %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), slot, what);
    }
};

static void slot_worker(bwt_pipe *p, uint32_t si) {
    Slot &s = p->slots[si];
    (void)hipSetDevice(p->ix->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || s.state == QUEUED; });
            if (p->stop) return;
        }
        int rc = 0;
        if (s.mode == M_PACK_ASCII) {
            // the caller left ASCII in the staging buffer: pack it into the pinned buffers here, off the caller's thread
            for (int attempt = 0; attempt < 2; attempt++) {
                uint64_t k = 0;
                rc = bwt_pack_reads(s.ascii, s.total, (uint32_t *)s.codes.p, (uint64_t *)s.exc.p, s.exc.cap / 8, &k, 0);
                s.n_exc = k;
                if (rc != BWT_ENOSPC) break;
                if ((rc = s.exc.ensure((size_t)k * 8 + 64))) break;
                rc = BWT_ENOSPC;
            }
        }
        p->trace(si, "picked up");
        using clk = std::chrono::steady_clock;
        auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
        s.ms[0] = s.ms[1] = s.ms[2] = s.ms[3] = 0;
        {
            const auto tw = clk::now();
            InTurn turn(p->h2d_turn, s.seq);
            s.ms[3] += ms_since(tw);
            if (!rc) {
                p->trace(si, "upload begins");
                const auto t0 = clk::now();
                if (s.mode == M_ASCII) {
                    rc = bwt_batch_set_reads(s.batch, (const uint8_t *)s.bases.p, (const uint64_t *)s.off.p, s.n);
                    s.ms[0] = ms_since(t0);
                } else {
                    // the link is the next batch's as soon as this one's buffers are across: the unpack kernels run beside its upload
                    struct Hook { bwt_pipe *p; uint32_t si; InTurn *turn; double *ms; clk::time_point t0; } hk{p, si, &turn, &s.ms[0], t0};
                    rc = bwt_batch_set_reads_packed_hook(s.batch, (const uint32_t *)s.codes.p, s.total, s.n, s.read_len,
                                                         s.read_len ? nullptr : (const uint64_t *)s.off.p, (const uint64_t *)s.exc.p, s.n_exc,
                                                         [](void *a) {
                                                             Hook *h = (Hook *)a;
                                                             *h->ms = std::chrono::duration<double, std::milli>(clk::now() - h->t0).count();
                                                             h->p->trace(h->si, "upload across");
                                                             h->turn->leave_now();
                                                         }, &hk);
                }
            }
        }
        {
            const auto tw = clk::now();
            InTurn turn(p->gpu_turn, s.seq);
            s.ms[3] += ms_since(tw);
            if (!rc) {
                p->trace(si, "kernels begin");
                const auto t0 = clk::now();
                rc = bwt_batch_run(s.batch, BWT_STAGE_EXTEND, &s.counts);
                s.ms[1] = ms_since(t0);
                p->trace(si, "kernels done");
            }
        }
        // (the table as the device made it: its narrow form when the slot's batch runs with bwt_options.hits_narrow, else no narrow
        // reads at all and this is the compact table)
        uint64_t need[4] = {0, 0, 0, 0};                   // narrow, hits, extras, text
        if (!rc) {
            // sizes first (no table copied); the pinned result buffers grow outside the download's turn
            rc = bwt_batch_fetch_hits_narrow(s.batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, need);
            if (rc == BWT_ENOSPC) rc = 0;
            if (!rc && ((rc = s.rstat.ensure((size_t)s.n + 16)) || (rc = s.rec.ensure(((size_t)need[1] + 1) * 16)) ||
                        (rc = s.extra.ensure(((size_t)need[2] + 1) * 16)) || (rc = s.cig.ensure((size_t)need[3] + 16)) ||
                        (rc = s.nwoff.ensure(((size_t)need[0] + 4) * 4)) || (rc = s.nscore.ensure(((size_t)need[0] + 8) * 2)))) {
            }
        }
        {
            const auto tw = clk::now();
            InTurn turn(p->d2h_turn, s.seq);
            s.ms[3] += ms_since(tw);
            if (!rc) {
                // the compact table into the slot's pinned buffers
                p->trace(si, "download begins");
                const auto t0 = clk::now();
                struct T { double &d; clk::time_point a; ~T() { d = std::chrono::duration<double, std::milli>(clk::now() - a).count(); } } tt{s.ms[2], t0};
                rc = bwt_batch_fetch_hits_narrow(s.batch, (int8_t *)s.rstat.p, (uint32_t *)s.nwoff.p, (int16_t *)s.nscore.p, (bwt_hit_rec *)s.rec.p,
                                                 (bwt_hit_extra *)s.extra.p, (char *)s.cig.p, need[0], need[1], need[2], need[3], need);
            }
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            s.rc = rc;
            s.err = rc ? bwt_last_error() : "";
            s.n_narrow = need[0]; s.n_hits = need[1]; s.n_extra = need[2]; s.cig_bytes = need[3];
            s.state = DONE;
        }
        p->trace(si, "done");
        p->cv.notify_all();
    }
}

extern "C" int bwt_pipe_create(bwt_index *ix, uint32_t depth, bwt_pipe **out) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_pipe_create: null argument");
    if (depth < 1 || depth > 8) return fail(BWT_EINVAL, "bwt_pipe_create: depth must be 1..8");
    HIPCHK(hipSetDevice(ix->device));
    bwt_pipe *p = new bwt_pipe();
    p->ix = ix;
    p->debug = getenv("BWT_PIPE_DEBUG") != nullptr;
    p->slots.resize(depth);
    for (uint32_t i = 0; i < depth; i++) {
        int rc = bwt_batch_create(ix, &p->slots[i].batch);
        if (rc) { bwt_pipe_destroy(p); return rc; }
    }
    for (uint32_t i = 0; i < depth; i++) p->slots[i].th = std::thread(slot_worker, p, i);
    *out = p;
    return 0;
}

extern "C" void bwt_pipe_destroy(bwt_pipe *p) {
    if (!p) return;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        // let running batches finish: a worker only looks at `stop` between batches
        p->cv.wait(lk, [&] { for (auto &s : p->slots) if (s.state == QUEUED) return false; return true; });
        p->stop = true;
    }
    p->cv.notify_all();
    (void)hipSetDevice(p->ix->device);
    for (auto &s : p->slots) {
        if (s.th.joinable()) s.th.join();
        if (s.batch) bwt_batch_destroy(s.batch);
        PinBuf *all[] = {&s.bases, &s.off, &s.codes, &s.exc, &s.rstat, &s.rec, &s.extra, &s.cig, &s.nwoff, &s.nscore};
        for (auto b : all) b->release();
        free(s.ascii);
    }
    delete p;
}

// the pipe's slots must be idle for these (no batch submitted and not yet taken)
static bool pipe_idle(bwt_pipe *p) {
    std::lock_guard<std::mutex> lk(p->mu);
    for (auto &s : p->slots) if (s.state == QUEUED) return false;
    return true;
}

extern "C" int bwt_pipe_set_options(bwt_pipe *p, const bwt_options *opt) {
    if (!p || !opt) return fail(BWT_EINVAL, "bwt_pipe_set_options: null argument");
    if (!pipe_idle(p)) return fail(BWT_EINVAL, "bwt_pipe_set_options: batches are running");
    for (auto &s : p->slots) { int rc = bwt_batch_set_options(s.batch, opt); if (rc) return rc; }
    return 0;
}

extern "C" int bwt_pipe_profile(bwt_pipe *p, int enable) {
    if (!p) return fail(BWT_EINVAL, "bwt_pipe_profile: null pipe");
    if (!pipe_idle(p)) return fail(BWT_EINVAL, "bwt_pipe_profile: batches are running");
    for (auto &s : p->slots) { int rc = bwt_batch_profile(s.batch, enable); if (rc) return rc; }
    return 0;
}

extern "C" int bwt_pipe_kernel_times(bwt_pipe *p, char *names, uint32_t names_cap, double *ms, uint64_t *launches, uint32_t cap) {
    if (!p || !names || !ms || !launches) return fail(BWT_EINVAL, "bwt_pipe_kernel_times: null argument");
    if (!pipe_idle(p)) return fail(BWT_EINVAL, "bwt_pipe_kernel_times: batches are running");
    std::vector<std::string> nm;
    std::vector<double> tm;
    std::vector<uint64_t> ln;
    for (auto &s : p->slots) {
        char buf[4096];
        double m[64];
        uint64_t l[64];
        const int k = bwt_batch_kernel_times(s.batch, buf, sizeof buf, m, l, 64);
        if (k < 0) return k;
        const char *q = buf;
        for (int i = 0; i < k; i++) {
            const std::string name(q);
            q += name.size() + 1;
            size_t j = 0;
            while (j < nm.size() && nm[j] != name) j++;
            if (j == nm.size()) { nm.push_back(name); tm.push_back(0); ln.push_back(0); }
            tm[j] += m[i]; ln[j] += l[i];
        }
    }
    uint32_t k = 0, np = 0;
    for (size_t j = 0; j < nm.size(); j++) {
        if (k >= cap || np + nm[j].size() + 1 > names_cap) break;
        memcpy(names + np, nm[j].c_str(), nm[j].size() + 1);
        np += (uint32_t)nm[j].size() + 1;
        ms[k] = tm[j]; launches[k] = ln[j];
        k++;
    }
    return (int)k;
}

static int take_free_slot(bwt_pipe *p, const char *who, uint32_t &si) {
    std::unique_lock<std::mutex> lk(p->mu);
    // a FREE slot; if there is none but batches are still running or waiting to be taken, that is the caller's turn
    // to call bwt_pipe_wait / bwt_pipe_release -- blocking here would deadlock a single-threaded caller
    for (uint32_t i = 0; i < p->slots.size(); i++)
        if (p->slots[i].state == FREE) { si = i; p->slots[i].state = FILLING; return 0; }
    return fail(BWT_ENOSPC, std::string(who) + ": every slot is in use (wait for / release a result first)");
}

extern "C" int bwt_pipe_acquire(bwt_pipe *p, uint64_t bases_cap, uint32_t reads_cap, uint8_t **bases, uint64_t **off, uint32_t *slot) {
    if (!p || !bases || !off || !slot) return fail(BWT_EINVAL, "bwt_pipe_acquire: null argument");
    uint32_t si = 0;
    int rc = take_free_slot(p, "bwt_pipe_acquire", si);
    if (rc) return rc;
    Slot &s = p->slots[si];
    (void)hipSetDevice(p->ix->device);
    if ((rc = s.bases.ensure((size_t)bases_cap + 16)) || (rc = s.off.ensure(((size_t)reads_cap + 1) * 8))) {
        std::lock_guard<std::mutex> lk(p->mu);
        s.state = FREE;
        return rc;
    }
    s.mode = M_ASCII; s.bases_cap = bases_cap; s.reads_cap = reads_cap;
    *bases = (uint8_t *)s.bases.p; *off = (uint64_t *)s.off.p; *slot = si;
    return 0;
}

extern "C" int bwt_pipe_acquire_packed(bwt_pipe *p, uint64_t bases_cap, uint32_t reads_cap, uint64_t exc_cap, bwt_pipe_input *in, uint32_t *slot) {
    if (!p || !in || !slot) return fail(BWT_EINVAL, "bwt_pipe_acquire_packed: null argument");
    uint32_t si = 0;
    int rc = take_free_slot(p, "bwt_pipe_acquire_packed", si);
    if (rc) return rc;
    Slot &s = p->slots[si];
    (void)hipSetDevice(p->ix->device);
    if ((rc = s.codes.ensure((size_t)bwt_packed_words(bases_cap) * 4 + 16)) || (rc = s.exc.ensure((size_t)exc_cap * 8 + 64)) ||
        (rc = s.off.ensure(((size_t)reads_cap + 1) * 8))) {
        std::lock_guard<std::mutex> lk(p->mu);
        s.state = FREE;
        return rc;
    }
    if (s.ascii_cap < bases_cap + 16) {
        free(s.ascii);
        s.ascii_cap = (size_t)bases_cap + bases_cap / 4 + 4096;
        s.ascii = (uint8_t *)malloc(s.ascii_cap);
        if (!s.ascii) { s.ascii_cap = 0; std::lock_guard<std::mutex> lk(p->mu); s.state = FREE; return fail(BWT_ENOMEM, "bwt_pipe_acquire_packed: no memory for the ASCII staging buffer"); }
    }
    s.mode = M_PACKED; s.bases_cap = bases_cap; s.reads_cap = reads_cap; s.exc_cap = s.exc.cap / 8;
    in->codes = (uint32_t *)s.codes.p; in->exc = (uint64_t *)s.exc.p; in->off = (uint64_t *)s.off.p; in->ascii = s.ascii;
    in->bases_cap = bases_cap; in->exc_cap = s.exc_cap; in->reads_cap = reads_cap;
    *slot = si;
    return 0;
}

static int queue_slot(bwt_pipe *p, uint32_t slot) {
    p->trace(slot, "submitted");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->slots[slot].seq = p->next_seq++;
        p->slots[slot].state = QUEUED;
        p->order.push_back(slot);
    }
    p->cv.notify_all();
    return 0;
}

extern "C" int bwt_pipe_submit(bwt_pipe *p, uint32_t slot, uint32_t n_reads) {
    if (!p || slot >= p->slots.size()) return fail(BWT_EINVAL, "bwt_pipe_submit: no such slot");
    Slot &s = p->slots[slot];
    if (s.state != FILLING || s.mode != M_ASCII) return fail(BWT_EINVAL, "bwt_pipe_submit: slot was not acquired with bwt_pipe_acquire");
    if (n_reads > s.reads_cap) return fail(BWT_EINVAL, "bwt_pipe_submit: more reads than the slot was acquired for");
    const uint64_t *off = (const uint64_t *)s.off.p;
    if (off[0] != 0 || off[n_reads] > s.bases_cap) return fail(BWT_EINVAL, "bwt_pipe_submit: offsets must start at 0 and stay inside the acquired capacity");
    s.n = n_reads; s.read_len = 0;
    return queue_slot(p, slot);
}

extern "C" int bwt_pipe_submit_packed(bwt_pipe *p, uint32_t slot, uint32_t n_reads, uint32_t read_len, uint64_t total_bases, uint64_t n_exc) {
    if (!p || slot >= p->slots.size()) return fail(BWT_EINVAL, "bwt_pipe_submit_packed: no such slot");
    Slot &s = p->slots[slot];
    if (s.state != FILLING || s.mode != M_PACKED) return fail(BWT_EINVAL, "bwt_pipe_submit_packed: slot was not acquired with bwt_pipe_acquire_packed");
    if (n_reads > s.reads_cap || total_bases > s.bases_cap || n_exc > s.exc_cap) return fail(BWT_EINVAL, "bwt_pipe_submit_packed: more than the slot was acquired for");
    if (read_len ? (uint64_t)n_reads * read_len != total_bases : (((const uint64_t *)s.off.p)[0] != 0 || ((const uint64_t *)s.off.p)[n_reads] != total_bases))
        return fail(BWT_EINVAL, "bwt_pipe_submit_packed: total_bases disagrees with read_len / the offsets");
    s.n = n_reads; s.read_len = read_len; s.total = total_bases; s.n_exc = n_exc;
    return queue_slot(p, slot);
}

extern "C" int bwt_pipe_submit_ascii(bwt_pipe *p, uint32_t slot, uint32_t n_reads, uint32_t read_len) {
    if (!p || slot >= p->slots.size()) return fail(BWT_EINVAL, "bwt_pipe_submit_ascii: no such slot");
    Slot &s = p->slots[slot];
    if (s.state != FILLING || s.mode != M_PACKED) return fail(BWT_EINVAL, "bwt_pipe_submit_ascii: slot was not acquired with bwt_pipe_acquire_packed");
    if (n_reads > s.reads_cap) return fail(BWT_EINVAL, "bwt_pipe_submit_ascii: more reads than the slot was acquired for");
    const uint64_t *off = (const uint64_t *)s.off.p;
    const uint64_t total = read_len ? (uint64_t)n_reads * read_len : off[n_reads];
    if (total > s.bases_cap || (!read_len && off[0] != 0)) return fail(BWT_EINVAL, "bwt_pipe_submit_ascii: offsets must start at 0 and stay inside the acquired capacity");
    s.n = n_reads; s.read_len = read_len; s.total = total; s.n_exc = 0;
    s.mode = M_PACK_ASCII;
    return queue_slot(p, slot);
}

extern "C" int bwt_pipe_wait(bwt_pipe *p, bwt_pipe_result *res) {
    if (!p || !res) return fail(BWT_EINVAL, "bwt_pipe_wait: null argument");
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->order.empty()) return fail(BWT_EINVAL, "bwt_pipe_wait: nothing submitted");
    const uint32_t si = p->order.front();
    Slot &s = p->slots[si];
    p->cv.wait(lk, [&] { return s.state == DONE; });
    p->order.pop_front();
    s.state = TAKEN;
    p->trace(si, "taken");
    res->slot = si; res->n_reads = s.n; res->status = s.rc; res->read_len = s.read_len;
    res->n_hits = s.n_hits; res->n_extra = s.n_extra; res->cigar_bytes = s.cig_bytes;
    res->read_status = (const int8_t *)s.rstat.p; res->hits = (const bwt_hit_rec *)s.rec.p;
    res->extra = (const bwt_hit_extra *)s.extra.p; res->cigar = (const char *)s.cig.p;
    res->counts = s.counts;
    res->ms_upload = s.ms[0]; res->ms_kernels = s.ms[1]; res->ms_download = s.ms[2]; res->ms_waiting = s.ms[3];
    res->n_narrow = s.n_narrow; res->narrow_off = (const uint32_t *)s.nwoff.p; res->narrow_score = (const int16_t *)s.nscore.p;
    if (s.rc) return fail(s.rc, "bwt_pipe: batch failed: " + s.err);
    return 0;
}

extern "C" int bwt_pipe_release(bwt_pipe *p, uint32_t slot) {
    if (!p || slot >= p->slots.size()) return fail(BWT_EINVAL, "bwt_pipe_release: no such slot");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        Slot &s = p->slots[slot];
        if (s.state != TAKEN && s.state != FILLING) return fail(BWT_EINVAL, "bwt_pipe_release: slot holds no result");
        s.state = FREE;
    }
    p->cv.notify_all();
    return 0;
}

extern "C" bwt_batch *bwt_pipe_batch(bwt_pipe *p, uint32_t slot) {
    if (!p || slot >= p->slots.size()) return nullptr;
    return p->slots[slot].batch;
}
