// libbwtgpu: bwt_pipe -- the host-buffer boundary as a pipeline (include/bwt_gpu.h).
//
// bwt_align_batch is one synchronous call: pageable host buffers in, host tables out, nothing overlaps (62 M reads/s at
// GRCh38 scale against 1.3 G reads/s with the reads resident).  A pipe keeps `depth` slots, each with its own resident
// batch (own HIP stream), PINNED host buffers for the reads and for the hit tables, and a host thread that runs
//     H2D of the slot's reads  ->  seed search + clustering + extension  ->  D2H of the hit tables
// so that the upload of batch k+1, the kernels of batch k and the download of batch k-1 proceed together (separate
// streams, both DMA directions, the kernels).  The caller fills the pinned input buffers directly (bwt_pipe_acquire
// hands them out: a FASTQ parser writes there, no extra host copy), submits, and takes the results in submission order;
// result pointers stay valid until the slot is released.
#include "common.h"

#include <condition_variable>
#include <deque>
#include <mutex>
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

struct Slot {
    bwt_batch *batch = nullptr;
    PinBuf bases, off, rstat, hoff, rec, woff, wlen, cigoff, cig;
    SlotState state = FREE;
    uint32_t n = 0;
    int rc = 0;
    std::string err;
    bwt_batch_counts counts{};
    uint64_t n_hits = 0, cig_bytes = 0;
    std::thread th;
};
}  // namespace

struct bwt_pipe {
    bwt_index *ix = nullptr;
    std::vector<Slot> slots;
    std::deque<uint32_t> order;          // submitted, not yet handed to the caller (submission order)
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    // one upload and one download at a time: slots that copy side by side share the link and stay in lock step (all upload,
    // all compute, all download -- no overlap between the phases); taking turns staggers them into a real pipeline
    std::mutex h2d_turn, d2h_turn;
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
        int rc;
        {
            std::lock_guard<std::mutex> turn(p->h2d_turn);
            rc = bwt_batch_set_reads(s.batch, (const uint8_t *)s.bases.p, (const uint64_t *)s.off.p, s.n);
        }
        if (!rc) rc = bwt_batch_run(s.batch, BWT_STAGE_EXTEND, &s.counts);
        uint64_t need[2] = {0, 0};
        if (!rc) {
            // sizes first (no table copied), then the tables into the slot's pinned buffers
            rc = bwt_batch_fetch_hits_raw(s.batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, need);
            if (rc == BWT_ENOSPC || rc == 0) {
                const size_t nh = (size_t)need[0];
                rc = 0;
                if ((rc = s.rstat.ensure(((size_t)s.n + 1) * 4)) || (rc = s.hoff.ensure(((size_t)s.n + 1) * 8)) ||
                    (rc = s.rec.ensure((nh + 1) * 16)) || (rc = s.woff.ensure((nh + 1) * 8)) || (rc = s.wlen.ensure((nh + 1) * 4)) ||
                    (rc = s.cigoff.ensure((nh + 2) * 8)) || (rc = s.cig.ensure((size_t)need[1] + 16))) {
                } else {
                    std::lock_guard<std::mutex> turn(p->d2h_turn);
                    rc = bwt_batch_fetch_hits_raw(s.batch, (int32_t *)s.rstat.p, (uint64_t *)s.hoff.p, (bwt_hit_rec *)s.rec.p,
                                                  (int64_t *)s.woff.p, (int32_t *)s.wlen.p, (uint64_t *)s.cigoff.p, (char *)s.cig.p,
                                                  need[0], need[1], need);
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            s.rc = rc;
            s.err = rc ? bwt_last_error() : "";
            s.n_hits = need[0]; s.cig_bytes = need[1];
            s.state = DONE;
        }
        p->cv.notify_all();
    }
}

extern "C" int bwt_pipe_create(bwt_index *ix, uint32_t depth, bwt_pipe **out) {
    if (!ix || !out) return fail(BWT_EINVAL, "bwt_pipe_create: null argument");
    if (depth < 1 || depth > 8) return fail(BWT_EINVAL, "bwt_pipe_create: depth must be 1..8");
    HIPCHK(hipSetDevice(ix->device));
    bwt_pipe *p = new bwt_pipe();
    p->ix = ix;
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
        PinBuf *all[] = {&s.bases, &s.off, &s.rstat, &s.hoff, &s.rec, &s.woff, &s.wlen, &s.cigoff, &s.cig};
        for (auto b : all) b->release();
    }
    delete p;
}

extern "C" int bwt_pipe_acquire(bwt_pipe *p, uint64_t bases_cap, uint32_t reads_cap, uint8_t **bases, uint64_t **off, uint32_t *slot) {
    if (!p || !bases || !off || !slot) return fail(BWT_EINVAL, "bwt_pipe_acquire: null argument");
    uint32_t si = 0;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        // a FREE slot; if there is none but batches are still running or waiting to be taken, that is the caller's turn
        // to call bwt_pipe_wait / bwt_pipe_release -- blocking here would deadlock a single-threaded caller
        bool found = false;
        for (uint32_t i = 0; i < p->slots.size(); i++) if (p->slots[i].state == FREE) { si = i; found = true; break; }
        if (!found) return fail(BWT_ENOSPC, "bwt_pipe_acquire: every slot is in use (wait for / release a result first)");
        p->slots[si].state = FILLING;
    }
    Slot &s = p->slots[si];
    (void)hipSetDevice(p->ix->device);
    int rc;
    if ((rc = s.bases.ensure((size_t)bases_cap + 16)) || (rc = s.off.ensure(((size_t)reads_cap + 1) * 8))) {
        std::lock_guard<std::mutex> lk(p->mu);
        s.state = FREE;
        return rc;
    }
    *bases = (uint8_t *)s.bases.p; *off = (uint64_t *)s.off.p; *slot = si;
    return 0;
}

extern "C" int bwt_pipe_submit(bwt_pipe *p, uint32_t slot, uint32_t n_reads) {
    if (!p || slot >= p->slots.size()) return fail(BWT_EINVAL, "bwt_pipe_submit: no such slot");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        Slot &s = p->slots[slot];
        if (s.state != FILLING) return fail(BWT_EINVAL, "bwt_pipe_submit: slot was not acquired");
        if ((size_t)(n_reads + 1ull) * 8 > s.off.cap) return fail(BWT_EINVAL, "bwt_pipe_submit: more reads than the slot was acquired for");
        s.n = n_reads;
        s.state = QUEUED;
        p->order.push_back(slot);
    }
    p->cv.notify_all();
    return 0;
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
    res->slot = si; res->n_reads = s.n; res->status = s.rc;
    res->n_hits = s.n_hits; res->cigar_bytes = s.cig_bytes;
    res->read_status = (const int32_t *)s.rstat.p; res->hit_off = (const uint64_t *)s.hoff.p;
    res->hits = (const bwt_hit_rec *)s.rec.p; res->win_off = (const int64_t *)s.woff.p; res->win_len = (const int32_t *)s.wlen.p;
    res->cig_off = (const uint64_t *)s.cigoff.p; res->cigar = (const char *)s.cig.p;
    res->counts = s.counts;
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
