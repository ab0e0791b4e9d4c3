// Shared host-side declarations of libbwtgpu's translation units (bwt_gpu.hip, index_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <shared_mutex>
#include <string>
#include "kernels_index.h"
#include "../../include/bwt_gpu.h"

int bwt_fail(int code, const std::string &msg);     // records the thread-local error text, returns code
static inline int fail(int code, const std::string &msg) { return bwt_fail(code, msg); }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return fail(_e == hipErrorOutOfMemory ? BWT_ENOMEM : BWT_EHIP, _b);                   \
        }                                                                                         \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();      // the failed attempt with slack is not an error of the batch: clear the sticky code
            e = hipMalloc(&p, bytes);
            want = bytes;
            if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return fail(BWT_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
        }
        cap = want;
        return 0;
    }
    int exact(size_t bytes) {           // no slack: for the big index arrays
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return fail(BWT_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

struct bwt_index {
    int device = 0;
    DevIndex d{};
    DevBuf blocks, sa, ref1, refplanes, ktab, ktw, ktsum;
    uint32_t Pc = 0, Pg = 0, Pt = 0, Last = 0;
    int32_t shift = 0;
    uint64_t ref1_len = 0;
    uint64_t ktw_bytes = 0;           // the table's 64-byte levels (the top one + the widened lower ones)
    bwt_options opt;                  // defaults of the batches made on this index (bwt_index_set_options); -1 = library's choice
    bwt_index() { bwt_options_init(&opt); }
    hipStream_t stream = nullptr;     // index-level work only (table build, RefSeq gathers); every batch has its own stream
    // Threading contract (include/bwt_gpu.h): any number of threads may run *_batch calls / their own bwt_batch on one
    // index at a time (shared lock while kernels read the index); bwt_index_set_ktable swaps device arrays under the
    // exclusive lock.
    mutable std::shared_mutex mu;
};

// builds the Ref1 bit-planes of an index whose ref1 buffer is resident (used by both constructors)
int bwt_index_make_refplanes(bwt_index *ix, hipStream_t stream);

// bwt_batch_set_reads_packed with a hook: `uploaded(arg)` is called once, as soon as the host buffers have crossed the link
// (before the unpack kernels are waited for), or not at all when the call fails before that point (bwt_gpu.hip; used by bwt_pipe)
int bwt_batch_set_reads_packed_hook(bwt_batch *b, const uint32_t *codes, uint64_t total, uint32_t n, uint32_t read_len,
                                    const uint64_t *off, const uint64_t *exc, uint64_t n_exc, void (*uploaded)(void *), void *arg);
