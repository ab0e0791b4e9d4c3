"""ctypes binding of libbwtgpu.so -- the C ABI of include/bwt_gpu.h.  Fails loudly when the
library has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbwtgpu.so")

STAGE_SEARCH, STAGE_CLUSTER, STAGE_EXTEND = 1, 2, 4
ENOSPC = -28

EXPORTS = [
    "bwt_constants", "bwt_last_error", "bwt_index_open", "bwt_index_from_arrays", "bwt_index_close",
    "bwt_index_meta", "bwt_index_device_bytes", "bwt_index_set_ktable", "bwt_index_ktable", "bwt_find_seeds_batch", "bwt_sga_batch", "bwt_sw_batch",
    "bwt_align_batch", "bwt_ref1_read", "bwt_batch_create", "bwt_batch_destroy", "bwt_batch_set_reads",
    "bwt_batch_run", "bwt_batch_profile", "bwt_batch_kernel_times", "bwt_batch_hits_to_device",
    "bwt_batch_fetch_tasks", "bwt_batch_fetch_clusters", "bwt_batch_fetch_hits", "bwt_fm_check", "bwt_fm_rewrite",
    "bwt_build_index", "bwt_index_save", "bwt_index_export", "bwt_fastq_open", "bwt_fastq_close", "bwt_fastq_read_seqs", "bwt_ref1_gather",
    "bwt_seeds_generate_fs", "bwt_fs_find_seeds", "bwt_fs_find_seeds_index",
    "bwt_batch_fetch_hits_raw", "bwt_pipe_create", "bwt_pipe_destroy", "bwt_pipe_acquire", "bwt_pipe_submit", "bwt_pipe_wait",
    "bwt_pipe_release", "bwt_reverse_complement_batch", "bwt_batch_hits_to_device_global",
    "bwt_index_save_native", "bwt_index_open_native", "bwt_native_check", "bwt_fastq_set_threads",
    "bwt_run_fastq", "bwt_otp_pick", "bwt_batch_set_reads_both_strands", "bwt_batch_pair_hits",
    "bwt_options_init", "bwt_index_set_options", "bwt_batch_set_options", "bwt_index_set_ktable_ex", "bwt_packed_words", "bwt_pack_reads",
    "bwt_batch_set_reads_packed", "bwt_batch_fetch_hits_compact", "bwt_hits_expand", "bwt_pipe_set_options", "bwt_pipe_profile",
    "bwt_pipe_kernel_times", "bwt_pipe_acquire_packed", "bwt_pipe_submit_packed", "bwt_pipe_submit_ascii", "bwt_pipe_batch",
    "bwt_batch_compact_to_device",
    "bwt_batch_fetch_hits_narrow", "bwt_hits_expand_narrow",
    "bwt_small_create", "bwt_small_destroy", "bwt_small_set_options", "bwt_small_align", "bwt_small_profile", "bwt_small_kernel_times",
]


class BwtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libbwtgpu error %d: %s" % (code, text))
        self.code = code


class Constants(C.Structure):
    _fields_ = [("tolerance", C.c_int32), ("min_len", C.c_int32), ("max_range", C.c_int32), ("max_len", C.c_int32),
                ("match", C.c_int32), ("mismatch", C.c_int32), ("gap_penalty", C.c_int32),
                ("gap_ext_penalty", C.c_int32), ("unknown", C.c_double), ("ref_extension_len", C.c_int32)]


class BatchCounts(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_reads", "n_tasks", "lf_steps", "sa_reads", "n_clusters", "sw_cells",
                                          "n_hits", "cigar_bytes", "window_bytes", "n_float", "n_traced", "n_dp", "n_lookups", "n_small",
                                          "n_handed_back", "n_banded", "n_certified")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class RunStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("reads", "with_clusters", "no_cigar", "cigars", "lines", "crashes")] + [("seconds", C.c_double)]


OPTION_FIELDS = ("seed_fuse", "seed_fast", "seed_big", "search_ilp", "sw_prefilter", "sw_fuse_cand", "sw_wave", "sw_band", "sw_band_min",
                 "sw_lds", "sw_pk", "sw_trace_all", "pf_blocks_per_cu", "sw_blocks_per_cu", "sw_waves_per_cu", "debug", "hits_narrow", "sw_band12", "up_zero_copy", "small_graph", "sw_cert", "pf_kcap")


class Options(C.Structure):
    """bwt_options: which of the library's equivalent kernels run (-1 = its own choice); never read from the environment by the library."""
    _fields_ = [("struct_size", C.c_uint32)] + [(k, C.c_int32) for k in OPTION_FIELDS]


def make_options(**kw):
    o = Options()
    load().bwt_options_init(C.byref(o))
    for k, v in kw.items():
        if k not in OPTION_FIELDS:
            raise KeyError("no such option: %s" % k)
        setattr(o, k, int(v))
    return o


def options_from_env(env=None):
    """BWT_SEED_FAST=0 etc. -> keyword arguments for set_options (bench.py / scripts / tests: experiments are switched from outside
    the library, which itself never looks at the environment)."""
    env = os.environ if env is None else env
    out = {}
    for k in OPTION_FIELDS:
        v = env.get("BWT_" + k.upper())
        if v not in (None, ""):
            out[k] = int(v)
    return out


HIT_REC = np.dtype([("read", np.uint32), ("score", np.int32), ("pos", np.int64)])
PAIR_REC = np.dtype([("hit1", np.uint32), ("hit2", np.uint32), ("insert", np.int32), ("score", np.int32), ("n_proper", np.uint32),
                     ("flags", np.uint32)])
NO_HIT = 0xFFFFFFFF


HIT_EXTRA = np.dtype([("hit", np.uint32), ("win_len", np.uint32), ("cig_off", np.uint32), ("cig_len", np.uint32)])


class PipeInput(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("exc", C.c_void_p), ("off", C.c_void_p), ("ascii", C.c_void_p), ("bases_cap", C.c_uint64),
                ("exc_cap", C.c_uint64), ("reads_cap", C.c_uint32)]


class PipeResult(C.Structure):
    _fields_ = [("slot", C.c_uint32), ("n_reads", C.c_uint32), ("status", C.c_int32), ("read_len", C.c_uint32), ("n_hits", C.c_uint64),
                ("n_extra", C.c_uint64), ("cigar_bytes", C.c_uint64), ("read_status", C.c_void_p), ("hits", C.c_void_p),
                ("extra", C.c_void_p), ("cigar", C.c_void_p), ("counts", BatchCounts), ("ms_upload", C.c_double), ("ms_kernels", C.c_double),
                ("ms_download", C.c_double), ("ms_waiting", C.c_double), ("n_narrow", C.c_uint64), ("narrow_off", C.c_void_p), ("narrow_score", C.c_void_p)]


class SmallResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("path", C.c_uint32), ("n_hits", C.c_uint64), ("n_extra", C.c_uint64), ("cigar_bytes", C.c_uint64),
                ("read_status", C.c_void_p), ("hits", C.c_void_p), ("extra", C.c_void_p), ("cigar", C.c_void_p),
                ("n_candidates", C.c_uint64), ("n_dp", C.c_uint64), ("n_float", C.c_uint64), ("overflow_flags", C.c_uint64),
                ("us_copy_in", C.c_double), ("us_enqueue", C.c_double), ("us_wait", C.c_double), ("us_total", C.c_double),
                ("graph", C.c_uint32), ("reserved_", C.c_uint32)]


SMALL_MAX_READS = 4096

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libbwtgpu.so is not built (%s missing): run __graft_entry__.build(); "
                          "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the header and the library ever diverge
        getattr(lib, name).restype = C.c_int
    lib.bwt_last_error.restype = C.c_char_p
    lib.bwt_index_close.restype = None
    lib.bwt_fastq_close.restype = None
    lib.bwt_batch_destroy.restype = None
    lib.bwt_pipe_destroy.restype = None
    lib.bwt_small_destroy.restype = None
    lib.bwt_constants.restype = None
    lib.bwt_options_init.restype = None
    lib.bwt_packed_words.restype = C.c_uint64
    lib.bwt_pipe_batch.restype = C.c_void_p
    _lib = lib
    return lib


def _chk(rc):
    if rc < 0:
        raise BwtError(rc, load().bwt_last_error().decode("utf8", "replace"))
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def constants():
    c = Constants()
    load().bwt_constants(C.byref(c))
    return c


def fm_check(fm_path):
    """Parse a .fm on the host exactly as bwt_index_open does -> ((Pc,Pg,Pt,Last), Shift)."""
    out = (C.c_uint32 * 5)()
    _chk(load().bwt_fm_check(os.fsencode(fm_path), out))
    return tuple(int(x) for x in out[:4]), C.c_int32(int(out[4])).value


def native_check(path):
    """Header and checksums of a native index file, on the host -> ((Pc,Pg,Pt,Last), Shift)."""
    out = (C.c_uint32 * 5)()
    _chk(load().bwt_native_check(os.fsencode(path), out))
    return tuple(int(x) for x in out[:4]), C.c_int32(int(out[4])).value


def fm_rewrite(in_path, out_path):
    _chk(load().bwt_fm_rewrite(os.fsencode(in_path), os.fsencode(out_path)))


def pack_2bit(bases, threads=0):
    """bwt_pack_reads: ASCII stream -> (codes u32[bwt_packed_words], exceptions u64[]) -- the packed wire format."""
    bases = np.ascontiguousarray(bases, np.uint8)
    total = len(bases)
    codes = np.zeros(int(load().bwt_packed_words(C.c_uint64(total))), np.uint32)
    n_exc = C.c_uint64(0)
    exc = np.zeros(max(16, total // 64), np.uint64)
    rc = load().bwt_pack_reads(_p(bases), C.c_uint64(total), _p(codes), _p(exc), C.c_uint64(len(exc)), C.byref(n_exc), C.c_uint32(threads))
    if rc == ENOSPC:
        exc = np.zeros(n_exc.value, np.uint64)
        rc = load().bwt_pack_reads(_p(bases), C.c_uint64(total), _p(codes), _p(exc), C.c_uint64(len(exc)), C.byref(n_exc), C.c_uint32(threads))
    _chk(rc)
    return codes, exc[:n_exc.value].copy()


def hits_expand(n, read_len, off, shift, hits, extra, xcigar, read_status=None, narrow_off=None, narrow_score=None):
    """bwt_hits_expand / bwt_hits_expand_narrow: a compact hit table (HIT_REC array, HIT_EXTRA array, text bytes; for its narrow form
    also the status bytes and the narrow arrays) -> the per-hit arrays of align_batch."""
    hits = np.ascontiguousarray(hits, HIT_REC); extra = np.ascontiguousarray(extra, HIT_EXTRA)
    nn = 0 if narrow_off is None else len(narrow_off)
    if nn:
        read_status = np.ascontiguousarray(read_status, np.int8); narrow_off = np.ascontiguousarray(narrow_off, np.uint32)
        narrow_score = np.ascontiguousarray(narrow_score, np.int16)
    nh = len(hits) + nn
    xc = np.frombuffer(bytes(xcigar), np.uint8) if not isinstance(xcigar, np.ndarray) else np.ascontiguousarray(xcigar, np.uint8)
    off64 = None if off is None else np.ascontiguousarray(off, np.uint64)
    hoff = np.empty(n + 1, np.uint64); score = np.empty(nh, np.int32); pos = np.empty(nh, np.int64); woff = np.empty(nh, np.int64)
    wlen = np.empty(nh, np.int32); coff = np.empty(nh + 1, np.uint64)
    need = C.c_uint64(0)

    def call(cig, cap):
        return load().bwt_hits_expand_narrow(C.c_uint32(n), C.c_uint32(read_len), _p(off64), C.c_int32(shift), _p(read_status) if nn else None,
                                             _p(narrow_off) if nn else None, _p(narrow_score) if nn else None, C.c_uint64(nn), _p(hits),
                                             C.c_uint64(len(hits)), _p(extra), C.c_uint64(len(extra)), _p(xc), C.c_uint64(len(xc)), _p(hoff), _p(score),
                                             _p(pos), _p(woff), _p(wlen), _p(coff), _p(cig), C.c_uint64(cap), C.byref(need))
    _chk(call(None, 0))
    cig = np.empty(need.value, np.uint8)
    _chk(call(cig, len(cig)))
    return dict(hit_off=hoff, score=score, pos=pos, win_off=woff, win_len=wlen, cig_off=coff, cigar=cig.tobytes())


def pack_reads(reads):
    """list of bytes -> (u8 array, u64 offsets[n+1])"""
    off = np.zeros(len(reads) + 1, np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), np.uint8) if reads else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), off


class Index:
    """An index resident in HBM: rank-sampled BWT blocks + SA + Ref1."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def open(cls, fm_path, ref_path, device=0):
        h = C.c_void_p()
        _chk(load().bwt_index_open(os.fsencode(fm_path), os.fsencode(ref_path), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, L, SA, pointers, shift, ref, device=0):
        L = np.ascontiguousarray(L, np.uint8)
        SA = np.ascontiguousarray(SA, np.uint32)
        ptr = (C.c_uint32 * 4)(*[int(x) for x in pointers])
        if not isinstance(ref, np.ndarray):
            ref = np.frombuffer(bytes(ref), np.uint8)
        ref = np.ascontiguousarray(ref, np.uint8)
        h = C.c_void_p()
        _chk(load().bwt_index_from_arrays(_p(L), _p(SA), C.c_uint32(len(L)), ptr, C.c_int32(int(shift)), _p(ref),
                                          C.c_uint64(len(ref)), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, ref, shift=1, rng_state=None, device=0):
        """bwt_build_index: ref = the chunk's .ref bytes.  rng_state: list of 3 ints (AS183), updated in place."""
        if not isinstance(ref, np.ndarray):
            ref = np.frombuffer(bytes(ref), np.uint8)
        ref = np.ascontiguousarray(ref, np.uint8)
        st = (C.c_uint32 * 3)(*(rng_state if rng_state is not None else (3172, 9814, 20125)))
        h = C.c_void_p()
        _chk(load().bwt_build_index(_p(ref), C.c_uint64(len(ref)), C.c_int32(int(shift)), st, int(device), C.byref(h)))
        if rng_state is not None:
            rng_state[:] = [int(x) for x in st]
        return cls(h)

    @classmethod
    def open_native(cls, path, device=0):
        h = C.c_void_p()
        _chk(load().bwt_index_open_native(os.fsencode(path), int(device), C.byref(h)))
        return cls(h)

    def save_native(self, path):
        _chk(load().bwt_index_save_native(self.h, os.fsencode(path)))

    def export(self):
        """-> (L u8[n_rows], SA u32[n_rows]) copied back from HBM."""
        n = self.meta()[0][3]
        L = np.empty(n, np.uint8); SA = np.empty(n, np.uint32)
        _chk(load().bwt_index_export(self.h, _p(L), _p(SA)))
        return L, SA

    def save(self, fm_path, ref_path):
        _chk(load().bwt_index_save(self.h, os.fsencode(fm_path), os.fsencode(ref_path)))

    def close(self):
        if self.h:
            load().bwt_index_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def meta(self):
        out = (C.c_uint32 * 5)()
        _chk(load().bwt_index_meta(self.h, out))
        pc, pg, pt, last, shift = [int(x) for x in out]
        return (pc, pg, pt, last), C.c_int32(shift).value

    def device_bytes(self):
        out = (C.c_uint64 * 4)()
        _chk(load().bwt_index_device_bytes(self.h, out))
        return dict(rank_blocks=int(out[0]), sa=int(out[1]), ref1=int(out[2]), ktable=int(out[3]))

    def set_ktable(self, k, wide=-1, summary=1):
        """Suffix-interval table: depth k (0 = none, 8..15, negative = the library's default for this index), whether its
        top level carries SA values (1; 2 = and the lower levels where that pays; 0; negative = default) and the summary bitmap
        (0 none, 1 where it pays, 2 everywhere)."""
        _chk(load().bwt_index_set_ktable_ex(self.h, int(k), int(wide), int(summary)))

    def set_options(self, **kw):
        """bwt_index_set_options: defaults of the batches made on this index from now on (no arguments = the library's own choices)."""
        o = make_options(**kw)
        _chk(load().bwt_index_set_options(self.h, C.byref(o)))

    @property
    def ktable(self):
        """(K, wide)"""
        v = int(load().bwt_index_ktable(self.h))
        return v & 15, bool(v & 16)

    @property
    def ktable_wide_levels(self):
        """levels below the top one that are kept as 64-byte entries as well (set_ktable(15, 2) / the default on a large index)"""
        v = int(load().bwt_index_ktable(self.h))
        return [m for m in range(9, 15) if (v >> m) & 1]

    def ref1(self, off, length):
        out = np.empty(length, np.uint8)
        _chk(load().bwt_ref1_read(self.h, C.c_uint64(off), C.c_uint64(length), _p(out)))
        return out.tobytes()

    def ref1_gather(self, win_off, win_len):
        """RefSeq of many hits in one call -> list of bytes."""
        win_off = np.ascontiguousarray(win_off, np.int64); win_len = np.ascontiguousarray(win_len, np.int32)
        n = len(win_off)
        ooff = np.empty(n + 1, np.uint64)
        need = C.c_uint64(0)
        out = np.empty(int(win_len.sum()) + 1, np.uint8)
        _chk(load().bwt_ref1_gather(self.h, _p(win_off), _p(win_len), C.c_uint64(n), _p(out), C.c_uint64(len(out)),
                                    _p(ooff), C.byref(need)))
        raw = out.tobytes()
        return [raw[int(ooff[i]):int(ooff[i + 1])] for i in range(n)]

    # ---- host-buffer entry points -------------------------------------------------------------
    def find_seeds_batch(self, bases, off):
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        status = np.empty(n, np.int32); sp = np.empty(n, np.uint32); ep = np.empty(n, np.uint32)
        nm = np.empty(n, np.uint32); soff = np.empty(n + 1, np.uint64)
        need = C.c_uint64(0)
        seeds = np.empty(0, np.int64)
        rc = load().bwt_find_seeds_batch(self.h, _p(bases), _p(off), C.c_uint32(n), _p(status), _p(sp), _p(ep), _p(nm),
                                         _p(soff), _p(seeds), C.c_uint64(0), C.byref(need))
        if rc == ENOSPC:
            seeds = np.empty(need.value, np.int64)
            rc = load().bwt_find_seeds_batch(self.h, _p(bases), _p(off), C.c_uint32(n), _p(status), _p(sp), _p(ep),
                                             _p(nm), _p(soff), _p(seeds), C.c_uint64(len(seeds)), C.byref(need))
        _chk(rc)
        return dict(status=status, sp=sp, ep=ep, nmatch=nm, seed_off=soff, seeds=seeds)

    def sga_batch(self, bases, off):
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        coff = np.empty(n + 1, np.uint64); rs = np.empty(n, np.int32)
        need = C.c_uint64(0)
        S = np.empty(0, np.int64); D = np.empty(0, np.int32)
        rc = load().bwt_sga_batch(self.h, _p(bases), _p(off), C.c_uint32(n), _p(coff), _p(S), _p(D), C.c_uint64(0),
                                  C.byref(need), _p(rs))
        if rc == ENOSPC:
            S = np.empty(need.value, np.int64); D = np.empty(need.value, np.int32)
            rc = load().bwt_sga_batch(self.h, _p(bases), _p(off), C.c_uint32(n), _p(coff), _p(S), _p(D),
                                      C.c_uint64(len(S)), C.byref(need), _p(rs))
        _chk(rc)
        return dict(clus_off=coff, S=S, D=D, read_status=rs)

    def align_batch(self, bases, off):
        """bwt_align_batch through the two-call pattern."""
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        need = (C.c_uint64 * 2)()
        rs = np.empty(n, np.int32); hoff = np.empty(n + 1, np.uint64)

        def call(nh, cb):
            score = np.empty(nh, np.int32); pos = np.empty(nh, np.int64); woff = np.empty(nh, np.int64)
            wlen = np.empty(nh, np.int32); coff = np.empty(nh + 1, np.uint64); cig = np.empty(cb, np.uint8)
            rc = load().bwt_align_batch(self.h, _p(bases), _p(off), C.c_uint32(n), _p(rs), _p(hoff), _p(score), _p(pos),
                                        _p(woff), _p(wlen), _p(coff), _p(cig), C.c_uint64(nh), C.c_uint64(cb), need)
            return rc, dict(read_status=rs, hit_off=hoff, score=score, pos=pos, win_off=woff, win_len=wlen,
                            cig_off=coff, cigar=cig.tobytes())
        rc, out = call(0, 0)
        if rc == ENOSPC:
            rc, out = call(int(need[0]), int(need[1]))
        _chk(rc)
        return out


def reverse_complement_batch(bases, off, device=0):
    """fastq:reverse_complement/1 on a packed batch -> (u8 array at the same offsets, status i32[n]: -1 = no clause)."""
    bases = np.ascontiguousarray(bases, np.uint8); off = np.ascontiguousarray(off, np.uint64)
    n = len(off) - 1
    out = np.zeros(len(bases), np.uint8); st = np.zeros(n, np.int32)
    _chk(load().bwt_reverse_complement_batch(int(device), _p(bases), _p(off), C.c_uint32(n), _p(out), _p(st)))
    return out, st


def otp_pick(scores):
    """Index of the head of lists:sort(fun R1 > R2, L) for the candidates' scores in list order (bwt_otp_pick)."""
    a = np.ascontiguousarray(scores, np.int32)
    return int(load().bwt_otp_pick(_p(a), C.c_uint32(len(a))))


def run_fastq(index, fastq_path, chromosome, threshold=265, batch_reads=0, out_path=None):
    """bwt_run_fastq: master + worker over a whole FASTQ file -> dict of the run's statistics."""
    st = RunStats()
    _chk(load().bwt_run_fastq(index.h, os.fsencode(fastq_path), str(chromosome).encode(), C.c_int32(threshold), C.c_uint32(batch_reads),
                              os.fsencode(out_path) if out_path else None, C.byref(st)))
    return {k: (float if k == "seconds" else int)(getattr(st, k)) for k, _ in st._fields_}


def seeds_generate_fs(qsec, length, defects):
    """bwt_seeds_generate_fs: the text of fs.erl as seeds:generate_fs/3 writes it (bytes)."""
    q = np.frombuffer(bytes(qsec), np.uint8)
    need = C.c_uint64(0)
    rc = load().bwt_seeds_generate_fs(_p(q), C.c_uint32(len(q)), C.c_uint32(length), C.c_uint32(defects), None, C.c_uint64(0),
                                      C.byref(need))
    if rc != ENOSPC:
        _chk(rc)
    buf = C.create_string_buffer(need.value)
    _chk(load().bwt_seeds_generate_fs(_p(q), C.c_uint32(len(q)), C.c_uint32(length), C.c_uint32(defects), buf,
                                      C.c_uint64(need.value), C.byref(need)))
    return buf.raw[:need.value]


def fs_find_seeds(qsec, length, defects, ref=None, index=None, device=0):
    """fs:find_seeds/1 of the module generate_fs(qsec, length, defects) writes, on `ref` (host bytes) or on the chunk of
    a resident index: 1-based positions, newest first (int64 array)."""
    q = np.frombuffer(bytes(qsec), np.uint8)
    need = C.c_uint64(0)
    if index is None:
        r = ref if isinstance(ref, np.ndarray) else np.frombuffer(bytes(ref), np.uint8)
        r = np.ascontiguousarray(r, np.uint8)

        def call(out, cap):
            return load().bwt_fs_find_seeds(int(device), _p(q), C.c_uint32(len(q)), C.c_uint32(length), C.c_uint32(defects),
                                            _p(r), C.c_uint64(len(r)), _p(out), C.c_uint64(cap), C.byref(need))
    else:
        def call(out, cap):
            return load().bwt_fs_find_seeds_index(index.h, _p(q), C.c_uint32(len(q)), C.c_uint32(length),
                                                  C.c_uint32(defects), _p(out), C.c_uint64(cap), C.byref(need))
    rc = call(None, 0)
    if rc == ENOSPC:
        out = np.empty(need.value, np.int64)
        rc = call(out, len(out))
        _chk(rc)
        return out
    _chk(rc)
    return np.empty(0, np.int64)


def sw_batch(q, qoff, r, roff, device=0):
    q = np.ascontiguousarray(q, np.uint8); r = np.ascontiguousarray(r, np.uint8)
    qoff = np.ascontiguousarray(qoff, np.uint64); roff = np.ascontiguousarray(roff, np.uint64)
    n = len(qoff) - 1
    status = np.empty(n, np.int32); score = np.empty(n, np.int32); coff = np.empty(n + 1, np.uint64)
    need = C.c_uint64(0)
    cig = np.empty(0, np.uint8)
    rc = load().bwt_sw_batch(int(device), _p(q), _p(qoff), _p(r), _p(roff), C.c_uint32(n), _p(status), _p(score),
                             _p(coff), _p(cig), C.c_uint64(0), C.byref(need))
    if rc == ENOSPC:
        cig = np.empty(need.value, np.uint8)
        rc = load().bwt_sw_batch(int(device), _p(q), _p(qoff), _p(r), _p(roff), C.c_uint32(n), _p(status), _p(score),
                                 _p(coff), _p(cig), C.c_uint64(len(cig)), C.byref(need))
    _chk(rc)
    return dict(status=status, score=score, cig_off=coff, cigar=cig.tobytes())


class Batch:
    """Resident batch: reads, intermediates and hits stay in HBM between calls."""

    def __init__(self, index, handle=None):
        self.index = index
        self.owned = handle is None
        self.h = C.c_void_p() if handle is None else handle       # handle: a batch owned by someone else (a pipe slot's)
        if self.owned:
            _chk(load().bwt_batch_create(index.h, C.byref(self.h)))
        self.n = 0

    def destroy(self):
        if self.h and self.owned:
            load().bwt_batch_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def set_options(self, **kw):
        o = make_options(**kw)
        _chk(load().bwt_batch_set_options(self.h, C.byref(o)))

    def set_reads(self, bases, off):
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        self.n = len(off) - 1
        _chk(load().bwt_batch_set_reads(self.h, _p(bases), _p(off), C.c_uint32(self.n)))

    def set_reads_packed(self, codes, exc, n, read_len=0, off=None, total=None):
        """bwt_batch_set_reads_packed: 2-bit codes + exceptions (pack_2bit); read_len > 0 for a uniform batch, else off[n+1]."""
        codes = np.ascontiguousarray(codes, np.uint32); exc = np.ascontiguousarray(exc, np.uint64)
        off64 = None if off is None else np.ascontiguousarray(off, np.uint64)
        if total is None:
            total = n * read_len if read_len else int(off64[n])
        self.n = n
        _chk(load().bwt_batch_set_reads_packed(self.h, _p(codes), C.c_uint64(total), C.c_uint32(n), C.c_uint32(read_len), _p(off64),
                                               _p(exc), C.c_uint64(len(exc))))

    def fetch_hits_compact(self):
        """The compact hit table as it sits in HBM -> dict(read_status i8[n], hits HIT_REC[], extra HIT_EXTRA[], cigar bytes)."""
        need = (C.c_uint64 * 3)()
        rc = load().bwt_batch_fetch_hits_compact(self.h, None, None, None, None, C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), need)
        if rc not in (0, ENOSPC):
            _chk(rc)
        nh, nx, cb = int(need[0]), int(need[1]), int(need[2])
        rs = np.empty(self.n, np.int8); hits = np.empty(nh, HIT_REC); extra = np.empty(nx, HIT_EXTRA); cig = np.empty(cb, np.uint8)
        _chk(load().bwt_batch_fetch_hits_compact(self.h, _p(rs), _p(hits), _p(extra), _p(cig), C.c_uint64(nh), C.c_uint64(nx), C.c_uint64(cb), need))
        return dict(read_status=rs, hits=hits, extra=extra, cigar=cig.tobytes())

    def fetch_hits_narrow(self):
        """The table in its narrow form (set_options(hits_narrow=1) before run) -> dict(read_status i8[n] with 2 = one plain hit in the
        narrow arrays, narrow_off u32[], narrow_score i16[], hits, extra, cigar)."""
        need = (C.c_uint64 * 4)()
        rc = load().bwt_batch_fetch_hits_narrow(self.h, None, None, None, None, None, None, C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), need)
        if rc not in (0, ENOSPC):
            _chk(rc)
        nn, nh, nx, cb = (int(x) for x in need)
        rs = np.empty(self.n, np.int8); no = np.empty(nn, np.uint32); ns = np.empty(nn, np.int16)
        hits = np.empty(nh, HIT_REC); extra = np.empty(nx, HIT_EXTRA); cig = np.empty(cb, np.uint8)
        _chk(load().bwt_batch_fetch_hits_narrow(self.h, _p(rs), _p(no), _p(ns), _p(hits), _p(extra), _p(cig), C.c_uint64(nn), C.c_uint64(nh), C.c_uint64(nx),
                                                C.c_uint64(cb), need))
        return dict(read_status=rs, narrow_off=no, narrow_score=ns, hits=hits, extra=extra, cigar=cig.tobytes())

    def set_reads_both_strands(self, bases, off):
        """SURVEY 8f-4 (not in the reference): the batch then holds 2n reads, read n + i = reverse complement of read i."""
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        _chk(load().bwt_batch_set_reads_both_strands(self.h, _p(bases), _p(off), C.c_uint32(n)))
        self.n = 2 * n

    def pair_hits(self, min_insert, max_insert):
        """Best proper combination per pair of interleaved mates (include/bwt_gpu.h: bwt_batch_pair_hits) -> PAIR_REC array."""
        n_pairs = self.n // 4
        out = np.empty(n_pairs, PAIR_REC)
        _chk(load().bwt_batch_pair_hits(self.h, C.c_uint32(n_pairs), C.c_int32(min_insert), C.c_int32(max_insert), _p(out)))
        return out

    def run(self, stages):
        c = BatchCounts()
        _chk(load().bwt_batch_run(self.h, int(stages), C.byref(c)))
        return c.as_dict()

    def profile(self, enable=True):
        _chk(load().bwt_batch_profile(self.h, 1 if enable else 0))

    def kernel_times(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)(); ln = (C.c_uint64 * 64)()
        k = _chk(load().bwt_batch_kernel_times(self.h, names, 4096, ms, ln, 64))
        parts = names.raw.split(b"\0")[:k]
        return {parts[i].decode(): dict(ms=float(ms[i]), launches=int(ln[i])) for i in range(k)}

    def hits_to_device(self, dev_ptr, cap_hits):
        _chk(load().bwt_batch_hits_to_device(self.h, C.c_void_p(dev_ptr), C.c_uint64(cap_hits)))

    def compact_to_device(self, hits_ptr, cap_hits, extra_ptr, cap_extra, cigar_ptr, cap_cigar, read_base):
        """bwt_batch_compact_to_device: the compact table into device memory (RCCL send buffer), read ids made global"""
        _chk(load().bwt_batch_compact_to_device(self.h, C.c_void_p(hits_ptr), C.c_uint64(cap_hits), C.c_void_p(extra_ptr), C.c_uint64(cap_extra),
                                                C.c_void_p(cigar_ptr), C.c_uint64(cap_cigar), C.c_uint32(read_base)))

    def hits_to_device_global(self, dev_ptr, cap_hits, read_base):
        _chk(load().bwt_batch_hits_to_device_global(self.h, C.c_void_p(dev_ptr), C.c_uint64(cap_hits), C.c_uint32(read_base)))

    def fetch_tasks(self):
        need = C.c_uint64(0)
        rc = load().bwt_batch_fetch_tasks(self.h, None, None, None, None, None, None, C.c_uint64(0), C.byref(need))
        if rc not in (0, ENOSPC):
            _chk(rc)
        k = need.value
        st = np.empty(k, np.int32); sp = np.empty(k, np.uint32); ep = np.empty(k, np.uint32)
        nm = np.empty(k, np.uint32); tr = np.empty(k, np.uint32); te = np.empty(k, np.uint32)
        _chk(load().bwt_batch_fetch_tasks(self.h, _p(st), _p(sp), _p(ep), _p(nm), _p(tr), _p(te), C.c_uint64(k),
                                          C.byref(need)))
        return dict(status=st, sp=sp, ep=ep, nmatch=nm, read=tr, end=te)

    def fetch_clusters(self):
        need = C.c_uint64(0)
        coff = np.empty(self.n + 1, np.uint64); rs = np.empty(self.n, np.int32)
        rc = load().bwt_batch_fetch_clusters(self.h, _p(coff), None, None, C.c_uint64(0), C.byref(need), _p(rs))
        if rc not in (0, ENOSPC):
            _chk(rc)
        S = np.empty(need.value, np.int64); D = np.empty(need.value, np.int32)
        _chk(load().bwt_batch_fetch_clusters(self.h, _p(coff), _p(S), _p(D), C.c_uint64(len(S)), C.byref(need), _p(rs)))
        return dict(clus_off=coff, S=S, D=D, read_status=rs)

    def fetch_hits(self):
        need = (C.c_uint64 * 2)()
        rs = np.empty(self.n, np.int32); hoff = np.empty(self.n + 1, np.uint64)
        rc = load().bwt_batch_fetch_hits(self.h, _p(rs), _p(hoff), None, None, None, None, None, None, C.c_uint64(0),
                                         C.c_uint64(0), need)
        if rc not in (0, ENOSPC):
            _chk(rc)
        nh, cb = int(need[0]), int(need[1])
        score = np.empty(nh, np.int32); pos = np.empty(nh, np.int64); woff = np.empty(nh, np.int64)
        wlen = np.empty(nh, np.int32); coff = np.empty(nh + 1, np.uint64); cig = np.empty(cb, np.uint8)
        _chk(load().bwt_batch_fetch_hits(self.h, _p(rs), _p(hoff), _p(score), _p(pos), _p(woff), _p(wlen), _p(coff),
                                         _p(cig), C.c_uint64(nh), C.c_uint64(cb), need))
        return dict(read_status=rs, hit_off=hoff, score=score, pos=pos, win_off=woff, win_len=wlen, cig_off=coff,
                    cigar=cig.tobytes())



def _view(ptr, count, dtype):
    """numpy view of `count` items of library-owned (pinned) memory"""
    if not count:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class Pipe:
    """bwt_pipe: `depth` slots with pinned host buffers; H2D, kernels and D2H of consecutive batches overlap.  Reads go up packed
    (2 bits per base + exceptions) or as ASCII; the compact hit table comes back."""

    def __init__(self, index, depth=3):
        self.index = index
        self.h = C.c_void_p()
        _chk(load().bwt_pipe_create(index.h, C.c_uint32(depth), C.byref(self.h)))

    def close(self):
        if self.h:
            load().bwt_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, **kw):
        o = make_options(**kw)
        _chk(load().bwt_pipe_set_options(self.h, C.byref(o)))

    def profile(self, enable=True):
        _chk(load().bwt_pipe_profile(self.h, 1 if enable else 0))

    def kernel_times(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)(); ln = (C.c_uint64 * 64)()
        k = _chk(load().bwt_pipe_kernel_times(self.h, names, 4096, ms, ln, 64))
        parts = names.raw.split(b"\0")[:k]
        return {parts[i].decode(): dict(ms=float(ms[i]), launches=int(ln[i])) for i in range(k)}

    def acquire(self, bases_cap, reads_cap):
        """ASCII form -> (slot, bases u8 view, off u64 view) of a free slot's pinned input buffers, or None when all are busy."""
        pb = C.c_void_p(); po = C.c_void_p(); slot = C.c_uint32(0)
        rc = load().bwt_pipe_acquire(self.h, C.c_uint64(bases_cap), C.c_uint32(reads_cap), C.byref(pb), C.byref(po), C.byref(slot))
        if rc == ENOSPC:
            return None
        _chk(rc)
        return slot.value, _view(pb.value, bases_cap, np.uint8), _view(po.value, reads_cap + 1, np.uint64)

    def submit(self, slot, n_reads):
        _chk(load().bwt_pipe_submit(self.h, C.c_uint32(slot), C.c_uint32(n_reads)))

    def acquire_packed(self, bases_cap, reads_cap, exc_cap=0):
        """Packed form -> (slot, dict of views: codes u32, exc u64, off u64, ascii u8 staging), or None when all slots are busy."""
        pin = PipeInput(); slot = C.c_uint32(0)
        rc = load().bwt_pipe_acquire_packed(self.h, C.c_uint64(bases_cap), C.c_uint32(reads_cap), C.c_uint64(exc_cap), C.byref(pin), C.byref(slot))
        if rc == ENOSPC:
            return None
        _chk(rc)
        words = int(load().bwt_packed_words(C.c_uint64(bases_cap)))
        return slot.value, dict(codes=_view(pin.codes, words, np.uint32), exc=_view(pin.exc, int(pin.exc_cap), np.uint64),
                                off=_view(pin.off, reads_cap + 1, np.uint64), ascii=_view(pin.ascii, bases_cap, np.uint8))

    def submit_packed(self, slot, n_reads, read_len, total_bases, n_exc):
        _chk(load().bwt_pipe_submit_packed(self.h, C.c_uint32(slot), C.c_uint32(n_reads), C.c_uint32(read_len), C.c_uint64(total_bases),
                                           C.c_uint64(n_exc)))

    def submit_ascii(self, slot, n_reads, read_len=0):
        _chk(load().bwt_pipe_submit_ascii(self.h, C.c_uint32(slot), C.c_uint32(n_reads), C.c_uint32(read_len)))

    def wait(self):
        """The oldest submitted batch -> dict of views into the slot's pinned result buffers (valid until release(slot)): the
        compact hit table (read_status i8, hits, extra, cigar); expand() gives the per-hit arrays of align_batch."""
        r = PipeResult()
        _chk(load().bwt_pipe_wait(self.h, C.byref(r)))
        n, nh, nx = r.n_reads, int(r.n_hits), int(r.n_extra)
        nn = int(r.n_narrow)
        return dict(slot=r.slot, n_reads=n, read_len=int(r.read_len), read_status=_view(r.read_status, n, np.int8),
                    narrow_off=_view(r.narrow_off, nn, np.uint32), narrow_score=_view(r.narrow_score, nn, np.int16),
                    hits=_view(r.hits, nh, HIT_REC), extra=_view(r.extra, nx, HIT_EXTRA),
                    cigar=_view(r.cigar, int(r.cigar_bytes), np.uint8).tobytes(), counts=r.counts.as_dict(),
                    stage_ms=dict(upload=r.ms_upload, kernels=r.ms_kernels, download=r.ms_download, waiting=r.ms_waiting))

    def expand(self, res, off=None):
        """per-hit arrays of a wait() result (host-side bwt_hits_expand); off = the batch's read offsets when the lengths differ"""
        out = hits_expand(res["n_reads"], res["read_len"], off, self.index.meta()[1], res["hits"], res["extra"], res["cigar"],
                          res["read_status"], res["narrow_off"], res["narrow_score"])
        st = res["read_status"].astype(np.int32)
        st[st == 2] = 0                                    # (the narrow form's "one plain hit" is an ordinary read with hits)
        out["read_status"] = st
        return out

    def batch_handle(self, slot):
        """the slot's resident batch (hit tables still in HBM until release): for the RCCL gather"""
        return C.c_void_p(load().bwt_pipe_batch(self.h, C.c_uint32(slot)))

    def release(self, slot):
        _chk(load().bwt_pipe_release(self.h, C.c_uint32(slot)))


class SmallBatch:
    """bwt_small: one call per batch of the reference's message size (<= SMALL_MAX_READS reads): pinned staging, no host round trip
    between the kernels, the compact hit table straight into pinned memory.  Larger or odd batches, and batches that overflow one of
    its fixed capacities, go through the general path inside the same call (path 0 / 2); results are identical either way."""

    def __init__(self, index):
        self.index = index
        self.h = C.c_void_p()
        _chk(load().bwt_small_create(index.h, C.byref(self.h)))
        self.last = None
        self.fallbacks = 0
        self.paths = {0: 0, 1: 0, 2: 0}
        self.graphs = {0: 0, 1: 0, 2: 0}              # calls by SmallResult.graph (launched one by one / replayed / captured)

    def close(self):
        if self.h:
            load().bwt_small_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, **kw):
        o = make_options(**kw)
        _chk(load().bwt_small_set_options(self.h, C.byref(o)))

    def align_raw(self, bases, off):
        """the C call alone (arrays must be contiguous u8 / u64 already) -> SmallResult"""
        r = SmallResult()
        _chk(load().bwt_small_align(self.h, _p(bases), _p(off), C.c_uint32(len(off) - 1), C.byref(r)))
        self.last = r
        self.paths[int(r.path)] += 1
        if r.path == 1:
            self.graphs[int(r.graph)] += 1
        if r.path == 2:
            self.fallbacks += 1
        return r

    def align(self, bases, off, expand=True):
        """-> the compact table (views into the handle's buffers, valid until the next call), or with expand the per-hit arrays of
        Index.align_batch"""
        bases = np.ascontiguousarray(bases, np.uint8); off = np.ascontiguousarray(off, np.uint64)
        r = self.align_raw(bases, off)
        n, nh, nx = int(r.n_reads), int(r.n_hits), int(r.n_extra)
        comp = dict(read_status=_view(r.read_status, n, np.int8), hits=_view(r.hits, nh, HIT_REC), extra=_view(r.extra, nx, HIT_EXTRA),
                    cigar=_view(r.cigar, int(r.cigar_bytes), np.uint8), path=int(r.path), overflow_flags=int(r.overflow_flags), graph=int(r.graph),
                    counts=dict(n_clusters=int(r.n_candidates), n_dp=int(r.n_dp), n_float=int(r.n_float), n_hits=nh))
        if not expand:
            return comp
        out = hits_expand(n, 0, off - off[0], self.index.meta()[1], comp["hits"], comp["extra"], comp["cigar"])
        out["read_status"] = comp["read_status"].astype(np.int32)
        out["path"] = comp["path"]; out["counts"] = comp["counts"]; out["overflow_flags"] = comp["overflow_flags"]; out["graph"] = comp["graph"]
        return out

    def stage_us(self):
        r = self.last
        return dict(copy_in=r.us_copy_in, enqueue=r.us_enqueue, wait=r.us_wait, total=r.us_total)

    def profile(self, enable=True):
        _chk(load().bwt_small_profile(self.h, 1 if enable else 0))

    def kernel_times(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)(); ln = (C.c_uint64 * 64)()
        k = _chk(load().bwt_small_kernel_times(self.h, names, 4096, ms, ln, 64))
        parts = names.raw.split(b"\0")[:k]
        return {parts[i].decode(): dict(ms=float(ms[i]), launches=int(ln[i])) for i in range(k)}
