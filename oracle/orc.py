"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liborc.so (bwt_oracle.c)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _FM(C.Structure):
    _fields_ = [("n", C.c_uint32), ("L", C.c_void_p), ("I", C.c_void_p), ("SA", C.c_void_p),
                ("Pc", C.c_uint32), ("Pg", C.c_uint32), ("Pt", C.c_uint32), ("Last", C.c_uint32),
                ("shift", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_find_seeds.restype = C.c_int
        _LIB.orc_sga.restype = C.c_int
        _LIB.orc_sw.restype = C.c_int
        _LIB.orc_align_read.restype = C.c_int
        _LIB.orc_batch.restype = C.c_int
    return _LIB


class OrcFM:
    """Holds the L / I / SA columns (u8 / u32 / u32) and the pointers for the C oracle."""

    def __init__(self, L, I, SA, pointers, shift):
        self.L = np.ascontiguousarray(L, np.uint8)
        self.I = np.ascontiguousarray(I, np.uint32)
        self.SA = np.ascontiguousarray(SA, np.uint32)
        self.pointers = tuple(int(x) for x in pointers)
        self.shift = int(shift)
        self.c = _FM(len(self.L), self.L.ctypes.data, self.I.ctypes.data, self.SA.ctypes.data,
                     *self.pointers, self.shift)

    @classmethod
    def from_L_SA(cls, L, SA, pointers, shift):
        """Index columns exported from anywhere: the I column is derived by the oracle itself."""
        L = np.ascontiguousarray(L, np.uint8)
        I = np.empty(len(L), np.uint32)
        lib().orc_make_I(L.ctypes.data_as(C.c_void_p), C.c_uint32(len(L)), C.c_uint32(pointers[0]),
                         C.c_uint32(pointers[1]), C.c_uint32(pointers[2]), I.ctypes.data_as(C.c_void_p))
        return cls(L, I, SA, pointers, shift)

    @classmethod
    def from_etf(cls, d):
        return cls(d["L"], d["I"], d["SA"], d["pointers"], d["shift"])


def find_seeds(fm, sub):
    """-> (status, [seed ends in reference list order], sp, ep, n).  status: 0 seeds, 1 no_seeds,
    2 too_many_seeds, -1 crash."""
    sub = bytes(sub)
    cap = 1 << 16
    out = np.empty(cap, np.int64)
    n = C.c_int(0); sp = C.c_uint32(0); ep = C.c_uint32(0); nn = C.c_int(0)
    st = lib().orc_find_seeds(C.byref(fm.c), sub, len(sub), out.ctypes.data_as(C.c_void_p), cap,
                              C.byref(n), C.byref(sp), C.byref(ep), C.byref(nn))
    return st, out[:n.value].tolist() if st == 0 else [], sp.value, ep.value, nn.value


def sga(fm, q, cap=4096):
    q = bytes(q)
    S = np.empty(cap, np.int64); D = np.empty(cap, np.int32)
    stats = np.zeros(3, np.int64)
    n = lib().orc_sga(C.byref(fm.c), q, len(q), S.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p),
                      cap, stats.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise RuntimeError("orc_sga failed: %d" % n)
    return list(zip(S[:n].tolist(), D[:n].tolist()))


def sw(q, ref):
    q = bytes(q); ref = bytes(ref)
    cap = 4 * (len(q) + len(ref)) + 16
    buf = C.create_string_buffer(cap)
    score = C.c_int(0)
    rc = lib().orc_sw(q, len(q), ref, len(ref), C.byref(score), buf, cap)
    if rc == 1:
        return "no_match"
    if rc != 0:
        raise RuntimeError("orc_sw failed: %d" % rc)
    return score.value, buf.value.decode()


def align_read(fm, ref1, q, cap=256):
    """-> None (read emits nothing) | [(score, cigar, pos, win_off, win_len)] in reference list order;
    raises RuntimeError on the inputs the reference would crash on."""
    q = bytes(q)
    while True:
        score = np.empty(cap, np.int32); pos = np.empty(cap, np.int64)
        woff = np.empty(cap, np.int64); wlen = np.empty(cap, np.int32)
        cigcap = max(1 << 16, 64 * cap)
        cig = C.create_string_buffer(cigcap)
        n = lib().orc_align_read(C.byref(fm.c), ref1, C.c_int64(len(ref1)), q, len(q),
                                 score.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                                 woff.ctypes.data_as(C.c_void_p), wlen.ctypes.data_as(C.c_void_p),
                                 cig, cigcap, cap, None)
        if n == -4 and cap < (1 << 20):           # the caller's arrays were too small (a read with hundreds of hits): once more, larger
            cap *= 16
            continue
        break
    if n == -1:
        return None
    if n < 0:
        raise RuntimeError("orc_align_read failed: %d" % n)
    texts = cig.raw.split(b"\0")[:n]
    return [(int(score[k]), texts[k].decode(), int(pos[k]), int(woff[k]), int(wlen[k])) for k in range(n)]


def batch(fm, ref1, bases, off, mode, nthreads):
    """Time-able batch driver.  bases: u8 array of concatenated reads, off: u32[n+1].
    -> (reads_with_alignment, clusters_or_candidates, sw_cells, checksum)"""
    bases = np.ascontiguousarray(bases, np.uint8); off = np.ascontiguousarray(off, np.uint32)
    out = np.zeros(4, np.int64)
    lib().orc_batch(C.byref(fm.c), ref1, C.c_int64(len(ref1)), bases.ctypes.data_as(C.c_void_p),
                    off.ctypes.data_as(C.c_void_p), C.c_uint32(len(off) - 1), mode, nthreads,
                    out.ctypes.data_as(C.c_void_p))
    return tuple(int(x) for x in out)


def fs_patterns(qsec, length, defects):
    """Patterns of seeds:generate_fs(Qsec, Len, Defects) in generation order (items: byte, or 256 for `_`) as an
    (npat, length) u16 array; raises ValueError where the generated fs.erl would not compile."""
    qsec = bytes(qsec)
    lib().orc_fs_patterns.restype = C.c_long
    p = C.c_void_p()
    n = lib().orc_fs_patterns(qsec, len(qsec), int(length), int(defects), C.byref(p))
    if n < 0:
        raise ValueError("generate_fs(%r, %d, %d): the generated module would not compile" % (qsec, length, defects))
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(n * length,)).copy().reshape(n, length) if n else \
        np.zeros((0, length), np.uint16)
    lib().orc_free(p)
    return out


def fs_find_seeds(pats, ref):
    """fs:find_seeds/1 of the generated module on `ref`: 1-based positions, newest first."""
    pats = np.ascontiguousarray(pats, np.uint16)
    ref = bytes(ref)
    lib().orc_fs_find_seeds.restype = C.c_long
    cap = len(ref) // max(1, pats.shape[1]) + 2
    out = np.empty(cap, np.int64)
    n = lib().orc_fs_find_seeds(pats.ctypes.data_as(C.c_void_p), C.c_long(pats.shape[0]), int(pats.shape[1]), ref,
                                C.c_long(len(ref)), out.ctypes.data_as(C.c_void_p), C.c_long(cap))
    return out[:n].tolist()


def batch_digest(fm, ref1, bases, off, nthreads):
    """Per-read digests of the complete answer (see bwt_oracle.c: orc_batch_digest) -> u64 array."""
    bases = np.ascontiguousarray(bases, np.uint8); off = np.ascontiguousarray(off, np.uint32)
    dig = np.zeros(len(off) - 1, np.uint64)
    lib().orc_batch_digest(C.byref(fm.c), ref1, C.c_int64(len(ref1)), bases.ctypes.data_as(C.c_void_p),
                           off.ctypes.data_as(C.c_void_p), C.c_uint32(len(off) - 1), int(nthreads), dig.ctypes.data_as(C.c_void_p))
    return dig


def hits_digest(out):
    """The same digests from a hit table as the C ABI returns it (dict of arrays: read_status, hit_off, score, pos, win_off,
    win_len, cig_off, cigar), vectorised."""
    hoff = np.asarray(out["hit_off"]).astype(np.int64)
    n = len(hoff) - 1
    nh = int(hoff[-1])
    dig = np.zeros(n, np.uint64)
    st = np.asarray(out["read_status"])
    dig[st < 0] = 1
    if nh == 0:
        return dig
    coff = np.asarray(out["cig_off"]).astype(np.int64)
    text = np.frombuffer(out["cigar"], np.uint8).astype(np.uint64)
    lens = np.diff(coff)
    idx = np.arange(len(text), dtype=np.int64) - np.repeat(coff[:-1], lens) + 1
    th = np.zeros(nh, np.uint64)
    nonempty = lens > 0
    th[nonempty] = np.add.reduceat(text * (idx * idx).astype(np.uint64), coff[:-1][nonempty])
    with np.errstate(over="ignore"):
        v = (np.asarray(out["pos"]).astype(np.int64).astype(np.uint64) * np.uint64(1000003) +
             np.asarray(out["score"]).astype(np.int64).astype(np.uint64) * np.uint64(10007) +
             np.asarray(out["win_len"]).astype(np.int64).astype(np.uint64) * np.uint64(101) +
             np.asarray(out["win_off"]).astype(np.int64).astype(np.uint64) * np.uint64(7) + th)
        counts = np.diff(hoff)
        rank = (np.arange(nh, dtype=np.int64) - np.repeat(hoff[:-1], counts) + 1).astype(np.uint64)
        v = v * rank
        has = counts > 0
        dig[has] = np.add.reduceat(v, hoff[:-1][has])
    return dig
