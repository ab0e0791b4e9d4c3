"""Host mirror of fastq:read_seq/1, read_seqs/2 (/root/reference/src/fastq.erl:5-40) over the native
parser of libbwtgpu (bwt_fastq_*): records come back in the reference's list order (reverse file order)
as (Name, Seq) byte strings, plus the packed arrays the GPU path takes directly."""
import ctypes as C

import numpy as np

from . import lib


class Dev:
    """The `Dev` of file:open/2: a FASTQ opened for sequential read_seqs calls."""

    def __init__(self, path):
        self.h = C.c_void_p()
        lib._chk(lib.load().bwt_fastq_open(str(path).encode(), C.byref(self.h)))

    def close(self):
        if self.h:
            lib.load().bwt_fastq_close(self.h)
            self.h = None

    def set_threads(self, threads=0, min_records=65536):
        """0 = automatic, 1 = the sequential reader only; batches of at least min_records records use the threaded reader."""
        lib._chk(lib.load().bwt_fastq_set_threads(self.h, C.c_uint32(threads), C.c_uint64(min_records)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_seqs_packed(dev, n):
    """-> (state, bases u8, off u64[k+1], names bytes, name_off u64[k+1]);  state: 'ok' | 'eof' | 'eof_partial'."""
    l = lib.load()
    off = np.zeros(n + 1, np.uint64); noff = np.zeros(n + 1, np.uint64)
    k = C.c_uint32(0); st = C.c_int(0); need = (C.c_uint64 * 2)()
    bases = np.empty(n * 160 + 64, np.uint8); names = np.empty(n * 64 + 64, np.uint8)
    rc = l.bwt_fastq_read_seqs(dev.h, C.c_uint32(n), lib._p(bases), C.c_uint64(len(bases)), lib._p(off), lib._p(names),
                               C.c_uint64(len(names)), lib._p(noff), C.byref(k), C.byref(st), need)
    if rc == lib.ENOSPC:
        bases = np.empty(int(need[0]) + 64, np.uint8); names = np.empty(int(need[1]) + 64, np.uint8)
        rc = l.bwt_fastq_read_seqs(dev.h, C.c_uint32(n), lib._p(bases), C.c_uint64(len(bases)), lib._p(off),
                                   lib._p(names), C.c_uint64(len(names)), lib._p(noff), C.byref(k), C.byref(st), need)
    lib._chk(rc)
    kk = k.value
    state = ("ok", "eof", "eof_partial")[st.value]
    return state, bases[: int(off[kk])], off[: kk + 1], names[: int(noff[kk])].tobytes(), noff[: kk + 1]


def read_seqs(dev, n):
    """fastq:read_seqs(Dev, N) -> ('ok', Acc) | 'eof' | ('eof', Acc) with Acc = [(Name, Seq)] newest first."""
    state, bases, off, names, noff = read_seqs_packed(dev, n)
    acc = [(names[int(noff[i]):int(noff[i + 1])], bases[int(off[i]):int(off[i + 1])].tobytes())
           for i in range(len(off) - 1)]
    if state == "ok":
        return "ok", acc
    if state == "eof":
        return "eof"
    return "eof", acc


def reverse_complement(ls):
    """fastq:reverse_complement(Ls) (src/fastq.erl:43-50) on the GPU; a base outside ACGTN raises (the reference has no clause)."""
    ls = bytes(ls)
    if not ls:
        return b""
    out, st = lib.reverse_complement_batch(*lib.pack_reads([ls]))
    if st[0] != 0:
        raise ValueError("fastq:rc/2: no function clause matching (a base outside ACGTN)")
    return out.tobytes()
