"""Host mirror of bwa:find_seeds/6 (/root/reference/src/bwa.erl:12-25).  The LF steps run in the
k_search HIP kernel; results keep the reference's shapes: 'no_seeds' | 'too_many_seeds' | [int]."""
from . import lib

NO_SEEDS = "no_seeds"
TOO_MANY = "too_many_seeds"


class CaseClause(Exception):
    """bwa.erl:15-24: the last base of the sub-read is none of A,C,G,T,N (the reference crashes)."""


def find_seeds_many(fm, subseqs):
    bases, off = lib.pack_reads([bytes(s) for s in subseqs])
    out = fm.index.find_seeds_batch(bases, off)
    res = []
    for i in range(len(subseqs)):
        st = int(out["status"][i])
        if st == 0:
            res.append(out["seeds"][int(out["seed_off"][i]):int(out["seed_off"][i + 1])].tolist())
        elif st == 1:
            res.append(NO_SEEDS)
        elif st == 2:
            res.append(TOO_MANY)
        else:
            res.append(CaseClause(bytes(subseqs[i])[-1:]))
    return res


def find_seeds(fm, pc, pg, pt, last, subseq):
    """find_seeds(FM, Pc,Pg,Pt,Last, Subseq): the pointer arguments are part of the reference's
    signature; they must be the index's own (they are baked into the device copy)."""
    assert (pc, pg, pt, last) == (fm.Pc, fm.Pg, fm.Pt, fm.Last)
    r = find_seeds_many(fm, [subseq])[0]
    if isinstance(r, Exception):
        raise r
    return r
