"""Host mirror of sw:sw/2 (/root/reference/src/sw.erl:47-49): 'no_match' | (Score, Cigar)."""
from . import lib

NO_MATCH = "no_match"


def sw_many(pairs, device=0):
    q, qoff = lib.pack_reads([bytes(p[0]) for p in pairs])
    r, roff = lib.pack_reads([bytes(p[1]) for p in pairs])
    out = lib.sw_batch(q, qoff, r, roff, device)
    res = []
    for i in range(len(pairs)):
        if out["status"][i] != 0:
            res.append(NO_MATCH)
        else:
            res.append((int(out["score"][i]), out["cigar"][int(out["cig_off"][i]):int(out["cig_off"][i + 1])].decode()))
    return res


def sw(qseq, ref, device=0):
    """sw([_|_]=Qseq,[_|_]=Ref): both must be non-empty (anything else leaves the clause the worker uses)."""
    if len(qseq) == 0 or len(ref) == 0:
        raise ValueError("sw:sw/2 needs two non-empty sequences")
    return sw_many([(qseq, ref)], device)[0]
