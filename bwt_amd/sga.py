"""Host mirror of sga:sga/6 (/root/reference/src/sga.erl:17): -> [(S, D)] in the reference's list
order (descending S).  skip_Ns, the sub-read chain, the SA gather, sort and get_similar all run on
the GPU (k_subreads, k_search, k_cluster_*)."""
from . import lib
from .bwa import CaseClause


def sga_many(fm, qseqs):
    bases, off = lib.pack_reads([bytes(q) for q in qseqs])
    out = fm.index.sga_batch(bases, off)
    res = []
    for i in range(len(qseqs)):
        if out["read_status"][i] < 0:
            res.append(CaseClause(i))
            continue
        a, b = int(out["clus_off"][i]), int(out["clus_off"][i + 1])
        res.append(list(zip(out["S"][a:b].tolist(), out["D"][a:b].tolist())))
    return res


def sga(fm, pc, pg, pt, last, qseq):
    assert (pc, pg, pt, last) == (fm.Pc, fm.Pg, fm.Pt, fm.Last)
    r = sga_many(fm, [qseq])[0]
    if isinstance(r, Exception):
        raise r
    return r
