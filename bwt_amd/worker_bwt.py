"""Host mirror of the `running` clause of worker_bwt:worker_loop/10
(/root/reference/src/worker_bwt.erl:73-112): one batch of reads -> the messages the worker
casts to the master, in the reference's order.

    {cigar, SeqName, {Score,Cigar}, Pos, RefSeq}   worker_bwt.erl:102,105
    no_cigar                                       worker_bwt.erl:100
    (nothing for a read whose sga result is [])    worker_bwt.erl:77

The two call sites the reference makes per read (sga:sga/6 at :76, sw:sw/2 at :92) are replaced
by ONE bwt_align_batch call for the whole QseqList; the final pick among several alignments
(lists:sort/2 with a non-total comparator, :104) is done here with otp_lists.sort.
"""
from . import lib, otp_lists


class WorkerCrash(Exception):
    """The reference process would have crashed on this read (bad last base: bwa.erl:15-24;
    window outside Ref1: worker_bwt.erl:89).  Carries the read name; other reads are unaffected
    (the documented divergence of SURVEY 8b)."""


def _small(fm):
    """the index's small-batch handle (bwt_small: pinned staging, one synchronisation per call), made on first use"""
    sb = getattr(fm, "_small_batch", None)
    if sb is None:
        sb = fm._small_batch = lib.SmallBatch(fm.index)
    return sb


def align_batch(fm, qseq_list):
    """qseq_list: [(Name, Seq)].  -> per read None (sga = []), WorkerCrash, or the `Cigars` list
    [((Score,Cigar), Pos, RefSeq)] in the reference's list order.  One bwt_small_align call: a QseqList (200 reads,
    src/master.erl:59) or a whole workload message (10 of them) is the size that entry point is for; anything it does not take
    runs through the general path inside the same call."""
    reads = [bytes(s) for _, s in qseq_list]
    bases, off = lib.pack_reads(reads)
    out = _small(fm).align(bases, off)
    refseqs = fm.index.ref1_gather(out["win_off"], out["win_len"])         # RefSeq of every hit in one call
    res = []
    for i, (name, _) in enumerate(qseq_list):
        st = int(out["read_status"][i])
        if st == 1:
            res.append(None)
            continue
        if st < 0:
            res.append(WorkerCrash(name))
            continue
        cig = []
        for h in range(int(out["hit_off"][i]), int(out["hit_off"][i + 1])):
            text = out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode()
            cig.append(((int(out["score"][h]), text), int(out["pos"][h]), refseqs[h]))
        res.append(cig)
    return res


def pick(cigars):
    """worker_bwt.erl:99-106."""
    if not cigars:
        return "no_cigar"
    if len(cigars) == 1:
        return cigars[0]
    return otp_lists.sort(lambda a, b: a[0][0] > b[0][0], cigars)[0]


def running(fm, qseq_list):
    """One QseqList of the workload -> list of casts in the order the reference sends them.
    Seeds is built by prepending in the foldl (worker_bwt.erl:74-80), so reads are answered in
    REVERSE batch order."""
    return _casts(qseq_list, align_batch(fm, qseq_list))


def _casts(qseq_list, per_read):
    casts = []
    for (name, _), cig in reversed(list(zip(qseq_list, per_read))):
        if cig is None:
            continue
        if isinstance(cig, Exception):
            casts.append(("error", name, cig))
            continue
        top = pick(cig)
        if top == "no_cigar":
            casts.append("no_cigar")
        else:
            (score_cigar, pos, refseq) = top
            casts.append(("cigar", name.decode() if isinstance(name, bytes) else name, score_cigar, pos, refseq))
    return casts



def workload(fm, qseq_lists):
    """A whole {workload, [QseqList, ...]} message (src/worker_bwt.erl:62-69) in ONE device call: the reads of all its lists go up
    together, and the casts come out exactly as the reference sends them -- list by list in message order
    (worker_loop(running, [QseqList | WorkloadRest], ...), :73,112), inside a list in reverse read order (:74-80)."""
    flat = [item for ql in qseq_lists for item in ql]
    per_read = align_batch(fm, flat) if flat else []
    casts = []
    at = 0
    for ql in qseq_lists:
        casts.extend(_casts(ql, per_read[at:at + len(ql)]))
        at += len(ql)
    return casts
