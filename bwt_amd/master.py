"""Host mirror of the master's side of the job (/root/reference/src/master.erl): workload production
(produce_workload/2,3, :162-178), the result sink (handle_cast cigar / no_cigar, :149-158) and the
end-of-run statistics (:72-91) -- enough to run a FASTQ against an index end to end on the GPU:

    python -m bwt_amd.master SeqFileName Chromosome [--fm-indices DIR] [--bwt-files DIR] [--threshold 265]

prints the reference's result lines  "Name      Chrom      Pos      Cigar      Score      RefSeq".
The transport (navel) and the gen_server shell are out of scope; the message shapes are the reference's."""
import sys
import time

from . import fastq, fm_index, worker_bwt

WORKLOAD_SIZE = 200          # master.erl:59 #state.workload_size
WATERLINE = 10               # worker_bwt.erl:16
CIGAR_RATE_THRESHOLD = 265   # master.erl:149


class State:
    def __init__(self, fastq_dev, chromosome):
        self.fastq = fastq_dev
        self.fastq_eof = False
        self.stopping = False
        self.chromosome = chromosome
        self.stat_workload_amount = 0


def produce_workload(n, s):
    """master.erl:162-178 -> 'stop' | [[(Name,Seq)]] (lists prepended: newest batch first)."""
    acc = []
    while n > 0:
        if s.fastq_eof:
            if not acc:
                s.stopping = True
                return "stop"
            raise RuntimeError("function_clause produce_workload")     # the reference has no such clause either
        r = fastq.read_seqs(s.fastq, WORKLOAD_SIZE)
        if r == "eof":
            s.fastq_eof = True
            return acc
        acc.insert(0, r[1])                                              # {_, SeqList}: ok and partial-eof alike
        n -= 1
    return acc


def get_workload(n, s):
    """handle_call({get_workload, N}) (master.erl:127-137)"""
    if s.stopping:
        return "stop"
    r = produce_workload(n, s)
    if r != "stop":
        s.stat_workload_amount += sum(len(q) for q in r)
    return r


def cigar_line(cast, chromosome, threshold=CIGAR_RATE_THRESHOLD):
    """handle_cast({cigar,...}) (master.erl:149-155): None when the rate is under the threshold, else the
    printed line (io:format("~n~s      ~s      ~b      ~s      ~b      ~s~n", ...))."""
    _, name, (rate, cigar), pos, refseq = cast
    if rate < threshold:
        return None
    if isinstance(name, bytes):
        name = name.decode("latin1")
    if isinstance(refseq, bytes):
        refseq = refseq.decode("latin1")
    return "\n%s      %s      %d      %s      %d      %s\n" % (name, chromosome, pos, cigar, rate, refseq)


def run(seq_file, chromosome, out=sys.stdout, threshold=CIGAR_RATE_THRESHOLD, device=0):
    """master:run/4 + one GPU worker: returns (reads dispatched, result lines printed, seconds)."""
    dev = fastq.Dev(seq_file)
    s = State(dev, chromosome)
    _, fm = fm_index.get_index(chromosome, 1, device)
    t0 = time.time()
    lines = 0
    try:
        while True:
            wl = get_workload(WATERLINE, s)
            if wl == "stop":
                break
            # one align_batch for the whole workload (the worker would do one per QseqList); cast order kept
            for qlist in wl:
                for cast in worker_bwt.running(fm, qlist):
                    if cast == "no_cigar" or cast[0] != "cigar":
                        continue
                    line = cigar_line(cast, chromosome, threshold)
                    if line is not None:
                        out.write(line)
                        lines += 1
    finally:
        fm.close()
        dev.close()
    sec = time.time() - t0
    out.write("\n\nReads: %r\nReference seq: %r\nChromosomes: %r\nReads aligned: %d\nAlignment completion time: %.1f sec\n"
              "Workers: %d\n\n" % (seq_file, "human_g1k_v37_decoy.fasta", chromosome, s.stat_workload_amount, sec, 1))
    return s.stat_workload_amount, lines, sec


def run_native(seq_file, chromosome, out_path=None, threshold=CIGAR_RATE_THRESHOLD, device=0, batch_reads=0):
    """The same job through bwt_run_fastq: threaded FASTQ reader -> pinned-buffer pipeline -> pick -> result lines, all in the
    library (no per-read Python).  Writes the result lines (not the statistics block) to out_path; returns the statistics."""
    from . import lib
    _, fm = fm_index.get_index(chromosome, 1, device)
    try:
        return lib.run_fastq(fm.index, seq_file, chromosome, threshold, batch_reads, out_path)
    finally:
        fm.close()


def main(argv=None):
    import argparse
    import os
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("seq_file")
    ap.add_argument("chromosome")
    ap.add_argument("--fm-indices", default="fm_indices")
    ap.add_argument("--bwt-files", default="bwt_files")
    ap.add_argument("--threshold", type=int, default=CIGAR_RATE_THRESHOLD,
                    help="master.erl:149 drops results whose score is below this (265 silences every 100 bp read)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--native", metavar="OUT", help="run the whole job in the library (bwt_run_fastq) and write the result lines to OUT")
    a = ap.parse_args(argv)
    fm_index.set_env("fm_indices", a.fm_indices)
    path = a.seq_file if os.path.exists(a.seq_file) else os.path.join(a.bwt_files, a.seq_file)
    if a.native:
        st = run_native(path, a.chromosome, a.native, a.threshold, a.device)
        sys.stdout.write("\n\nReads: %r\nChromosomes: %r\nReads aligned: %d\nAlignment completion time: %.1f sec\nWorkers: %d\n\n"
                         % (path, a.chromosome, st["reads"], st["seconds"], 1))
        return
    run(path, a.chromosome, threshold=a.threshold, device=a.device)


if __name__ == "__main__":
    main()
