#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/erlang/capture_inputs.txt -- the seeded inputs on which a machine
that HAS Erlang/OTP should run the real reference (oracle/capture_golden.escript) to pin the oracle.
One record per line:  sga <read>  |  seeds <sub-read>  |  sw <query> <window>  |  align <name> <read>  |  pick <scores>  |
fastq <file> <counts>  |  genfs <query> <len> <defects> <reference>  |  line <name> <chrom> <pos> <cigar> <score> <refseq>
(record kinds and the reference functions they drive: oracle/capture_golden.escript).  The FASTQ files of the `fastq` records are
written next to it (tests/golden/erlang/fastq/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import REF_PATH, synth_reads          # noqa: E402
import numpy as np                                   # noqa: E402


# the reference's own harness inputs: the 33 queries of bwt:test/0 (src/bwt.erl:24-57) -- data, not code
BWT_TEST_QUERIES = """
ACCCCACGTTTTTGGAGTTATATGTTGGCACTGATACTGGCCATAGAATTCCCTATGGTA
ACCNNNNCCACGTTTTTGGAGTTATATGTTGATACTGGCCATAGAATTCCCTATNGGTA
ACCCCACGTTTTTGGAGTTATATGTTGTTCTTGCACTGATACTGGCCATAGAATTCCCTATGGTA
TAATCAGAACAGGTTTACAACATAAATAAATAGATTGAACTTACTTTGTATAAAAATTGT
CTCAGCCTCCATAATTATGTGAACCAGTTCCCCTAATGAATCTTCTCTCATCTGTCTACA
TATATCCTATTGATTCTGCCTTTCTGGAGACCCCTGACTAATGTGATTACAATAACTACA
CAATTCACTAGTTTATATAGAAGACTTGGTTTTTGTCTTTGCCCCATTTTATATTTGTAT
TATAACTATGTATCTGGAAAATGGAACAAGTTTTTTCTTCTTCATATGAGGGCTAAGGCT
TTTTTCTCACCAATATTTTTGGAGATTTTAAAGATTTTCTTTTTTTTTGACATAGAATCT
TATGGAGGCTGAGAAATAATTTTTTTTCTATTTTATTCTTCAGCCCCAGGTGTTTGCTTT
TGCAGATTCTTGAGCACACTGAGAGCCTCCAAGGCATGGAGTGGGGTGCCTGAAGTTTCA
CAATATCATCAGAAGCCTATAACCAGAGTACCTGTCATAGTCTTTTCTGTGAGTCTCAGA
GGGAGTCCTGTCTTGGAGACGAACATTCTGACCTGTAGTTGATTGCAGGAGCTTTCAGGA
AAGCATCAGGGGGAAATAATATCTAAATGACGAAAAGTATGAAATGGCTGTGATGAAAGA
TCTGATGAGAGTTCATTATACCACAACTGACAAGGATATTCGATTTTTTCTGTGGCAGAC
AACATTTATTTATTTCTTTATTTAGAGACAGAGTCTTGCTCTGTCGCCCAGGCTGGAGTG
CAGCGGTGCGATCTCGGCTCACTGCAAGCTCTGTCTCCTGGGTTCACGCCATTCTCCTGC
CTCAGCCTCCCGAGTAGCTGGGACTACAGGTGCCTGCCATCACGCCCGGCTAATTTTTTA
TATTTTTAATAGAGATGGAGATTCACCGGGTTAGCCAGGATGGTCTGGATCTCCTGACCT
TGTGATCCACCCGCCTCAGCCTCCTAAAGTGCTGGGATTATAGGCATGAGCCACTGTGCC
TGGCACAACATTTAAAGTAATAATTGGAATTATGACTCATTACTCTATAGTGGCACATAG
CATGGATAAGGAGGACATTGACAAACTTCCAGGAATTTTATATAATTTCTGAAAACATAA
CATTTTACCCATACAAATATAACACAGGGAAGGTTAGGTATCTCTTTTTATTTGTATCTT
CTGTATGGTTTTCCTTATAAAAAATGCAACCTACTTTACTTGCGAAACATGCCCTACTTT
TCTTGCATGCTTTGCATAGAGTTGTTTCTAGTTATTCTATTATTTCTAGTAGTTTTATTT
ACATATATTGATTATAATTTTAATACTTAGTAATCTTTTATTTTCCAGAGAAAACTAGGA
AGTAGACAGTTATAAACTGTCATATATTAGCATTCTATAGTAGGTTAGAAAATGTATGAA
TATACCATCTCCCAACATCTAGAGGGATGTGTTTTCTCATAATACAATTCCTCAGTGTGG
CAGAAAAAAACATGTTTATTAACGGGCCAAAATATCTTTAGTCTCTCTGTAAAAACAGGA
AGCCAAAAGTATATAAACTTGAATTATTTATGTTCAGTAATTAATGTTTTAGTATTGTAT
CTTATTTATAAATGGTCTAGATATTTAATGCAAATCTTTTACTTAGCTTAACTTTAAGGT
TAAAAATTACCAAAAGTACTTTGGAAACTATTCTTAGGCAGATTTACTGTAAACAAATTA
""".split()


def fastq_files():
    import random
    r = random.Random(7)

    def recs(n, crlf=False):
        nl = "\r\n" if crlf else "\n"
        out = []
        for i in range(n):
            seq = "".join(r.choice("ACGTN") for _ in range(r.randrange(20, 120)))
            qual = "".join(r.choice("@IJ#+") for _ in seq)          # quality lines may start with '@'
            out.append("@read.%d extra words%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, qual, nl))
        return "".join(out)
    return {
        "plain7.fastq": recs(7).encode(),
        "junk_first.fastq": ("junk line before the first record\nmore junk\n" + recs(5)).encode(),
        "no_final_newline.fastq": recs(4)[:-1].encode(),
        "crlf.fastq": recs(3, crlf=True).encode(),
        "many.fastq": recs(450).encode(),
        "header_only.fastq": b"@only a header\n",
        "empty.fastq": b"",
    }


def main():
    ref = open(REF_PATH, "rb").read()
    out = []
    reads = (synth_reads(ref, 400, 100, 1001) + synth_reads(ref, 100, 100, 1002, n_frac=1.0) +
             synth_reads(ref, 100, 150, 1003, indel_frac=0.6) + [ref[:100], ref[-100:]])
    for i, q in enumerate(reads):
        out.append("align synth.%d %s" % (i, q.decode()))
        out.append("sga %s" % q.decode())
    for q in reads[:60]:
        for cut in (len(q), 61, 23, 9):
            out.append("seeds %s" % q[:cut].decode())
    rng = np.random.default_rng(1004)
    alpha = list(b"ACGT")
    for i in range(300):
        lq = int(rng.integers(4, 120))
        base = bytes(rng.choice(alpha, lq + 30).tolist())
        q = bytearray(base[8:8 + lq]); r = bytearray(base[:lq + int(rng.integers(0, 25))])
        for _ in range(int(rng.integers(0, 4))):
            q[int(rng.integers(0, lq))] = int(rng.choice(alpha))
        if i % 3 == 1:
            q[int(rng.integers(0, lq))] = ord("N")
        if i % 3 == 2:
            r[int(rng.integers(0, len(r)))] = int(rng.choice(list(b"NRYKMSWBDVH")))
        out.append("sw %s %s" % (q.decode(), r.decode()))
    # ---- round 3: the rows that were restated without a vector to check them against
    # all 33 query strings of the reference's own harness, bwt:test/0 (src/bwt.erl:24-57; inputs, not results)
    for i, q in enumerate(BWT_TEST_QUERIES):
        out.append("sga %s" % q)
        out.append("align bwt_test.%d %s" % (i, q))
    # the worker's pick among several alignments (src/worker_bwt.erl:99-106): every tie pattern of up to 4 candidates over three
    # score levels, and seeded longer lists (OTP's merge sort decides among equal scores: bwt_amd/csrc/otp_sort.h restates it)
    import itertools
    for k in (2, 3, 4):
        for combo in itertools.product((150, 170, 200), repeat=k):
            out.append("pick %s" % ",".join(str(s) for s in combo))
    for i in range(200):
        k = int(rng.integers(5, 41))
        out.append("pick %s" % ",".join(str(int(s)) for s in rng.choice([100, 120, 140, 160, 180, 200], k)))
    # fastq:read_seqs/2 (src/fastq.erl:5-40) on odd files: junk before the first record, quality lines that start with '@',
    # no final newline, CRLF, a header without a sequence, an empty file
    fq_dir = os.path.join(ROOT, "tests", "golden", "erlang", "fastq")
    os.makedirs(fq_dir, exist_ok=True)
    for name, data in fastq_files().items():
        with open(os.path.join(fq_dir, name), "wb") as f:
            f.write(data)
        # (one call on the file the reference crashes on: what a second call on the same device would see is no part of the contract)
        out.append("fastq %s %s" % (name, "3" if name == "header_only.fastq" else "3,200,1,5"))
    # seeds:generate_fs/3 + the generated fs:find_seeds/1 (src/seeds.erl:11-85)
    for i in range(24):
        length, defects = [(8, 0), (8, 1), (10, 1), (10, 2), (12, 2), (6, 3)][i % 6]
        st = int(rng.integers(0, len(ref) - 5000))
        q = ref[st + 1000:st + 1000 + length + defects + int(rng.integers(0, 4))]
        window = bytearray(ref[st:st + 3000])
        for _ in range(8):                                   # a few planted near-copies of the query
            p = int(rng.integers(0, len(window) - 40))
            window[p:p + len(q)] = q
            if defects:
                window[p + int(rng.integers(0, length))] = int(rng.choice(alpha))
        out.append("genfs %s %d %d %s" % (q.decode(), length, defects, bytes(window).decode()))
    out.append("genfs ACGTACG 8 1 ACGTACGTACGTACGT")        # query shorter than Len + Defects: the generated module does not compile
    # the master's printed line (src/master.erl:151-153)
    for i in range(12):
        lq = [100, 150, 36][i % 3]
        out.append("line read.%d/1 GL000193.1 %d %s %d %s" % (i, int(rng.integers(-150, 189000)), ["%dM" % lq, "50M2D%dM" % (lq - 50), "3M1I%dM" % (lq - 4)][i % 3],
                                                            int(rng.integers(100, 301)), bytes(rng.choice(list(b"ACGTN"), lq + i % 4).tolist()).decode()))
    # ---- round 4: sw:sw/2 on SQUARE windows whose diagonal has 3..10 mismatches -- what the certificate kernel (bwt_amd/csrc/sw_cert.h)
    # settles without filling the table: spread and clumped substitutions, read ends, periodic sequences (neighbouring diagonals match)
    rng4 = np.random.default_rng(4004)
    for i in range(240):
        lq = int(rng4.choice([100, 100, 150, 64, 33, 128]))
        if i % 4 == 3:
            unit = bytes(rng4.choice(alpha, int(rng4.integers(1, 8))).tolist())
            r = bytearray((unit * (lq // len(unit) + 2))[:lq])
            for _ in range(int(rng4.integers(0, 4))):
                r[int(rng4.integers(0, lq))] = int(rng4.choice(alpha))
        else:
            r = bytearray(rng4.choice(alpha, lq).tolist())
        q = bytearray(r)
        m = int(rng4.integers(3, 11))
        if i % 4 == 1:
            c0 = [0, lq - 13, int(rng4.integers(0, lq - 12))][i % 3]
            ps = c0 + rng4.choice(12, m, replace=False)
        else:
            ps = rng4.choice(lq, m, replace=False)
        for p_ in ps:
            q[int(p_)] = int(rng4.choice([x for x in alpha if x != q[int(p_)]]))
        out.append("sw %s %s" % (q.decode(), bytes(r).decode()))
    path = os.path.join(ROOT, "tests", "golden", "erlang", "capture_inputs.txt")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote %d records to %s" % (len(out), path))


if __name__ == "__main__":
    main()
