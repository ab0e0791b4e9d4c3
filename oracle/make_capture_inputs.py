#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/erlang/capture_inputs.txt -- the seeded inputs on which a machine
that HAS Erlang/OTP should run the real reference (oracle/capture_golden.escript) to pin the oracle.
One record per line:  sga <read>  |  seeds <sub-read>  |  sw <query> <window>  |  align <name> <read>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import REF_PATH, synth_reads          # noqa: E402
import numpy as np                                   # noqa: E402


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
    path = os.path.join(ROOT, "tests", "golden", "erlang", "capture_inputs.txt")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote %d records to %s" % (len(out), path))


if __name__ == "__main__":
    main()
