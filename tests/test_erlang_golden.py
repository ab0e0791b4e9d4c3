"""If golden vectors captured from the REAL Erlang reference exist (tests/golden/erlang/erlang_vectors.txt, written
by oracle/capture_golden.escript on a machine that has OTP), the oracle is checked against them line by line.
Until someone captures them the test is skipped and parity stays "unpinned" (DESIGN.md)."""
import os
import re

import pytest

from conftest import GOLDEN

IN = os.path.join(GOLDEN, "erlang", "capture_inputs.txt")
OUT = os.path.join(GOLDEN, "erlang", "erlang_vectors.txt")


def test_capture_inputs_are_committed_and_reproducible():
    import subprocess, sys, hashlib
    from conftest import ROOT
    before = hashlib.sha256(open(IN, "rb").read()).hexdigest()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_capture_inputs.py")], stdout=subprocess.DEVNULL)
    assert hashlib.sha256(open(IN, "rb").read()).hexdigest() == before


@pytest.mark.skipif(not os.path.exists(OUT), reason="no Erlang-captured golden vectors yet (run oracle/capture_golden.escript)")
def test_oracle_matches_erlang_vectors(orc_fm, ref_bytes):
    from oracle import orc, erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    ins = open(IN).read().splitlines()
    outs = open(OUT).read().splitlines()
    assert len(ins) == len(outs)

    def pairs(text):
        return [(int(a), int(b)) for a, b in re.findall(r"\{(-?\d+),(-?\d+)\}", text)]
    for i, o in zip(ins, outs):
        f = i.split(" ")
        if f[0] == "sga":
            try:
                exp = orc.sga(orc_fm, f[1].encode())
            except RuntimeError:
                assert o == "sga crash"
                continue
            assert pairs(o) == exp, i
        elif f[0] == "seeds":
            st, seeds, *_ = orc.find_seeds(orc_fm, f[1].encode())
            want = {1: "seeds no_seeds", 2: "seeds too_many_seeds", -1: "seeds crash"}.get(st)
            if want:
                assert o == want, i
            else:
                assert [int(x) for x in re.findall(r"-?\d+", o)] == seeds, i
        elif f[0] == "sw":
            exp = orc.sw(f[1].encode(), f[2].encode())
            if exp == "no_match":
                assert o == "sw no_match", i
            else:
                assert o == 'sw {%d,"%s"}' % exp, i
        elif f[0] == "align":
            try:
                exp = orc.align_read(orc_fm, ref1, f[2].encode())
            except RuntimeError:
                assert o.endswith(" crash")
                continue
            if exp is None:
                assert o.endswith(" nothing"), i
            else:
                got = [(int(s), c, int(p)) for s, c, p in re.findall(r'\{\{(-?\d+),"([^"]*)"\},(-?\d+)\}', o)]
                assert got == [(s, c, p) for (s, c, p, _, _) in exp], i
