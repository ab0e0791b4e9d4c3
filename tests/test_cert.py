"""bwt_amd/csrc/sw_cert.h on the CPU: the header k_sw_cert compiles is built into tests/cert_check/cert_check.cpp with g++ and run
against the oracle's full Smith-Waterman table (oracle/bwt_oracle.c: orc_sw, src/sw.erl:47-160) and an exact affine DP:
whatever the certificate settles is the oracle's answer, and its per-diagonal bounds are upper bounds, on instances built to
break it (clumped and adjacent mismatches, read ends, periodic and two-letter sequences, 18..160 bases)."""
import json
import os
import subprocess

from conftest import ROOT


def build(tmp_path):
    exe = str(tmp_path / "cert_check")
    orc_o = str(tmp_path / "orc.o")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-pthread", "-c", os.path.join(ROOT, "oracle", "bwt_oracle.c"), "-o", orc_o])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "cert_check", "cert_check.cpp"),
                           orc_o, "-lpthread", "-lm", "-o", exe])
    return exe


def test_certificate_against_the_full_table(tmp_path):
    exe = build(tmp_path)
    out = json.loads(subprocess.check_output([exe, "120000", "20261005"]))
    for name, r in out.items():
        assert r["bound_checks"] > 5000 and r["certified"] > 3000, (name, r)
    # ordinary candidates (i.i.d. sequence, random places): the certificate settles nine in ten of the plain-diagonal answers
    plain = json.loads(subprocess.check_output([exe, "40000", "7", "1"]))
    r = plain["X5_NW4"]
    assert r["certified"] > 0.9 * (r["certified"] + r["plain_but_not_certified"]), r
    r = plain["X9_NW4"]                                  # seven to ten substitutions: four in five
    assert r["certified"] > 0.8 * (r["certified"] + r["plain_but_not_certified"]), r
