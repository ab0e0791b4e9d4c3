"""seeds.erl: the oracle's literal restatement against hand-derived clause sets, the generator text
of libbwtgpu (host-only entry point) against the oracle's clause list."""
import numpy as np
import pytest

from oracle import erl_literal as E, orc


def P(*items):
    return [None if c == "_" else ord(c) for c in items]


def test_zero_defects_is_the_prefix():
    assert E.seeds_patterns(b"ACGTAC", 4, 0) == [P("A", "C", "G", "T")]


def test_one_defect_patterns_len3():
    # query ACGT.., Len 3, one edit.  DFS order: mutation, insertion, deletion, match (seeds.erl:58-61)
    got = E.seeds_patterns(b"ACGT", 3, 1)
    exp = [
        P("_", "C", "G"),        # mutate base 1
        P("_", "A", "C"),        # insert before base 1
        P("C", "G", "T"),        # delete base 1
        P("A", "_", "G"),        # match A, mutate base 2
        P("A", "_", "C"),        # match A, insert
        P("A", "G", "T"),        # match A, delete C
        P("A", "C", "_"),        # match AC, mutate G
        P("A", "C", "_"),        # match AC, insert
        P("A", "C", "T"),        # match AC, delete G
    ]
    assert got == exp


def test_generated_scan_jumps_len_after_a_hit():
    pats = E.seeds_patterns(b"ACGT", 3, 0)
    assert E.fs_find_seeds(pats, b"TTACGACGTT") == [6, 3]          # newest first
    assert E.fs_find_seeds(pats, b"ACGACG") == [4, 1]              # the last window: the exactly-Len clause
    assert E.fs_find_seeds(E.seeds_patterns(b"AAAA", 2, 0), b"AAAAA") == [3, 1]   # overlapping hits are skipped
    assert E.fs_find_seeds(pats, b"AC") == []


def test_c_oracle_equals_literal_python():
    rng = np.random.default_rng(3)
    for _ in range(200):
        L = int(rng.integers(2, 9)); D = int(rng.integers(0, 4))
        q = bytes(rng.choice(list(b"ACGT"), L + D + int(rng.integers(0, 3))).tolist())
        a = [[None if x == 256 else int(x) for x in row] for row in orc.fs_patterns(q, L, D).tolist()]
        b = E.seeds_patterns(q, L, D)
        assert a == b
        ref = bytes(rng.choice(list(b"ACGTN"), int(rng.integers(0, 80))).tolist())
        assert orc.fs_find_seeds(orc.fs_patterns(q, L, D), ref) == E.fs_find_seeds(b, ref)


def test_short_query_does_not_compile():
    with pytest.raises(E.ErlCrash):
        E.seeds_patterns(b"ACG", 3, 1)
    with pytest.raises(ValueError):
        orc.fs_patterns(b"ACG", 3, 1)


def clause_text(pattern):
    return "matches(%s) -> true;\n" % ",".join("_" if c is None else "$" + chr(c) for c in pattern)


@pytest.mark.parametrize("q,L,D", [(b"ACGT", 3, 1), (b"ACGTACGTACGTAC", 10, 2), (b"GATTACAGATTACA", 12, 0), (b"ACGTTGCA", 4, 3)])
def test_library_generator_text(q, L, D):
    """bwt_seeds_generate_fs (host only, no GPU): header and scan clauses as the generator's io:format strings print
    them (src/seeds.erl:14-41), then one clause per pattern of the oracle, then the catch-all."""
    from bwt_amd import lib
    txt = lib.seeds_generate_fs(q, L, D).decode("latin1")
    args = ",".join("X%d" % j for j in range(1, L + 1))
    head = ("% Seeds finding algo\n%\n% Cloudozer(c), 2014\n\n\n-module(fs).\n-export([matches/" + str(L) + ", find_seeds/1]).\n"
            "\n\nfind_seeds(Ref_Seq) ->\n\tfind_seeds([],1, Ref_Seq).\n\nfind_seeds(Acc,N,[" + args + "]) ->\n\tcase matches(" + args +
            ") of\n\t\ttrue -> [N|Acc];\n\t\t_ -> Acc\n\tend;\n"
            "\nfind_seeds(Acc,N,[" + args + "|Ref_Seq]) ->\n\tcase matches(" + args + ") of\n\t\ttrue -> find_seeds([N|Acc],N+length([" +
            args + "]),Ref_Seq);\n\t\t_ -> find_seeds(Acc,N+1,[" + args[3:] + "|Ref_Seq])\n\tend;\n\nfind_seeds(Acc,_,_) -> Acc.\n\n\n")
    body = "".join(clause_text(p) for p in E.seeds_patterns(q, L, D))
    assert txt == head + body + "matches(%s) -> false.\n" % ",".join("_" * L)


def test_library_generator_rejects_what_the_reference_cannot_generate():
    from bwt_amd import lib
    for q, L, D in [(b"ACG", 3, 1), (b"ACGT", 1, 0), (b"ACGT", 0, 0)]:
        with pytest.raises(lib.BwtError) as e:
            lib.seeds_generate_fs(q, L, D)
        assert e.value.code == -22
