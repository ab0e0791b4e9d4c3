"""seeds.erl mirror: hand-derived pattern sets for small cases."""
from bwt_amd import seeds


def P(*items):
    return [None if c == "_" else ord(c) for c in items]


def test_zero_defects_is_the_prefix():
    assert seeds.patterns(b"ACGTAC", 4, 0) == [P("A", "C", "G", "T")]


def test_one_defect_patterns_len3():
    # query ACGT.., Len 3, one edit.  DFS order: mutation, insertion, deletion, match (seeds.erl:58-61)
    got = seeds.patterns(b"ACGT", 3, 1)
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


def test_generated_text_and_matcher():
    txt = seeds.generate_fs(b"ACGT", 3, 1)
    assert "-export([matches/3, find_seeds/1])." in txt
    assert "matches(_,$C,$G) -> true;" in txt and txt.rstrip().endswith("matches(_,_,_) -> false.")
    pats = seeds.patterns(b"ACGT", 3, 0)
    assert seeds.find_seeds(pats, b"TTACGACGTT") == [6, 3]          # newest first; jump Len after a hit
