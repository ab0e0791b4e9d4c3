"""lists:sort/2 of Erlang/OTP stdlib (lists.erl), restated clause by clause.

Third-party code the reference depends on but does not vendor (OTP 17, README.md:19).  Needed
because worker_bwt.erl:104 sorts the candidate alignments with the comparator
`fun({{R1,_},_,_},{{R2,_},_,_}) -> R1 > R2 end`, which is not a total order: which of several
equal-score alignments ends up first is decided by the run-detecting merge sort's internals.
An Erlang deployment calls lists:sort/2 itself; this file exists so that the
Python host mirror reports the same head.  Restated from the published algorithm; not run
against an Erlang VM (none available) -- property-tested as a stable merge sort.
"""
import sys

sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))


def sort(fun, l):
    if len(l) < 2:
        return list(l)
    x, y, t = l[0], l[1], list(l[2:])
    if fun(x, y):
        return _fsplit_1(y, x, fun, t, [], [])
    return _fsplit_2(y, x, fun, t, [], [])


# ---- ascending runs
def _fsplit_1(y, x, fun, l, r, rs):
    while l:
        z, l = l[0], l[1:]
        if fun(y, z):
            y, x, r = z, y, [x] + r
        elif fun(x, z):
            x, r = z, [x] + r
        elif not r:
            r = [z]
        else:
            return _fsplit_1_1(y, x, fun, l, r, rs, z)
    return _rfmergel([[y, x] + r] + rs, [], fun, "asc")


def _fsplit_1_1(y, x, fun, l, r, rs, s):
    while l:
        z, l = l[0], l[1:]
        if fun(y, z):
            y, x, r = z, y, [x] + r
        elif fun(x, z):
            x, r = z, [x] + r
        elif fun(s, z):
            return _fsplit_1(z, s, fun, l, [], [[y, x] + r] + rs)
        else:
            return _fsplit_1(s, z, fun, l, [], [[y, x] + r] + rs)
    return _rfmergel([[s], [y, x] + r] + rs, [], fun, "asc")


# ---- descending runs
def _fsplit_2(y, x, fun, l, r, rs):
    while l:
        z, l = l[0], l[1:]
        if not fun(y, z):
            y, x, r = z, y, [x] + r
        elif not fun(x, z):
            x, r = z, [x] + r
        elif not r:
            r = [z]
        else:
            return _fsplit_2_1(y, x, fun, l, r, rs, z)
    return _fmergel([[y, x] + r] + rs, [], fun, "desc")


def _fsplit_2_1(y, x, fun, l, r, rs, s):
    while l:
        z, l = l[0], l[1:]
        if not fun(y, z):
            y, x, r = z, y, [x] + r
        elif not fun(x, z):
            x, r = z, [x] + r
        elif not fun(s, z):
            return _fsplit_2(z, s, fun, l, [], [[y, x] + r] + rs)
        else:
            return _fsplit_2(s, z, fun, l, [], [[y, x] + r] + rs)
    return _fmergel([[s], [y, x] + r] + rs, [], fun, "desc")


def _fmergel(ls, acc, fun, o):
    while len(ls) >= 2:
        if o == "asc":
            t1, h2t2 = ls[0], ls[1]
        else:
            h2t2, t1 = ls[0], ls[1]
        acc = [_fmerge2_1(t1, h2t2[0], fun, h2t2[1:], [])] + acc
        ls = ls[2:]
    if len(ls) == 1:
        if not acc:
            return ls[0]
        return _rfmergel([ls[0][::-1]] + acc, [], fun, o)
    return _rfmergel(acc, [], fun, o)


def _rfmergel(ls, acc, fun, o):
    while len(ls) >= 2:
        if o == "asc":
            h2t2, t1 = ls[0], ls[1]
        else:
            t1, h2t2 = ls[0], ls[1]
        acc = [_rfmerge2_1(t1, h2t2[0], fun, h2t2[1:], [])] + acc
        ls = ls[2:]
    if len(ls) == 1:
        return _fmergel([ls[0][::-1]] + acc, [], fun, o)
    return _fmergel(acc, [], fun, o)


# lists:reverse(A, B) = reversed(A) ++ B
def _fmerge2_1(t1, h2, fun, t2, m):
    if not t1:
        return t2[::-1] + [h2] + m
    h1, t1 = t1[0], t1[1:]
    if fun(h1, h2):
        return _fmerge2_1(t1, h2, fun, t2, [h1] + m)
    return _fmerge2_2(h1, t1, fun, t2, [h2] + m)


def _fmerge2_2(h1, t1, fun, t2, m):
    if not t2:
        return t1[::-1] + [h1] + m
    h2, t2 = t2[0], t2[1:]
    if fun(h1, h2):
        return _fmerge2_1(t1, h2, fun, t2, [h1] + m)
    return _fmerge2_2(h1, t1, fun, t2, [h2] + m)


def _rfmerge2_1(t1, h2, fun, t2, m):
    if not t1:
        return t2[::-1] + [h2] + m
    h1, t1 = t1[0], t1[1:]
    if fun(h1, h2):
        return _rfmerge2_2(h1, t1, fun, t2, [h2] + m)
    return _rfmerge2_1(t1, h2, fun, t2, [h1] + m)


def _rfmerge2_2(h1, t1, fun, t2, m):
    if not t2:
        return t1[::-1] + [h1] + m
    h2, t2 = t2[0], t2[1:]
    if fun(h1, h2):
        return _rfmerge2_2(h1, t1, fun, t2, [h2] + m)
    return _rfmerge2_1(t1, h2, fun, t2, [h1] + m)
