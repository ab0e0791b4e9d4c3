"""TEST INFRASTRUCTURE ONLY (oracle/): literal, list-for-list CPU restatement of the
reference's seed-search / clustering / Smith-Waterman hot path.

PARITY UNPINNED: the reference is Erlang, no Erlang runtime exists in the authoring
container, and the reference ships no test that records a result for this path
(SURVEY.md section 8c).  This file is a careful clause-by-clause re-statement of the
Erlang source; it was never compared with a running Erlang VM.

Every function names the reference lines it follows (paths relative to
/root/reference).  Python was chosen for the literal form because its int/float
semantics coincide with Erlang's where this path depends on them:
  * ints are arbitrary precision, floats are IEEE doubles, int+float -> float;
  * 2 == 2.0 compares equal, tuple comparison is element-wise  (Erlang term order
    for {number, atom} tuples; atoms left < up < up_left are kept as the strings
    'left' < 'up' < 'up_left', which sort the same way);
  * max() keeps the first of equal elements (lists:max/1, erlang:max/2).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

# include/bwt.hrl:4-7,14-18
TOLERANCE = 20
MIN_LEN = 8
MAX_RANGE = 45
MAX_LEN = 15
MATCH = 2
UNKNOWN = 1.6
MISMATCH = -1
GAP_PENALTY = -2
GAP_EXT_PENALTY = -1
REF_EXTENSION_LEN = 200  # src/worker_bwt.erl:17

NO_SEEDS = "no_seeds"
TOO_MANY = "too_many_seeds"
NO_MATCH = "no_match"


class ErlCrash(Exception):
    """The reference would crash here (case_clause / badarg / badmatch / function_clause)."""


class FM:
    """The in-memory FM tuple: rows {F,L,I,SA}, 1-based (src/bwt.erl:84-120)."""

    def __init__(self, d):
        self.F = d["F"]
        self.L = d["L"]
        self.I = d["I"]
        self.SA = d["SA"]
        self.Pc, self.Pg, self.Pt, self.Last = d["pointers"]
        self.shift = d["shift"]

    def row(self, r):
        # element(R, FM): badarg outside 1..size
        if r < 1 or r > len(self.L):
            raise ErlCrash("element(%d, FM)" % r)
        return self.F[r - 1], self.L[r - 1], int(self.I[r - 1]), int(self.SA[r - 1])


# ---------------------------------------------------------------- src/fmi.erl:25-39
def se_down(sp, ep, c2, fm):
    while True:
        if sp == ep:                      # fmi.erl:25
            return None                   # not_found
        _, l, p, _ = fm.row(sp)           # fmi.erl:28
        if l == c2:
            return p
        sp += 1


def se_up(ep, c2, fm):
    while True:
        _, l, p, _ = fm.row(ep)           # fmi.erl:36
        if l == c2:
            return p
        ep -= 1


# ---------------------------------------------------------------- src/fmi.erl:89-100
def get_seed_ends(fm, sp, ep, n):
    if n < MIN_LEN:
        return NO_SEEDS
    acc = []
    while True:
        if sp == ep:
            acc.insert(0, fm.row(ep)[3] + n)
            return acc
        acc.insert(0, fm.row(sp)[3] + n)
        sp += 1


# ---------------------------------------------------------------- src/bwa.erl:12-45
def find_seeds(fm, subseq):
    """bwa:find_seeds/6.  subseq: bytes.  -> NO_SEEDS | TOO_MANY | [int] (descending row order)."""
    rev = subseq[::-1]
    h, tail = rev[0], rev[1:]
    if h == ord("A") or h == ord("N"):    # bwa.erl:16,21-23 (N handled as A)
        sp, ep = 1, fm.Pc - 1
    elif h == ord("C"):
        sp, ep = fm.Pc, fm.Pg - 1
    elif h == ord("G"):
        sp, ep = fm.Pg, fm.Pt - 1
    elif h == ord("T"):
        sp, ep = fm.Pt, fm.Last
    else:
        raise ErlCrash("case_clause %r" % chr(h))
    return _find_seeds5(fm, sp, ep, tail, 1)


def _find_seeds5(fm, sp, ep, q, n):
    while True:
        if n == MAX_LEN:                                   # bwa.erl:31-32
            if ep - sp > MAX_RANGE:
                return TOO_MANY
            return get_seed_ends(fm, sp, ep, MAX_LEN)
        if len(q) > 0 and q[0] == ord("N"):                # bwa.erl:33-34
            if ep - sp > MAX_RANGE:
                return TOO_MANY
            return get_seed_ends(fm, sp, ep, n)
        if len(q) > 0:                                     # bwa.erl:36-43
            c2 = q[0]
            sp1 = se_down(sp, ep, c2, fm)
            if sp1 is None:
                return get_seed_ends(fm, sp, ep, n)        # no range check here
            ep1 = se_up(ep, c2, fm)
            sp, ep, q, n = sp1, ep1, q[1:], n + 1
            continue
        if ep - sp > MAX_RANGE:                            # bwa.erl:44-45
            return TOO_MANY
        return get_seed_ends(fm, sp, ep, n)


# ---------------------------------------------------------------- src/sga.erl:69-76
def skip_Ns(q):
    if len(q) < MIN_LEN:
        return None                                        # no_more_subseqs
    rev = q[::-1]
    j, skipped, k = 0, 0, 0
    while True:
        if j == MIN_LEN:                                   # sga.erl:72-73 (before the list is examined)
            if skipped == 0:
                return 0, q
            return skipped, q[:len(q) - skipped]
        if k == len(rev):                                  # sga.erl:76
            return None
        if rev[k] == ord("N"):                             # sga.erl:74
            skipped, j = skipped + j, 1
        else:                                              # sga.erl:75
            j += 1
        k += 1


# ---------------------------------------------------------------- src/sga.erl:17-39
def sga_raw(fm, q):
    """The sga/9 loop up to (not including) get_similar: -> (Qty, Acc list)."""
    acc, qty, shift = [], 0, 0
    while True:
        r = skip_Ns(q)
        if r is None:
            return qty, acc
        skip, q1 = r
        qlen = len(q1)
        res = find_seeds(fm, q1)
        nxt = q1[:qlen - MIN_LEN]                          # lists:sublist(Qseq1,Qlen-?MIN_LEN)
        if res == NO_SEEDS or res == TOO_MANY:
            q, shift = nxt, shift + MIN_LEN + skip
        else:
            for s in res:                                  # add_seeds: foldl, prepend
                acc.insert(0, s + shift + skip)
            q, qty, shift = nxt, qty + 1, shift + MIN_LEN + skip


# ---------------------------------------------------------------- src/sga.erl:42-64
def get_similar(qty, ls):
    if qty == 0 or qty == 1:
        return []
    srt = sorted(ls)
    p1, rest = srt[0], srt[1:]
    n = 2 + (qty // 7)
    count, dist, acc = 1, 0, []
    for p2 in rest:
        if p2 - p1 <= TOLERANCE:
            count, dist, p1 = count + 1, dist + p2 - p1, p2
        elif count >= n:
            acc.insert(0, (p1, dist))
            p1, count, dist = p2, 1, 0
        else:
            p1, count, dist = p2, 1, 0
    if count >= n:
        acc.insert(0, (p1, dist))
    return acc


def sga(fm, q):
    """sga:sga/6 (src/sga.erl:17) -> [(S, D)] in descending position order."""
    qty, acc = sga_raw(fm, q)
    return get_similar(qty, acc)


# ---------------------------------------------------------------- src/sw.erl:164-202
def sigma(s, r):
    """First arg = read base, second = reference base (bytes as ints)."""
    c = chr
    if s == ord("N") and r == ord("N"):
        return UNKNOWN
    if s == r:
        return MATCH
    if s == ord("N"):
        return UNKNOWN
    if r == ord("N"):
        return UNKNOWN
    s, r = c(s), c(r)
    if r == "R":
        return MATCH if s in "AG" else MISMATCH
    if r == "Y":
        return MATCH if s in "CT" else MISMATCH
    if r == "B":
        return MISMATCH if s == "A" else MATCH
    if r == "D":
        return MISMATCH if s == "C" else MATCH
    if r == "K":
        return MATCH if s in "GT" else MISMATCH
    if r == "M":
        return MATCH if s in "AC" else MISMATCH
    if r == "S":
        return MATCH if s in "GC" else MISMATCH
    if r == "W":
        return MATCH if s in "AT" else MISMATCH
    if r == "V":
        return MISMATCH if s == "T" else MATCH
    return MISMATCH


# ---------------------------------------------------------------- src/sw.erl:47-107
def _get_Lvalue(v, d):
    return v + GAP_EXT_PENALTY if d == "left" else v + GAP_PENALTY


def _get_Uvalue(v, d):
    return v + GAP_EXT_PENALTY if d == "up" else v + GAP_PENALTY


def _add_row2tab(tab, ref, sq, acct, vmax):
    # tab: remaining previous columns (newest row at head); acct: finished columns, newest first
    for col, sr in zip(tab, ref):
        vup, dup = col[0]
        (vleft, dleft), (vul, _) = acct[0][0], acct[0][1]
        v, d = max([                                       # lists:max/1 on {V,Dir} tuples
            (vul + sigma(sq, sr), "up_left"),
            (_get_Lvalue(vleft, dleft), "left"),
            (_get_Uvalue(vup, dup), "up"),
        ])
        acct.insert(0, [(v, d)] + col)
        vmax = max(vmax, v)                                # erlang:max/2 keeps the first on ==
    if len(tab) != len(ref):
        raise ErlCrash("function_clause add_row2tab")
    return vmax, acct[::-1]


def sw_table(qseq, ref):
    """sw:sw/3 up to the callback: -> NO_MATCH | (Tab, Vmax).  Tab = list of columns,
    each a list with the newest row first, ending in (0,'undef')."""
    lq = len(qseq)
    tab = [[(0, "undef")] for _ in range(len(ref) + 1)]
    vthr = -MATCH * (lq // 2)
    v1max = 0
    qi = 0
    while True:
        if v1max < vthr:                                   # sw.erl:69
            return NO_MATCH
        if qi == lq:                                       # sw.erl:80
            return tab, v1max
        s = qseq[qi]
        v, d = tab[0][0]
        firstcol = tab[0][1:]
        v1 = v + GAP_PENALTY if not firstcol else v + GAP_EXT_PENALTY
        v1max, tab = _add_row2tab(tab[1:], ref, s, [[(v1, "up"), (v, d)] + firstcol], v1)
        vthr += MATCH
        qi += 1


# ---------------------------------------------------------------- src/sw.erl:110-160
def _exact_eq(a, b):
    # =:= : 2 =:= 2.0 is false
    return type(a) is type(b) and a == b


def get_CIGAR(tab, vmax):
    # get_starting_point/2,3
    acc = []
    k = 0
    while True:
        if k == len(tab):
            raise ErlCrash("function_clause get_starting_point")
        col = tab[k]
        if _exact_eq(col[0][0], vmax):
            tab1 = [col] + acc
            break
        acc.insert(0, col)
        k += 1
    (_, d0), column = tab1[0][0], tab1[0][1:]
    tab2 = tab1[1:]
    cigar = ""

    def get_str(n, d):
        return str(n) + {"up_left": "M", "left": "D", "up": "I"}[d]

    # state machine of get_CIGAR/5 and /6; `coltail` None means "fetch next column" (arity 5)
    if d0 == "up":
        coltail, skip, d, n = column, 1, d0, 1
    else:
        coltail, skip, d, n = None, 1, d0, 1
    while True:
        if coltail is None:                                # get_CIGAR/5: sw.erl:129-132
            if not tab2:
                raise ErlCrash("function_clause get_CIGAR/5")
            column, tab2 = tab2[0], tab2[1:]
            if skip > len(column):
                raise ErlCrash("nthtail")
            coltail = column[skip:]
        if len(coltail) == 1 and coltail[0][1] == "undef":  # sw.erl:135
            return get_str(n, d) + cigar
        if not coltail:
            raise ErlCrash("function_clause get_CIGAR/6")
        d2 = coltail[0][1]
        if d2 == "up":
            if d == "up":                                  # sw.erl:137-138
                coltail, skip, n = coltail[1:], skip + 1, n + 1
            else:                                          # sw.erl:139-140
                cigar = get_str(n, d) + cigar
                coltail, skip, d, n = coltail[1:], skip + 1, "up", 1
        elif d2 == d:                                      # sw.erl:141-145
            if d == "up_left":
                coltail, skip, n = None, skip + 1, n + 1
            elif d == "left":
                coltail, n = None, n + 1
            else:
                raise ErlCrash("case_clause")
        else:                                              # sw.erl:147-151
            cigar = get_str(n, d) + cigar
            if d2 == "up_left":
                coltail, skip, d, n = None, skip + 1, d2, 1
            elif d2 == "left":
                coltail, d, n = None, d2, 1
            else:
                raise ErlCrash("case_clause")


def sw(qseq, ref):
    """sw:sw/2 (src/sw.erl:47-49) for two non-empty byte strings -> NO_MATCH | (Score, Cigar)."""
    if len(qseq) == 0 or len(ref) == 0:
        raise ErlCrash("sw/2 with an empty list")
    r = sw_table(qseq, ref)
    if r == NO_MATCH:
        return NO_MATCH
    tab, vmax = r
    return math.trunc(vmax), _cigar_of(tab, vmax)


def _cigar_of(tab, vmax):
    return get_CIGAR(tab, vmax)


# ---------------------------------------------------------------- src/worker_bwt.erl:54-55,73-106
def make_ref1(ref):
    ext = b"N" * REF_EXTENSION_LEN
    return ext + ref + ext


def worker_candidates(fm, ref1, qseq):
    """The inner foldl of worker_loop(running,...) for one read (src/worker_bwt.erl:84-97).
    -> None when sga returns [] (the read emits nothing, :77), else the `Cigars` list
       [((Score,Cigar), Pos, RefSeq)] in the reference's list order."""
    seeds = sga(fm, qseq)
    if not seeds:
        return None
    cigars = []
    for s, d in seeds:
        ref_len = len(qseq) + d
        start = s - ref_len + REF_EXTENSION_LEN
        if start < 0 or start + ref_len > len(ref1):
            raise ErlCrash("badmatch window")               # worker_bwt.erl:89
        win = ref1[start:start + ref_len]
        r = sw(qseq, win)
        if r != NO_MATCH:
            cigars.insert(0, (r, s - ref_len + fm.shift, win))
    return cigars


# ---------------------------------------------------------------- OTP stdlib lists:sort/2
# Third-party arithmetic: Erlang/OTP 17 stdlib `lists.erl` (not vendored in the reference).
# Restated from the published algorithm (run-detecting merge sort: fsplit_*/fmergel/rfmergel).
# Used only for the worker's final pick (src/worker_bwt.erl:104) where the comparator
# `R1 > R2` is not a total order, so OTP's merge order decides ties.
def _fmerge2_1(t1, h2, fun, t2, m):
    # elements from the first list are prioritised
    t1 = list(t1); t2 = list(t2)
    while True:
        if not t1:
            return t2[::-1] + [h2] + m
        h1 = t1[0]
        if fun(h1, h2):
            m = [h1] + m; t1 = t1[1:]
            continue
        # fmerge2_2
        m = [h2] + m
        t1 = t1[1:]
        while True:
            if not t2:
                return t1[::-1] + [h1] + m
            h2 = t2[0]; t2 = t2[1:]
            if fun(h1, h2):
                m = [h1] + m
                break
            m = [h2] + m


def _rfmerge2_1(t1, h2, fun, t2, m):
    t1 = list(t1); t2 = list(t2)
    while True:
        if not t1:
            return t2[::-1] + [h2] + m
        h1 = t1[0]
        if not fun(h1, h2):
            m = [h1] + m; t1 = t1[1:]
            continue
        # rfmerge2_2
        m = [h2] + m
        t1 = t1[1:]
        while True:
            if not t2:
                return t1[::-1] + [h1] + m
            h2 = t2[0]; t2 = t2[1:]
            if not fun(h1, h2):
                m = [h1] + m
                break
            m = [h2] + m


def _fmergel(ls, acc, fun, o):
    while True:
        if len(ls) >= 2:
            if o == "asc":
                t1, (h2, *t2) = ls[0], ls[1]
            else:
                (h2, *t2), t1 = ls[0], ls[1]
            acc = [_fmerge2_1(t1, h2, fun, t2, [])] + acc
            ls = ls[2:]
            continue
        if len(ls) == 1:
            if not acc:
                return ls[0]
            return _rfmergel([ls[0][::-1]] + acc, [], fun, o)
        return _rfmergel(acc, [], fun, o)


def _rfmergel(ls, acc, fun, o):
    while True:
        if len(ls) >= 2:
            if o == "asc":
                (h2, *t2), t1 = ls[0], ls[1]
            else:
                t1, (h2, *t2) = ls[0], ls[1]
            acc = [_rfmerge2_1(t1, h2, fun, t2, [])] + acc
            ls = ls[2:]
            continue
        if len(ls) == 1:
            return _fmergel([ls[0][::-1]] + acc, [], fun, o)
        return _fmergel(acc, [], fun, o)


def _fsplit_1(y, x, fun, l, r, rs):
    s_pending = None  # fsplit_1_1 carries S
    l = list(l)
    while True:
        if not l:
            if s_pending is None:
                return _rfmergel([[y, x] + r] + rs, [], fun, "asc")
            return _rfmergel([[s_pending], [y, x] + r] + rs, [], fun, "asc")
        z, l = l[0], l[1:]
        if fun(y, z):
            x, y, r = y, z, [x] + r
        elif fun(x, z):
            x, r = z, [x] + r
        elif s_pending is None and not r:
            r = [z]
        elif s_pending is None:
            s_pending = z
        else:
            s = s_pending
            rs = [[y, x] + r] + rs
            r = []
            s_pending = None
            if fun(s, z):
                y, x = z, s
            else:
                y, x = s, z


def _fsplit_2(y, x, fun, l, r, rs):
    s_pending = None
    l = list(l)
    while True:
        if not l:
            if s_pending is None:
                return _fmergel([[y, x] + r] + rs, [], fun, "desc")
            return _fmergel([[s_pending], [y, x] + r] + rs, [], fun, "desc")
        z, l = l[0], l[1:]
        if not fun(y, z):
            x, y, r = y, z, [x] + r
        elif not fun(x, z):
            x, r = z, [x] + r
        elif s_pending is None and not r:
            r = [z]
        elif s_pending is None:
            s_pending = z
        else:
            s = s_pending
            rs = [[y, x] + r] + rs
            r = []
            s_pending = None
            if not fun(s, z):
                y, x = z, s
            else:
                y, x = s, z


def lists_sort2(fun, l):
    if len(l) < 2:
        return list(l)
    x, y, t = l[0], l[1], l[2:]
    if fun(x, y):
        return _fsplit_1(y, x, fun, t, [], [])
    return _fsplit_2(y, x, fun, t, [], [])


def worker_pick(cigars):
    """src/worker_bwt.erl:99-106: -> 'no_cigar' | ((Score,Cigar), Pos, RefSeq)."""
    if not cigars:
        return "no_cigar"
    if len(cigars) == 1:
        return cigars[0]
    return lists_sort2(lambda a, b: a[0][0] > b[0][0], cigars)[0]


# ---------------------------------------------------------------- offline index construction
# src/fm_index.erl:93-141, src/bwt.erl:84-120, src/dsa.erl:15-48, src/fmi.erl:43-76
class AS183:
    """Erlang `random` (OTP 17 stdlib, not vendored): uniform/0,1 with the default seed."""

    def __init__(self, state=(3172, 9814, 20125)):
        self.a1, self.a2, self.a3 = state

    def uniform(self, n=None):
        self.a1 = (self.a1 * 171) % 30269
        self.a2 = (self.a2 * 172) % 30307
        self.a3 = (self.a3 * 170) % 30323
        r = self.a1 / 30269 + self.a2 / 30307 + self.a3 / 30323
        r = r - math.trunc(r)
        return r if n is None else math.trunc(r * n) + 1


_IUPAC_MAP = {ord("B"): "C", ord("D"): "G", ord("R"): "A", ord("Y"): "C", ord("K"): "T", ord("M"): "A",
              ord("S"): "C", ord("W"): "A", ord("V"): "A", ord("A"): "A", ord("C"): "C", ord("G"): "G", ord("T"): "T"}


def ref_seq_of_chunk(chunk, rng):
    """fm_index.erl:131-137"""
    out = bytearray()
    for c in chunk:
        if c == ord("N"):
            out.append(b"ACGT"[rng.uniform(4) - 1])
        elif c in _IUPAC_MAP:
            out.append(ord(_IUPAC_MAP[c]))
        else:
            raise ErlCrash("error(%r)" % c)
    return bytes(out)


def fm_rows(ref_seq):
    """bwt:fm(st:append($$, Ref_seq)) -> dict of columns F, L, I, SA (rows in tuple order) + pointers.
    The suffix order is plain lexicographic with '$' smallest (dsa.erl:32-35: Vx < Vy on bytes,
    '$' = 36 < 'A'); computed here by numpy prefix doubling, independent of the GPU builder."""
    import numpy as np
    t = np.frombuffer(ref_seq + b"$", np.uint8).astype(np.int64)
    n = len(t)
    rank = np.unique(t, return_inverse=True)[1].astype(np.int64)
    h = 1
    while True:
        second = np.full(n, -1, np.int64)
        second[: n - h] = rank[h:]
        order = np.lexsort((second, rank))
        key = rank[order] * (n + 1) + (second[order] + 1)
        newr = np.concatenate([[0], np.cumsum(key[1:] != key[:-1])])
        rank = np.empty(n, np.int64)
        rank[order] = newr
        if newr[-1] == n - 1:
            break
        h *= 2
    sa = order
    F = t[sa].astype(np.uint8)
    L = t[sa - 1].astype(np.uint8)                     # SA = 0 -> t[-1] = '$'
    # I (bwt.erl:105-112, fmi.erl:43-60): the K-th row (0-based) whose F is c is popped by the K-th row whose L is c
    I = np.empty(n, np.int64)
    for c in b"$ACGT":
        I[L == c] = np.nonzero(F == c)[0]
    cnt = {c: int((F == c).sum()) for c in b"ACGT"}
    pc = 2 + cnt[ord("A")]; pg = pc + cnt[ord("C")]; pt = pg + cnt[ord("G")]
    return dict(F=F, L=L, I=I, SA=sa.astype(np.int64), pointers=(pc, pg, pt, n))


def ltrim_N(ls):
    j = k = 0
    while k < len(ls) and ls[k] in (ord("N"), 10, 13):
        j += ls[k] == ord("N")
        k += 1
    return j, ls[k:]


def make_index_chunks(chromo_bytes):
    """build_fm/4,5 (fm_index.erl:93-124) for the raw bytes of one contig (between two FASTA headers):
    -> [(J, Meta shift, chunk bytes)]"""
    _, c1 = ltrim_N(chromo_bytes[::-1])
    shift, c2 = ltrim_N(c1[::-1])
    body = bytes(c for c in c2 if c not in (10, 13))
    nbr = 1 + ((len(body) + 2) >> 24)
    out = []
    for j in range(1, nbr + 1):
        chunk_size = (len(body) + nbr * 2) // nbr
        start = 1 + (j - 1) * (chunk_size - 2)
        out.append((j, shift + start, body[start - 1:start - 1 + chunk_size]))
    return out


# ---------------------------------------------------------------- src/fastq.erl:5-40, src/master.erl:149-178
class LineDev:
    """file:open(Path, []) + file:read_line/1 on a byte string: lines keep their newline."""

    def __init__(self, data):
        self.data, self.pos = data, 0

    def read_line(self):
        if self.pos >= len(self.data):
            return None
        e = self.data.find(b"\n", self.pos)
        e = len(self.data) if e < 0 else e + 1
        line = self.data[self.pos:e]
        self.pos = e
        return line


def fastq_read_seq(dev):
    while True:
        line = dev.read_line()
        if line is None:
            return None                                        # eof
        if line[:1] == b"@":
            seq = dev.read_line()
            if seq is None:
                raise ErlCrash("badmatch {ok,SeqData} = eof")
            dev.read_line(); dev.read_line()                   # _Delimiter, _Quality
            return line[1:][:-1], seq[:-1]                     # lists:droplast/1 on both


def fastq_read_seqs(dev, n):
    acc = []
    while n > 0:
        r = fastq_read_seq(dev)
        if r is None:
            return "eof" if not acc else ("eof", acc)
        acc.insert(0, r)
        n -= 1
    return "ok", acc


def master_produce_workload(n, dev, state, workload_size=200):
    """master.erl:162-178; state = dict(fastq_eof=bool, stopping=bool)."""
    acc = []
    while n > 0:
        if state["fastq_eof"]:
            if not acc:
                state["stopping"] = True
                return "stop"
            raise ErlCrash("function_clause produce_workload")
        r = fastq_read_seqs(dev, workload_size)
        if r == "eof":
            state["fastq_eof"] = True
            return acc
        acc.insert(0, r[1])
        n -= 1
    return acc


def master_cigar_line(name, chromosome, pos, cigar_value, cigar_rate, refseq):
    """master.erl:149-153: dropped under 265, else the io:format line."""
    if cigar_rate < 265:
        return None
    return "\n%s      %s      %d      %s      %d      %s\n" % (name, chromosome, pos, cigar_value, cigar_rate, refseq)


# ------------------------------------------------------------------------------------------------
# seeds:generate_fs/3 (src/seeds.erl:11-85) and the module `fs` it writes, list for list.

def seeds_patterns(qsec, length, defects):
    """The argument patterns of the generated matches/Len clauses in generation order: lists of `length` items,
    a base (int) or None for `_`.  Raises ErlCrash where a clause of another arity would be printed (the generated
    file then has a head mismatch and does not compile)."""
    sub = list(qsec[:length + defects])                      # seeds.erl:43
    out = []

    def gen(acc, rest, d):                                   # acc: Erlang list, head first
        if d == 0:                                           # seeds.erl:51-55
            n = length - len(acc)
            tail = rest[:n]
            seq = list(reversed(tail)) + acc
            out.append(list(reversed(seq)))                  # print_pattern/2 folds it back into reading order (:65-74)
            return
        if rest and len(acc) < length:                       # seeds.erl:57
            s, more = rest[0], rest[1:]
            gen([None] + acc, more, d - 1)                   # mutation  :58
            gen([None] + acc, rest, d - 1)                   # insertion :59
            gen(acc, more, d - 1)                            # deletion  :60
            gen([s] + acc, more, d)                          # match     :61
        # seeds.erl:62: ok

    gen([], sub, defects)
    if any(len(p) != length for p in out):
        raise ErlCrash("fs.erl: head mismatch (a clause of arity /= Len)")
    return out


def fs_find_seeds(pats, ref_seq):
    """fs:find_seeds/1 as generated (seeds.erl:24-41): the clause for exactly Len remaining items tests once;
    otherwise a hit records N and continues Len further on, a miss one further on; Acc is built by prepending."""
    length = len(pats[0]) if pats else 0
    ref = list(ref_seq)

    def matches(win):
        for p in pats:                                       # clause order
            if all(c is None or c == w for c, w in zip(p, win)):
                return True
        return False                                         # matches(_,...,_) -> false

    acc, n, i = [], 1, 0
    while length:
        left = len(ref) - i
        if left == length:                                   # find_seeds(Acc,N,[X1..XLen])
            if matches(ref[i:]):
                acc.insert(0, n)
            break
        if left > length:                                    # find_seeds(Acc,N,[X1..XLen|Ref_Seq])
            if matches(ref[i:i + length]):
                acc.insert(0, n); n += length; i += length
            else:
                n += 1; i += 1
            continue
        break                                                # find_seeds(Acc,_,_) -> Acc
    return acc


def fastq_reverse_complement(ls):
    """fastq:reverse_complement/1 (src/fastq.erl:43-50): rc/2 walks the list prepending the complement; a base outside
    ACGTN has no clause (function_clause -> ErlCrash)."""
    acc = []
    comp = {ord("A"): ord("T"), ord("T"): ord("A"), ord("C"): ord("G"), ord("G"): ord("C"), ord("N"): ord("N")}
    for c in ls:
        if c not in comp:
            raise ErlCrash("fastq:rc/2: no function clause matching %r" % (c,))
        acc.insert(0, comp[c])
    return bytes(acc)
