"""Host mirror of seeds:generate_fs/3 (/root/reference/src/seeds.erl:11-85): the generator of the
`fs` module -- one `matches/Len` clause per way of spending exactly `Defects` edits (mutation,
insertion, deletion; edited positions become `_` wildcards) on the first Len+Defects bases of the
query, plus a sliding-window `find_seeds/1`.  The reference never wires it into the worker path
(SURVEY 8a-12); this mirror produces the same patterns and offers the generated module's behaviour
(`find_seeds`) directly, so the semantics stay available without an Erlang compiler."""


def patterns(qsec, length, defects):
    """The argument patterns of the generated matches/Len clauses, in generation order: lists of length
    `length` whose items are a base (int) or None for `_`."""
    sub = list(qsec[:length + defects])
    out = []

    def gen(acc, rest, d):
        # acc is built by prepending (newest first), exactly like the Erlang accumulator
        if d == 0:                                           # seeds.erl:51-55
            n = length - len(acc)
            tail = rest[:n]
            out.append(list(reversed(tail)) + acc)
            return
        if rest and len(acc) < length:                       # seeds.erl:57-61
            s, more = rest[0], rest[1:]
            gen([None] + acc, more, d - 1)                   # mutation
            gen([None] + acc, rest, d - 1)                   # insertion
            gen(acc, more, d - 1)                            # deletion
            gen([s] + acc, more, d)                          # match
        # seeds.erl:62: any other case produces nothing

    gen([], sub, defects)
    # print_pattern folds the (reversed) accumulator back into reading order
    return [list(reversed(p)) for p in out]


def clause_text(pattern):
    """One generated clause, as print_pattern/2 writes it."""
    return "matches(%s) -> true;\n" % ",".join("_" if c is None else "$" + chr(c) for c in pattern)


def generate_fs(qsec, length, defects):
    """The text of fs.erl as seeds:generate_fs/3 writes it."""
    args = ",".join("X%d" % j for j in range(1, length + 1))
    args1 = args[3:]
    head = ("%% Seeds finding algo\n%%\n%% Cloudozer(c), 2014\n\n\n-module(fs).\n-export([matches/%d, find_seeds/1]).\n"
            % length)
    fs = ("\n\nfind_seeds(Ref_Seq) ->\n\tfind_seeds([],1, Ref_Seq).\n\nfind_seeds(Acc,N,[%s]) ->\n\tcase matches(%s) of\n"
          "\t\ttrue -> [N|Acc];\n\t\t_ -> Acc\n\tend;\n" % (args, args))
    fs2 = ("\nfind_seeds(Acc,N,[%s|Ref_Seq]) ->\n\tcase matches(%s) of\n\t\ttrue -> find_seeds([N|Acc],N+length([%s]),Ref_Seq);\n"
           "\t\t_ -> find_seeds(Acc,N+1,[%s|Ref_Seq])\n\tend;\n\nfind_seeds(Acc,_,_) -> Acc.\n\n\n" % (args, args, args, args1))
    body = "".join(clause_text(p) for p in patterns(qsec, length, defects))
    false = "matches(%s) -> false.\n" % ",".join("_" for _ in range(length))
    return head + fs + fs2 + body + false


def find_seeds(pats, ref_seq):
    """What the generated fs:find_seeds/1 computes: slide over the reference, record the 1-based position of
    a window that matches any pattern and jump Len, else advance by 1; result newest first."""
    length = len(pats[0]) if pats else 0
    acc, n, i = [], 1, 0
    ref = list(ref_seq)

    def matches(win):
        return any(all(c is None or c == w for c, w in zip(p, win)) for p in pats)
    while length and i + length <= len(ref):
        if matches(ref[i:i + length]):
            acc.insert(0, n)
            if i + length == len(ref):
                break
            n += length; i += length
        else:
            n += 1; i += 1
    return acc
