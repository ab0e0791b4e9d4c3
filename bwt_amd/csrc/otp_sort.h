// lists:sort/2 of Erlang/OTP stdlib (lists.erl: fsplit_1/fsplit_2 run detection, fmergel/rfmergel merging), restated clause by
// clause on a list of integer handles with a comparator -- the C++ twin of bwt_amd/otp_lists.py, needed natively for ONE call
// site: worker_bwt.erl:104 sorts a read's candidate alignments with `fun({{R1,_},_,_},{{R2,_},_,_}) -> R1 > R2 end`, which is
// not a total order, so which of several equal-score alignments ends up first is decided by the merge sort's internals.
// OTP is a dependency the reference does not vendor (OTP 17, README.md:19); restated from the published algorithm, property- and
// cross-tested against the Python twin, never run against an Erlang VM.
#pragma once
#include <algorithm>
#include <vector>

namespace otp {
using L = std::vector<int>;                     // an Erlang list, head first

template <class F> L fmergel(std::vector<L> ls, F &fun, bool asc);
template <class F> L rfmergel(std::vector<L> ls, F &fun, bool asc);

inline L cons(int h, const L &t) { L r; r.reserve(t.size() + 1); r.push_back(h); r.insert(r.end(), t.begin(), t.end()); return r; }
inline L rev_onto(const L &a, const L &b) { L r(a.rbegin(), a.rend()); r.insert(r.end(), b.begin(), b.end()); return r; }   // lists:reverse(A, B)

// fmerge2_1 / fmerge2_2 and rfmerge2_1 / rfmerge2_2: merge T1 with [H2|T2], accumulating in M (newest first)
template <class F> L fmerge2(L t1, int h2, L t2, F &fun) {
    L m;
    size_t i1 = 0, i2 = 0;
    int state = 1, h1 = 0;
    for (;;) {
        if (state == 1) {                                                   // fmerge2_1(T1, H2, Fun, T2, M)
            if (i1 == t1.size()) { L rest(t2.begin() + (long)i2, t2.end()); L r = rev_onto(rest, cons(h2, L(m.rbegin(), m.rend()))); return r; }
            h1 = t1[i1++];
            if (fun(h1, h2)) m.push_back(h1);                               // M is kept oldest first here and reversed on exit
            else { m.push_back(h2); state = 2; }
        } else {                                                            // fmerge2_2(H1, T1, Fun, T2, M)
            if (i2 == t2.size()) { L rest(t1.begin() + (long)i1, t1.end()); L r = rev_onto(rest, cons(h1, L(m.rbegin(), m.rend()))); return r; }
            h2 = t2[i2++];
            if (fun(h1, h2)) { m.push_back(h1); state = 1; }
            else m.push_back(h2);
        }
    }
}
template <class F> L rfmerge2(L t1, int h2, L t2, F &fun) {
    L m;
    size_t i1 = 0, i2 = 0;
    int state = 1, h1 = 0;
    for (;;) {
        if (state == 1) {                                                   // rfmerge2_1(T1, H2, Fun, T2, M)
            if (i1 == t1.size()) { L rest(t2.begin() + (long)i2, t2.end()); return rev_onto(rest, cons(h2, L(m.rbegin(), m.rend()))); }
            h1 = t1[i1++];
            if (fun(h1, h2)) { m.push_back(h2); state = 2; }
            else m.push_back(h1);
        } else {                                                            // rfmerge2_2(H1, T1, Fun, T2, M)
            if (i2 == t2.size()) { L rest(t1.begin() + (long)i1, t1.end()); return rev_onto(rest, cons(h1, L(m.rbegin(), m.rend()))); }
            h2 = t2[i2++];
            if (fun(h1, h2)) m.push_back(h2);
            else { m.push_back(h1); state = 1; }
        }
    }
}

template <class F> L fmergel(std::vector<L> ls, F &fun, bool asc) {
    std::vector<L> acc;                                                     // Acc, newest first
    size_t i = 0;
    while (ls.size() - i >= 2) {
        const L &a = ls[i], &b = ls[i + 1];
        const L &t1 = asc ? a : b, &h2t2 = asc ? b : a;
        L t2(h2t2.begin() + 1, h2t2.end());
        acc.insert(acc.begin(), fmerge2(t1, h2t2[0], t2, fun));
        i += 2;
    }
    if (ls.size() - i == 1) {
        if (acc.empty()) return ls[i];
        L last(ls[i].rbegin(), ls[i].rend());
        acc.insert(acc.begin(), last);
        return rfmergel(acc, fun, asc);
    }
    return rfmergel(acc, fun, asc);
}
template <class F> L rfmergel(std::vector<L> ls, F &fun, bool asc) {
    std::vector<L> acc;
    size_t i = 0;
    while (ls.size() - i >= 2) {
        const L &a = ls[i], &b = ls[i + 1];
        const L &h2t2 = asc ? a : b, &t1 = asc ? b : a;
        L t2(h2t2.begin() + 1, h2t2.end());
        acc.insert(acc.begin(), rfmerge2(t1, h2t2[0], t2, fun));
        i += 2;
    }
    if (ls.size() - i == 1) {
        L last(ls[i].rbegin(), ls[i].rend());
        acc.insert(acc.begin(), last);
        return fmergel(acc, fun, asc);
    }
    return fmergel(acc, fun, asc);
}

// lists:sort(Fun, L)
template <class F> L sort(F fun, const L &l) {
    if (l.size() < 2) return l;
    int x = l[0], y = l[1];
    size_t i = 2;
    const bool up = fun(x, y);                                              // fsplit_1 (ascending runs) or fsplit_2 (descending)
    L r;                                                                    // R, newest first
    std::vector<L> rs;                                                      // Rs, newest first
    auto run = [&]() { L q; q.push_back(y); q.push_back(x); q.insert(q.end(), r.begin(), r.end()); return q; };   // [Y, X | R]
    // le(a, b): the relation the split phase tests -- fun(a, b) for fsplit_1, not fun(a, b) for fsplit_2
    auto le = [&](int a, int b) { return up ? fun(a, b) : !fun(a, b); };
    for (;;) {
        bool restarted = false;
        while (i < l.size()) {
            const int z = l[i++];
            if (le(y, z)) { r.insert(r.begin(), x); x = y; y = z; }
            else if (le(x, z)) { r.insert(r.begin(), x); x = z; }
            else if (r.empty()) { r.push_back(z); }
            else {
                // fsplit_1_1 / fsplit_2_1 with S = z
                const int s = z;
                bool done_inner = false;
                while (i < l.size()) {
                    const int z2 = l[i++];
                    if (le(y, z2)) { r.insert(r.begin(), x); x = y; y = z2; }
                    else if (le(x, z2)) { r.insert(r.begin(), x); x = z2; }
                    else {
                        rs.insert(rs.begin(), run());
                        r.clear();
                        if (le(s, z2)) { y = z2; x = s; } else { y = s; x = z2; }
                        done_inner = true; restarted = true;
                        break;
                    }
                }
                if (!done_inner) {                                          // list exhausted inside the _1 clause
                    std::vector<L> all;
                    all.push_back(L{s}); all.push_back(run());
                    all.insert(all.end(), rs.begin(), rs.end());
                    return up ? rfmergel(all, fun, true) : fmergel(all, fun, false);
                }
                break;
            }
        }
        if (restarted) continue;
        std::vector<L> all;
        all.push_back(run());
        all.insert(all.end(), rs.begin(), rs.end());
        return up ? rfmergel(all, fun, true) : fmergel(all, fun, false);
    }
}
}  // namespace otp
