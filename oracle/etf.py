"""TEST INFRASTRUCTURE ONLY (oracle/): Erlang external-term-format reader/writer for
`fm_indices/<Chrom>_p<J>.fm` files.

Follows the on-disk contract of the reference:
  * writer side  /root/reference/src/fm_index.erl:139-141
        Meta = [{pointers,{Pc,Pg,Pt,Last}},{shift,Shift}],
        Bin  = term_to_binary({Meta,FM})            (FM = tuple of {F,L,I,SA} rows)
  * reader side  /root/reference/src/fm_index.erl:154-156  (binary_to_term)

Only the tags that occur in such a file are handled (131 version, 104 small tuple,
105 large tuple, 108 list, 106 nil, 100 atom / 119 small-utf8 atom / 118 utf8 atom,
97 uint8, 98 int32, 110 small bignum).  Nothing here is imported by the product path.
"""
import struct
import numpy as np


class EtfError(ValueError):
    pass


def _atom(buf, p):
    t = buf[p]
    if t == 100:  # ATOM_EXT (latin-1), 2-byte length
        n = struct.unpack_from(">H", buf, p + 1)[0]
        return buf[p + 3:p + 3 + n].decode("latin1"), p + 3 + n
    if t == 118:  # ATOM_UTF8_EXT
        n = struct.unpack_from(">H", buf, p + 1)[0]
        return buf[p + 3:p + 3 + n].decode("utf8"), p + 3 + n
    if t == 119:  # SMALL_ATOM_UTF8_EXT
        n = buf[p + 1]
        return buf[p + 2:p + 2 + n].decode("utf8"), p + 2 + n
    if t == 115:  # SMALL_ATOM_EXT
        n = buf[p + 1]
        return buf[p + 2:p + 2 + n].decode("latin1"), p + 2 + n
    raise EtfError("atom expected at %d (tag %d)" % (p, t))


def _int(buf, p):
    t = buf[p]
    if t == 97:
        return buf[p + 1], p + 2
    if t == 98:
        return struct.unpack_from(">i", buf, p + 1)[0], p + 5
    if t == 110:  # SMALL_BIG_EXT: n, sign, little-endian digits
        n = buf[p + 1]
        sign = buf[p + 2]
        v = int.from_bytes(buf[p + 3:p + 3 + n], "little")
        return (-v if sign else v), p + 3 + n
    raise EtfError("integer expected at %d (tag %d)" % (p, t))


def _tuple_hdr(buf, p):
    t = buf[p]
    if t == 104:
        return buf[p + 1], p + 2
    if t == 105:
        return struct.unpack_from(">I", buf, p + 1)[0], p + 5
    raise EtfError("tuple expected at %d (tag %d)" % (p, t))


def read_fm(path):
    """Decode a .fm file.

    Returns dict(pointers=(Pc,Pg,Pt,Last), shift=Shift, F=u8[n], L=u8[n], I=i64[n], SA=i64[n])
    with rows in file order (row r of the Erlang tuple, 1-based, is index r-1 here).
    """
    with open(path, "rb") as f:
        buf = f.read()
    if buf[0] != 131:
        raise EtfError("not an external term (version byte %d)" % buf[0])
    p = 1
    n, p = _tuple_hdr(buf, p)
    if n != 2:
        raise EtfError("top-level tuple arity %d" % n)
    # Meta: proplist
    if buf[p] != 108:
        raise EtfError("meta list expected")
    nmeta = struct.unpack_from(">I", buf, p + 1)[0]
    p += 5
    meta = {}
    for _ in range(nmeta):
        a, p = _tuple_hdr(buf, p)
        if a != 2:
            raise EtfError("meta entry arity %d" % a)
        key, p = _atom(buf, p)
        if key == "pointers":
            a2, p = _tuple_hdr(buf, p)
            vals = []
            for _ in range(a2):
                v, p = _int(buf, p)
                vals.append(v)
            meta[key] = tuple(vals)
        else:
            v, p = _int(buf, p)
            meta[key] = v
    if buf[p] != 106:
        raise EtfError("meta list tail must be nil")
    p += 1
    rows, p = _tuple_hdr(buf, p)
    F = np.empty(rows, np.uint8)
    L = np.empty(rows, np.uint8)
    I = np.empty(rows, np.int64)
    SA = np.empty(rows, np.int64)
    for r in range(rows):
        # {F,L,I,SA}: 104 4 97 F 97 L <int> <int>
        if buf[p] != 104 or buf[p + 1] != 4 or buf[p + 2] != 97 or buf[p + 4] != 97:
            raise EtfError("row %d malformed at %d" % (r + 1, p))
        F[r] = buf[p + 3]
        L[r] = buf[p + 5]
        p += 6
        t = buf[p]
        if t == 97:
            I[r] = buf[p + 1]; p += 2
        elif t == 98:
            I[r] = struct.unpack_from(">i", buf, p + 1)[0]; p += 5
        else:
            I[r], p = _int(buf, p)
        t = buf[p]
        if t == 97:
            SA[r] = buf[p + 1]; p += 2
        elif t == 98:
            SA[r] = struct.unpack_from(">i", buf, p + 1)[0]; p += 5
        else:
            SA[r], p = _int(buf, p)
    if p != len(buf):
        raise EtfError("trailing bytes: consumed %d of %d" % (p, len(buf)))
    return dict(pointers=meta["pointers"], shift=meta["shift"], F=F, L=L, I=I, SA=SA)


def _enc_int(v):
    # term_to_binary: SMALL_INTEGER_EXT for 0..255, INTEGER_EXT for int32, else SMALL_BIG_EXT
    if 0 <= v <= 255:
        return bytes((97, v))
    if -(1 << 31) <= v < (1 << 31):
        return b"b" + struct.pack(">i", v)
    sign = 1 if v < 0 else 0
    v = abs(v)
    d = v.to_bytes((v.bit_length() + 7) // 8, "little")
    return bytes((110, len(d), sign)) + d


def _enc_tuple_hdr(n):
    return bytes((104, n)) if n <= 255 else b"i" + struct.pack(">I", n)


def write_fm(path, pointers, shift, F, L, I, SA):
    """Re-serialise exactly as term_to_binary/1 (OTP 17, no options) lays the term out."""
    out = bytearray()
    out += bytes((131,)) + _enc_tuple_hdr(2)
    out += b"l" + struct.pack(">I", 2)
    out += _enc_tuple_hdr(2) + b"d" + struct.pack(">H", 8) + b"pointers"
    out += _enc_tuple_hdr(4)
    for v in pointers:
        out += _enc_int(int(v))
    out += _enc_tuple_hdr(2) + b"d" + struct.pack(">H", 5) + b"shift" + _enc_int(int(shift))
    out += b"j"
    n = len(F)
    out += _enc_tuple_hdr(n)
    for r in range(n):
        out += bytes((104, 4, 97, int(F[r]), 97, int(L[r])))
        out += _enc_int(int(I[r]))
        out += _enc_int(int(SA[r]))
    with open(path, "wb") as f:
        f.write(out)
