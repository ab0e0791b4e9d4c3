"""CPU tests of the drop-in boundary: libbwtgpu.so loads, exports every symbol include/bwt_gpu.h
declares, its compiled constants equal include/bwt.hrl's, the host-only format helpers work, and the
product never routes through the oracle.  No compute calls (no GPU here)."""
import hashlib
import os
import re

import pytest

from conftest import FM_PATH, REF_PATH, ROOT


def test_header_symbols_exported():
    from bwt_amd import lib
    l = lib.load()
    hdr = open(os.path.join(ROOT, "include", "bwt_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(bwt_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), n
    assert names == set(lib.EXPORTS)


def test_constants_equal_bwt_hrl():
    from bwt_amd import lib
    c = lib.constants()
    # /root/reference/include/bwt.hrl:4-7,14-18 ; src/worker_bwt.erl:17
    assert (c.tolerance, c.min_len, c.max_range, c.max_len) == (20, 8, 45, 15)
    assert (c.match, c.mismatch, c.gap_penalty, c.gap_ext_penalty, c.unknown) == (2, -1, -2, -1, 1.6)
    assert c.ref_extension_len == 200


def test_fm_check_and_rewrite(tmp_path):
    from bwt_amd import lib
    assert lib.fm_check(FM_PATH) == ((53511, 94127, 134735, 189790), 1)
    out = tmp_path / "copy.fm"
    lib.fm_rewrite(FM_PATH, str(out))
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == \
        "9a403b6340c1229ad33a2b632c541958566c8849a997210c9fc89268207425b9"


def test_format_errors(tmp_path):
    from bwt_amd import lib
    with pytest.raises(lib.BwtError) as e:
        lib.fm_check(str(tmp_path / "missing.fm"))
    assert e.value.code == -5
    bad = tmp_path / "bad.fm"
    data = bytearray(open(FM_PATH, "rb").read())
    bad.write_bytes(bytes(data[:1000]))                       # truncated
    with pytest.raises(lib.BwtError) as e:
        lib.fm_check(str(bad))
    assert e.value.code == -74
    data[70] ^= 0x01                                          # flips a bit inside row 1
    bad.write_bytes(bytes(data))
    with pytest.raises(lib.BwtError):
        lib.fm_check(str(bad))
    with pytest.raises(lib.BwtError) as e:                    # file errors come before any HIP call
        lib.Index.open(str(tmp_path / "missing.fm"), REF_PATH)
    assert e.value.code == -5


def test_fm_name_mirror():
    from bwt_amd import fm_index
    assert fm_index.fm_name("GL000193.1", 1) == "fm_indices/GL000193.1_p1"      # src/fm_index.erl:145-151
    fm_index.set_env("fm_indices", "/data/idx")
    try:
        assert fm_index.fm_name("21", 3) == "/data/idx/21_p3"
    finally:
        fm_index._env.clear()


def test_worker_pick_uses_otp_sort_ties():
    from bwt_amd import worker_bwt
    a = ((200, "100M"), 10, b"x"); b = ((200, "100M"), 20, b"y"); c = ((150, "100M"), 30, b"z")
    assert worker_bwt.pick([]) == "no_cigar"
    assert worker_bwt.pick([a]) == a
    assert worker_bwt.pick([a, b]) == b            # lists:sort/2 with `>`: [X,Y] equal -> [Y,X]
    assert worker_bwt.pick([c, a]) == a
    assert worker_bwt.pick([a, c, b])[0][0] == 200


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under bwt_amd/ (or the C ABI sources) may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bwt_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), fn
                assert "liborc" not in text and "bwt_oracle" not in text, fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bwt_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        lib.load()


def test_native_check_refuses_what_is_not_an_index(tmp_path):
    """bwt_native_check is host-only: junk, a short file and a wrong magic are BWT_EFORMAT, a missing file BWT_EIO."""
    import pytest
    from bwt_amd import lib
    p = tmp_path / "x.bwtgpu"
    for blob in (b"", b"BWTGPUIX" + b"\0" * 50, b"NOTANIDX" + b"\0" * 200, b"BWTGPUIX" + b"\x01\0\0\0" + b"\0" * 116 + b"\0" * 64):
        p.write_bytes(blob)
        with pytest.raises(lib.BwtError) as e:
            lib.native_check(str(p))
        assert e.value.code == -74
    with pytest.raises(lib.BwtError) as e:
        lib.native_check(str(tmp_path / "missing.bwtgpu"))
    assert e.value.code == -5


def test_library_never_reads_the_environment():
    """Which kernels run is decided by bwt_options (include/bwt_gpu.h), not by ambient variables: the only getenv calls left in
    the product's sources are debug prints (BWT_*_DEBUG), none of them in the translation unit that launches kernels."""
    src_dir = os.path.join(ROOT, "bwt_amd", "csrc")
    for fn in os.listdir(src_dir):
        if not fn.endswith((".hip", ".cpp", ".h")):
            continue
        text = open(os.path.join(src_dir, fn), errors="replace").read()
        for m in re.finditer(r'getenv\("([A-Z_0-9]+)"\)', text):
            assert m.group(1).endswith("_DEBUG"), (fn, m.group(1))
        if fn in ("bwt_gpu.hip", "kernels.hip.h", "pack.cpp", "index_build.hip", "fs_seeds.hip"):
            assert "getenv" not in text, fn


def test_options_struct_round_trip():
    from bwt_amd import lib
    o = lib.make_options()
    assert o.struct_size == 4 * (1 + len(lib.OPTION_FIELDS)) and all(getattr(o, k) == -1 for k in lib.OPTION_FIELDS)
    o = lib.make_options(sw_band=0, sw_band_min=7)
    assert (o.sw_band, o.sw_band_min, o.seed_fast) == (0, 7, -1)
    assert lib.options_from_env({"BWT_SW_WAVE": "0", "BWT_SEED_BIG": "1", "BWT_KTAB": "9", "HOME": "/"}) == {"sw_wave": 0, "seed_big": 1}
    with pytest.raises(KeyError):
        lib.make_options(no_such_switch=1)


def _pack_reference(bases):
    """numpy restatement of the packed wire format (include/bwt_gpu.h)"""
    import numpy as np
    code = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    c = code[bases]
    bad = np.nonzero(c == 255)[0]
    c = np.where(c == 255, 0, c).astype(np.uint8)
    nw = (len(bases) + 31) // 32
    pad = np.zeros(nw * 32, np.uint8); pad[:len(c)] = c
    lo = np.packbits((pad & 1).reshape(nw, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
    hi = np.packbits((pad >> 1).reshape(nw, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
    codes = np.empty(2 * nw, np.uint32); codes[0::2] = lo; codes[1::2] = hi
    exc = bad.astype(np.uint64) | (bases[bad].astype(np.uint64) << np.uint64(56))
    return codes, exc


@pytest.mark.parametrize("total", [0, 1, 7, 31, 32, 33, 100, 4097, 3_000_001])
def test_pack_reads_matches_the_format(total):
    """bwt_pack_reads (host only): code words and exception list of random streams with N, IUPAC and arbitrary bytes, at lengths
    around the word and group boundaries, single- and multi-threaded."""
    import numpy as np
    from bwt_amd import lib
    rng = np.random.default_rng(total + 5)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total, dtype=np.uint8)].copy()
    if total:
        k = max(1, total // 50)
        bases[rng.integers(0, total, k)] = np.frombuffer(b"NNNNRYKMSWBDVHacgtn\x00\xff@", np.uint8)[rng.integers(0, 22, k)]
    want_c, want_e = _pack_reference(bases)
    for threads in (1, 0, 5):
        codes, exc = lib.pack_2bit(bases, threads)
        assert len(codes) == len(want_c) + 2 and np.array_equal(codes[:len(want_c)], want_c) and not codes[len(want_c):].any()
        assert np.array_equal(exc, want_e)


def test_hits_expand_on_the_host():
    """bwt_hits_expand: plain hits get "<Lq>M" on a window of Lq bases at pos - Shift + 200, extras their own text and window."""
    import numpy as np
    from bwt_amd import lib
    hits = np.array([(0, 200, 1000), (2, 190, 50), (2, 188, 5000), (5, 140, -3)], lib.HIT_REC)
    extra = np.array([(1, 103, 0, 7), (3, 73, 7, 8)], lib.HIT_EXTRA)
    text = b"50M3D50" + b"30M2I41M"
    off = np.array([0, 100, 200, 300, 400, 500, 573], np.uint64)
    out = lib.hits_expand(6, 0, off, 7, hits, extra, text)
    assert out["hit_off"].tolist() == [0, 1, 1, 3, 3, 3, 4]
    assert out["score"].tolist() == [200, 190, 188, 140] and out["pos"].tolist() == [1000, 50, 5000, -3]
    assert out["win_off"].tolist() == [1193, 243, 5193, 190] and out["win_len"].tolist() == [100, 103, 100, 73]
    cig = [out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])] for h in range(4)]
    assert cig == [b"100M", b"50M3D50", b"100M", b"30M2I41M"]
    same = lib.hits_expand(6, 100, None, 7, hits[:3], extra[:1], text)
    assert same["win_len"].tolist() == [100, 103, 100] and same["hit_off"].tolist() == [0, 1, 1, 3, 3, 3, 3]
    with pytest.raises(lib.BwtError):                          # an extra that names no hit
        lib.hits_expand(6, 100, None, 7, hits[:1], extra[:1], text)
    with pytest.raises(lib.BwtError):                          # hits out of order
        lib.hits_expand(6, 100, None, 7, hits[::-1].copy(), extra[:0], text)


def test_narrow_table_expands_on_the_host():
    """bwt_hits_expand_narrow: reads with status 2 take their one plain hit from the narrow arrays (Pos = offset - 200 + Shift), the
    others from the record stream; a table whose parts disagree is refused, never read out of bounds."""
    import numpy as np
    from bwt_amd import lib
    st = np.array([2, 0, 1, 2, 0, -1], np.int8)
    noff = np.array([1193, 777], np.uint32); nsc = np.array([200, 197], np.int16)
    hits = np.array([(1, 190, 50), (1, 188, 5000), (4, 140, -3)], lib.HIT_REC)
    extra = np.array([(0, 103, 0, 7), (2, 73, 7, 8)], lib.HIT_EXTRA)
    text = b"50M3D50" + b"30M2I41M"
    out = lib.hits_expand(6, 100, None, 7, hits, extra, text, st, noff, nsc)
    assert out["hit_off"].tolist() == [0, 1, 3, 3, 4, 5, 5]
    assert out["score"].tolist() == [200, 190, 188, 197, 140] and out["pos"].tolist() == [1000, 50, 5000, 584, -3]
    assert out["win_off"].tolist() == [1193, 243, 5193, 777, 190] and out["win_len"].tolist() == [100, 103, 100, 100, 73]
    cig = [out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])] for h in range(5)]
    assert cig == [b"100M", b"50M3D50", b"100M", b"100M", b"30M2I41M"]
    with pytest.raises(lib.BwtError):                          # a narrow read that also has a record
        lib.hits_expand(6, 100, None, 7, np.array([(0, 1, 1)], lib.HIT_REC), extra[:0], b"", st, noff, nsc)
    with pytest.raises(lib.BwtError):                          # more status-2 reads than narrow records
        lib.hits_expand(6, 100, None, 7, hits, extra, text, st, noff[:1], nsc[:1])
    with pytest.raises(lib.BwtError):                          # fewer
        lib.hits_expand(6, 100, None, 7, hits, extra, text, np.array([2, 0, 1, 0, 0, -1], np.int8), noff, nsc)
    with pytest.raises(lib.BwtError):                          # a record for a read the batch does not have
        lib.hits_expand(4, 100, None, 7, hits, extra, text, st[:4], noff, nsc)


def test_makefile_lists_every_local_header():
    """A header that the library's sources include but the Makefile does not name is a change that does not rebuild the library (round
    4: an A/B of sw_cert.h measured the old build)."""
    import re
    csrc = os.path.join(ROOT, "bwt_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    hdr = set(re.search(r"^HDR = (.*)$", mk, re.M).group(1).split())
    src = set(re.search(r"^SRC = (.*)$", mk, re.M).group(1).split())
    seen = set()
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".cpp", ".h")):
            continue
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(os.path.join(csrc, name)).read(), re.M):
            seen.add(os.path.normpath(os.path.join(csrc, inc)))
    listed = {os.path.normpath(os.path.join(csrc, h)) for h in hdr | src}
    missing = sorted(p for p in seen if p not in listed and os.path.exists(p))
    assert not missing, missing
