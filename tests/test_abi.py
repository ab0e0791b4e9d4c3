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
