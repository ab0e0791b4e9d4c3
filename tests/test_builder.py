"""Index construction: the oracle's numpy suffix sort reproduces the bundled fixture's
rows (CPU), and the GPU builder + ETF writer reproduce the bundled .fm byte for byte (GPU)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import FM_PATH, GOLDEN, REF_PATH

FASTA = os.path.join(GOLDEN, "bwt_files", "test_genome.fasta")


def test_oracle_builder_reproduces_fixture(fm_dict, ref_bytes):
    from oracle import erl_literal as E
    rows = E.fm_rows(ref_bytes)                       # GL000193.1 holds no N: no random stream involved
    assert rows["pointers"] == fm_dict["pointers"]
    for k in ("F", "L", "I", "SA"):
        assert (rows[k] == fm_dict[k]).all(), k


def test_as183_stream():
    from oracle import erl_literal as E
    r = E.AS183()
    # first values of random:uniform() from the default seed {3172,9814,20125} (OTP `random` docs' algorithm)
    a1 = (3172 * 171) % 30269; a2 = (9814 * 172) % 30307; a3 = (20125 * 170) % 30323
    v = a1 / 30269 + a2 / 30307 + a3 / 30323
    assert r.uniform() == v - int(v)
    assert all(1 <= r.uniform(4) <= 4 for _ in range(100))


def test_chunking_rules():
    from oracle import erl_literal as E
    # contig "3" of the toy genome: 13 leading N, 15 trailing N
    raw = b"NNNNNNNNNNNNNAAATACCTACACCACACACCACAGGGCACACACACACACACNNNNNNNNNNNNNNN\n"
    ch = E.make_index_chunks(raw)
    assert ch == [(1, 14, b"AAATACCTACACCACACACCACAGGGCACACACACACACAC")]
    body = b"ACGT" * ((1 << 22) + 10)                 # 2^24 + 40 bases -> 2 chunks, 2-bp overlap
    ch = E.make_index_chunks(body + b"\n")
    assert len(ch) == 2
    (j1, s1, c1), (j2, s2, c2) = ch
    size = (len(body) + 4) // 2
    assert (j1, s1, len(c1)) == (1, 1, size) and (j2, s2) == (2, 1 + size - 2)
    assert c1[-2:] == c2[:2] and len(c1) - 2 + len(c2) == len(body)


@pytest.mark.gpu
def test_gpu_builder_reproduces_bundled_fm(tmp_path, ref_bytes):
    from bwt_amd import lib
    ix = lib.Index.build(ref_bytes, shift=1)
    assert ix.meta() == ((53511, 94127, 134735, 189790), 1)
    ix.save(str(tmp_path / "x.fm"), str(tmp_path / "x.ref"))
    ix.close()
    assert hashlib.sha256(open(tmp_path / "x.fm", "rb").read()).hexdigest() == \
        "9a403b6340c1229ad33a2b632c541958566c8849a997210c9fc89268207425b9"
    assert open(tmp_path / "x.ref", "rb").read() == ref_bytes


@pytest.mark.gpu
def test_gpu_make_indices_toy_genome(tmp_path):
    """fm_index:t/0 builds contigs "2" and "4" of bwt_files/test_genome.fasta (src/fm_index.erl:15-17);
    here every contig that holds all four bases, against the oracle's restatement (N -> AS183 stream)."""
    from bwt_amd import fm_index, lib
    from oracle import erl_literal as E, etf
    names = ["2", "3", "4"]
    try:
        written = fm_index.make_indices(FASTA, names, out_dir=str(tmp_path))
    finally:
        fm_index._env.clear()
    assert [os.path.basename(w) for w in written] == ["2_p1", "3_p1", "4_p1"]
    rng = E.AS183()
    raw = {n: r for n, r in fm_index._contigs(FASTA)}
    for name in names:
        (j, shift, chunk), = E.make_index_chunks(raw[name])
        rows = E.fm_rows(E.ref_seq_of_chunk(chunk, rng))
        d = etf.read_fm(str(tmp_path / (name + "_p1.fm")))
        assert d["shift"] == shift and d["pointers"] == rows["pointers"]
        for k in ("F", "L", "I", "SA"):
            assert (d[k] == rows[k]).all(), (name, k)
        assert open(tmp_path / (name + "_p1.ref"), "rb").read() == chunk


@pytest.mark.gpu
def test_gpu_built_index_searches_like_loaded_index(ref_bytes, orc_fm):
    """An index built on the GPU answers exactly like the one parsed from the .fm."""
    from bwt_amd import lib
    from conftest import synth_reads
    from oracle import orc
    ix = lib.Index.build(ref_bytes, shift=1)
    reads = synth_reads(ref_bytes, 200, 100, 99)
    out = ix.sga_batch(*lib.pack_reads(reads))
    for i, q in enumerate(reads):
        a, b = int(out["clus_off"][i]), int(out["clus_off"][i + 1])
        assert list(zip(out["S"][a:b].tolist(), out["D"][a:b].tolist())) == orc.sga(orc_fm, q)
    ix.close()


@pytest.mark.gpu
def test_native_index_round_trip(tmp_path, ref_bytes, orc_fm):
    """The compact native format: an index saved from HBM and reopened is the same index -- columns, pointers,
    Ref1, and what it answers; the host-only checker agrees with the header, and a damaged file is refused."""
    import numpy as np
    from bwt_amd import lib
    from conftest import FM_PATH, REF_PATH, synth_reads
    ix = lib.Index.open(FM_PATH, REF_PATH, 0)
    path = str(tmp_path / "GL000193.1_p1.bwtgpu")
    ix.save_native(path)
    assert os.path.getsize(path) < 0.4 * os.path.getsize(FM_PATH)            # 5.4 bytes per row against the term format's 16
    assert lib.native_check(path) == ix.meta()
    ix2 = lib.Index.open_native(path, 0)
    try:
        assert ix2.meta() == ix.meta()
        L1, SA1 = ix.export(); L2, SA2 = ix2.export()
        assert np.array_equal(L1, L2) and np.array_equal(SA1, SA2)
        assert ix2.ref1(0, 600) == ix.ref1(0, 600)
        reads = synth_reads(ref_bytes, 400, 100, 9, indel_frac=0.2, n_frac=0.1)
        a = ix.align_batch(*lib.pack_reads(reads)); b = ix2.align_batch(*lib.pack_reads(reads))
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(a[k], b[k]), k
        assert a["cigar"] == b["cigar"] and len(a["score"]) > 100
        # rebuilt by the GPU builder and saved natively: the same bytes (the format is a function of the index)
        ix3 = lib.Index.build(ref_bytes, shift=1)
        p3 = str(tmp_path / "rebuilt.bwtgpu")
        ix3.save_native(p3)
        ix3.close()
        assert open(p3, "rb").read() == open(path, "rb").read()
    finally:
        ix2.close(); ix.close()
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 1
    bad = str(tmp_path / "bad.bwtgpu")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(lib.BwtError):
        lib.native_check(bad)
    with pytest.raises(lib.BwtError):
        lib.Index.open_native(bad, 0)
    open(bad, "wb").write(bytes(raw[:-5]))
    with pytest.raises(lib.BwtError):
        lib.native_check(bad)


def _fnv1a_words(data):
    """the section checksum of the native format (csrc/native_fmt.cpp): FNV-1a over 8-byte words, the tail byte-wise"""
    h, mask = 0xcbf29ce484222325, (1 << 64) - 1
    n8 = len(data) // 8 * 8
    for (w,) in __import__("struct").iter_unpack("<Q", data[:n8]):
        h = ((h ^ w) * 0x100000001b3) & mask
    for c in data[n8:]:
        h = ((h ^ c) * 0x100000001b3) & mask
    return h


@pytest.mark.gpu
def test_native_index_contents_are_verified(tmp_path):
    """A file whose checksums are CONSISTENT but whose contents break the index invariants (the checksum is not keyed: ADVICE r2)
    passes the host-only check and is refused when opened: an SA value >= rows, a rank block whose counts do not continue the
    previous block's.  Nothing is left behind under the final name by a save that fails half-way (tmp + rename)."""
    import struct
    from bwt_amd import lib
    from conftest import FM_PATH, REF_PATH
    ix = lib.Index.open(FM_PATH, REF_PATH, 0)
    path = str(tmp_path / "ok.bwtgpu")
    ix.save_native(path)
    ix.close()
    assert not os.path.exists(path + ".tmp")
    raw = bytearray(open(path, "rb").read())
    rows, = struct.unpack_from("<I", raw, 16)
    nblk, = struct.unpack_from("<I", raw, 44)
    o_blocks, o_sa = 128, 128 + nblk * 64
    o_ref = o_sa + rows * 4
    assert _fnv1a_words(bytes(raw[o_sa:o_ref])) == struct.unpack_from("<Q", raw, 64)[0]     # the restated checksum is the file's
    # 1. one SA value far outside the rows, checksum brought in line
    bad = bytearray(raw)
    struct.pack_into("<I", bad, o_sa + 4 * 1234, 0xFFFFFF00)
    struct.pack_into("<Q", bad, 64, _fnv1a_words(bytes(bad[o_sa:o_ref])))
    p1 = str(tmp_path / "bad_sa.bwtgpu")
    open(p1, "wb").write(bytes(bad))
    lib.native_check(p1)                                                       # consistent as a file
    with pytest.raises(lib.BwtError) as e:
        lib.Index.open_native(p1, 0)
    assert e.value.code == -74 and "SA value" in str(e.value)
    # 2. a rank block's count word off by one
    bad = bytearray(raw)
    w, = struct.unpack_from("<I", bad, o_blocks + 64 * 7)
    struct.pack_into("<I", bad, o_blocks + 64 * 7, w + 1)
    struct.pack_into("<Q", bad, 56, _fnv1a_words(bytes(bad[o_blocks:o_sa])))
    p2 = str(tmp_path / "bad_blocks.bwtgpu")
    open(p2, "wb").write(bytes(bad))
    lib.native_check(p2)
    with pytest.raises(lib.BwtError) as e:
        lib.Index.open_native(p2, 0)
    assert e.value.code == -74 and "counts" in str(e.value)
    # the untouched file still opens
    lib.Index.open_native(path, 0).close()
