"""c_src/bwt_nif.c EXECUTED: the shim is built against a stand-in NIF runtime (tests/nif_check/fake_enif.c: terms
as tagged C structs, reference-counted resources, a mutex) and driven by tests/nif_check/nif_driver.c through load/2, constants/0,
index_open/3, index_meta/1, align_batch/2 (Seq as charlist, binary and nested iolist; [], improper lists, a read the reference would
crash on), align_workload/2 and two threads on one Index term.
  * without a GPU: against a stub of libbwtgpu, under AddressSanitizer + UBSan + LeakSanitizer -- the shim's own memory handling;
  * on the GPU box (-m gpu): against the real library, its output compared with bwt_align_batch called directly.
This pins NOTHING about parity with Erlang and nothing about OTP's own behaviour: it is the first execution of the 300 lines that sit
on the boundary (NIF row; src/master.erl:153 prints Cigar / RefSeq with ~s: they must be latin-1 lists)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import FM_PATH, REF_PATH, ROOT, synth_reads

NC = os.path.join(ROOT, "tests", "nif_check")
SRC = [os.path.join(NC, "fake_enif.c"), os.path.join(NC, "nif_driver.c"), os.path.join(ROOT, "c_src", "bwt_nif.c")]
INC = ["-I", NC, "-I", os.path.join(ROOT, "include")]


def write_reads(path, reads):
    with open(path, "w") as f:
        for i, q in enumerate(reads):
            f.write("read.%d\t%s\n" % (i, q.decode("latin1")))


def parse(out):
    """-> {tag: [(name, value)]} for the R / T / Wk lines, and the single-value lines"""
    lists, single = {}, {}
    for ln in out.splitlines():
        tag, _, rest = ln.partition(" ")
        if tag in ("R", "T") or (tag.startswith("W") and tag[1:].isdigit()):
            name, _, val = rest.partition(" ")
            lists.setdefault(tag, []).append((name.strip('"'), val))
        else:
            single.setdefault(tag, []).append(rest)
    return lists, single


def test_shim_runs_clean_under_sanitizers_against_a_stub_library(tmp_path):
    exe = str(tmp_path / "nif_asan")
    objs = []
    for src, cc, std in ((os.path.join(ROOT, "bwt_amd", "csrc", "pack.cpp"), "g++", "-std=c++17"), (os.path.join(NC, "stub_bwtgpu.cpp"), "g++", "-std=c++17")):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call([cc, std, "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-c", src, "-o", o])
        objs.append(o)
    for src in SRC:
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=gnu11", "-g", "-O1", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] + INC + ["-c", src, "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", "-fsanitize=address,undefined"] + objs + ["-lpthread", "-o", exe])
    rng = np.random.default_rng(3)
    reads = []
    for i in range(1300):
        L = int(rng.integers(0, 160))
        q = bytes(rng.choice(list(b"ACGTZ" if i % 50 == 7 else b"ACGT"), L).tolist())
        if i % 41 == 3 and L > 10:
            q = q[:5] + b"X" + q[6:]
        reads.append(q)
    reads_path = str(tmp_path / "reads.tsv")
    write_reads(reads_path, reads)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, "stub.fm", "stub.ref", reads_path], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    lists, single = parse(r.stdout)
    assert single["DONE"] == [""] and single["OPEN"] == ["ok"] and single["OPEN_MISSING"] == ["error_tuple"] and single["OPEN_BADARG"] == ["badarg"]
    assert single["META"] == ["{10,20,30,40,1}"] and single["CONSTANTS"] == ["{20,8,45,15,2,-1,-2,-1,1.6,200}"]
    for k in ("IMPROPER", "NOT_A_PAIR", "NO_SEQ", "NO_INDEX", "META_BADARG", "WORKLOAD_BADARG"):
        assert single[k] == ["badarg"], k
    assert single["EMPTY"] == ["[]"]
    # the resource: alive until the last environment that holds the term goes, destroyed exactly once, nothing left behind
    assert single["BEFORE_FREE"] == ["resources 1 dtors 0"] and single["AFTER_FREE"] == ["resources 0 dtors 1 terms 0"]
    # what the stub "aligned" came back through the shim unchanged, for all three input forms of Seq
    assert len(lists["R"]) == len(reads)
    for i, ((name, val), q) in enumerate(zip(lists["R"], reads)):
        assert name == "read.%d" % i
        if len(q) < 8:
            assert val == "nothing"
        elif b"X" in q:
            assert val == "{error,bad_base}"
        elif q[:1] == b"Z":
            assert val == "{error,window_out_of_range}"
        else:
            k = q[0] % 4
            hits = []
            for h in range(k):
                cig = "%dM3D%dM" % (len(q) // 2, len(q) - len(q) // 2) if h & 1 else "%dM" % len(q)
                wl = len(q) + 3 if h & 1 else len(q)
                pos = 7 * sum(q) + h
                hits.append('{{%d,"%s"},%d,"%s"}' % (2 * len(q) - h, cig, pos, "ACGT"[(pos - 1 + 200) & 3] * wl))
            assert val == ("[" + ",".join(hits) + "]" if hits else "[]"), (i, val)
    # the whole-workload call = the single calls, list by list; the two threads together = the single call
    flat = [x for k in range(len(reads) // 200 + 1) for x in lists.get("W%d" % k, [])]
    assert flat == lists["R"] and single["WLEN"][-1].split()[1] == "0"
    assert lists["T"] == lists["R"]


@pytest.mark.gpu
def test_shim_against_the_real_library_on_the_gpu(tmp_path, gpu_index, ref_bytes):
    from bwt_amd import lib
    exe = str(tmp_path / "nif_gpu")
    subprocess.check_call(["gcc", "-std=gnu11", "-g", "-O1", "-Wall", "-Wextra", "-Werror"] + INC + SRC +
                          ["-L", os.path.join(ROOT, "bwt_amd"), "-lbwtgpu", "-Wl,-rpath," + os.path.join(ROOT, "bwt_amd"), "-lpthread", "-o", exe])
    reads = (synth_reads(ref_bytes, 700, 100, 11, max_subs=2) + synth_reads(ref_bytes, 500, 100, 12, max_subs=2, n_frac=0.1, iupac_frac=0.05, indel_frac=0.3) +
             [b"", b"ACGT", ref_bytes[1000:1100] + b"X", b"N" * 40, ref_bytes[7000:7150]])
    reads_path = str(tmp_path / "reads.tsv")
    write_reads(reads_path, reads)
    r = subprocess.run([exe, FM_PATH, REF_PATH, reads_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lists, single = parse(r.stdout)
    (pc, pg, pt, last), shift = gpu_index.meta()
    assert single["META"] == ["{%d,%d,%d,%d,%d}" % (pc, pg, pt, last, shift)] and single["OPEN"] == ["ok"] and single["OPEN_MISSING"] == ["error_tuple"]
    assert single["AFTER_FREE"] == ["resources 0 dtors 1 terms 0"] and single["DONE"] == [""]
    bases, off = lib.pack_reads(reads)
    out = gpu_index.align_batch(bases, off)
    refseqs = gpu_index.ref1_gather(out["win_off"], out["win_len"])
    expect = []
    for i in range(len(reads)):
        st = int(out["read_status"][i])
        if st == 1:
            expect.append("nothing")
        elif st == -1:
            expect.append("{error,bad_base}")
        elif st == -2:
            expect.append("{error,window_out_of_range}")
        else:
            hits = ['{{%d,"%s"},%d,"%s"}' % (int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(), int(out["pos"][h]),
                                             refseqs[h].decode()) for h in range(int(out["hit_off"][i]), int(out["hit_off"][i + 1]))]
            expect.append("[" + ",".join(hits) + "]" if hits else "[]")
    assert [v for _, v in lists["R"]] == expect
    assert sum(1 for v in expect if v.startswith("[{")) > 300 and "{error,bad_base}" in expect
    flat = [x for k in range(len(reads) // 200 + 2) for x in lists.get("W%d" % k, [])]
    assert flat == lists["R"] and lists["T"] == lists["R"]
