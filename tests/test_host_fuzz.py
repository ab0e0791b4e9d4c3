"""The host-side file parsers (.fm external term format, FASTQ -- sequential and threaded) and the wire-format code of the host
boundary (the 2-bit packer, the compact-table expansion) under AddressSanitizer + UBSan on a seeded stream of mutated inputs:
every input is parsed or refused, nothing is read or written out of bounds, the packer's output equals a byte-by-byte restatement.  CPU only (sanitizers
are not available on the GPU pool); built from the library's own sources."""
import os
import subprocess

import pytest

from conftest import FM_PATH, ROOT


def test_parsers_under_asan(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    src = [os.path.join(ROOT, "tests", "fuzz", "host_fuzz.cpp"), os.path.join(ROOT, "bwt_amd", "csrc", "etf_fm.cpp"),
           os.path.join(ROOT, "bwt_amd", "csrc", "fastq.cpp"), os.path.join(ROOT, "bwt_amd", "csrc", "pack.cpp")]
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                        "-o", exe] + src, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in r.stderr.lower() and "cannot find" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2):
        out = subprocess.run([exe, FM_PATH, "120", str(seed)], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, (out.stdout[-500:], out.stderr[-4000:])
        assert "refused" in out.stdout and "packer:" in out.stdout and "table expansion:" in out.stdout
