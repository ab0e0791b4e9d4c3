"""c_src/bwt_nif.c cannot be built here (no Erlang/OTP, no erl_nif.h).  What can be checked without a GPU and without
OTP: the shim type-checks (gcc -fsyntax-only, all warnings as errors) against include/bwt_gpu.h and a declaration-only
list of the NIF API functions it calls (tests/nif_check/erl_nif.h -- test infrastructure, not OTP's header), and every
libbwtgpu function it calls is exported by the library."""
import os
import re
import subprocess

from conftest import ROOT


def test_shim_type_checks():
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "nif_check"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "c_src", "bwt_nif.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shim_calls_only_exported_symbols():
    from bwt_amd import lib
    src = open(os.path.join(ROOT, "c_src", "bwt_nif.c")).read()
    called = set(re.findall(r"\b(bwt_[a-z0-9_]+)\s*\(", src)) - {"bwt_nif"}
    assert called and called <= set(lib.EXPORTS), called - set(lib.EXPORTS)


def test_shim_serialises_its_small_batch_handle():
    """every use of the handle's bwt_small (not re-entrant) sits between enif_mutex_lock and enif_mutex_unlock"""
    src = open(os.path.join(ROOT, "c_src", "bwt_nif.c")).read()
    body = src[src.index("static int align_staged"):src.index("static ERL_NIF_TERM nif_align_batch")]
    lock, unlock = body.index("enif_mutex_lock(r->lock)"), body.index("enif_mutex_unlock(r->lock)")
    uses = [m.start() for m in re.finditer(r"r->small|\bsr\.", body)]        # the handle, and the result pointers it owns
    assert uses and all(lock < u < unlock for u in uses)
    rest = src[src.index("static ERL_NIF_TERM nif_align_batch"):]
    assert "r->small" not in rest
