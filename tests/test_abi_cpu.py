"""The C-ABI library loads without a GPU and exports every symbol include/midivae_hip.h declares."""
import os
import re

import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import hiplib as hl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "midivae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvae_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(hl.LIB_PATH), "run `make -C midi-vae_amd/csrc` or __graft_entry__.build()"


def test_every_declared_symbol_is_exported_and_bound():
    lib = hl.load()
    declared = _declared_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
        assert name in hl.SIGNATURES, "binding misses %s" % name
    assert set(hl.SIGNATURES) == set(declared)
    assert lib.mvae_abi_version() == 9
    assert b"gfx950" in lib.mvae_build_info()


def test_argument_validation_needs_no_gpu():
    lib = hl.load()
    assert lib.mvae_rnn_fwd(None, None) == -1
    assert lib.mvae_gemm(None, None) == -1
    assert lib.mvae_head(None, None) == -1
    assert lib.mvae_head_np(61) == 64 and lib.mvae_head_np(16) == 16 and lib.mvae_head_np(1) == 16
    assert lib.mvae_head_np(1000) < 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(hl, "_lib", None)
    monkeypatch.setattr(hl, "LIB_PATH", "/nonexistent/libmidivae_hip.so")
    with pytest.raises(hl.HipLibraryMissing):
        hl.load()


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every argument struct of include/midivae_hip.h against its ctypes mirror: size and the offset of every field, from a
    C program compiled with the same header (a field added on one side only would silently shift everything after it)."""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"mvae_rnn_fwd_args": hl.RnnFwdArgs, "mvae_rnn_bwd_args": hl.RnnBwdArgs, "mvae_gemm_args": hl.GemmArgs,
             "mvae_head_args": hl.HeadArgs, "mvae_latent_fwd_args": hl.LatentFwdArgs, "mvae_latent_bwd_args": hl.LatentBwdArgs,
             "mvae_latent_chain_fwd_args": hl.LatentChainFwdArgs, "mvae_latent_chain_bwd_args": hl.LatentChainBwdArgs,
             "mvae_prep_job": hl.PrepJob, "mvae_xpand_args": hl.XpandArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "midivae_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, cls in pairs.items():
        assert got[(cname, "sizeof")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_problem_arrays_of_the_multi_launches_are_plain_c_arrays():
    """mvae_rnn_fwd_multi / mvae_rnn_bwd_multi / mvae_gemm_kstream_multi take arrays of the argument structs: the ctypes array
    stride must be the C sizeof (no padding between problems), element i at i * sizeof"""
    import ctypes
    for cls in (hl.RnnFwdArgs, hl.RnnBwdArgs, hl.GemmArgs, hl.XpandArgs, hl.PrepJob):
        arr = (cls * 3)()
        assert ctypes.sizeof(arr) == 3 * ctypes.sizeof(cls)
        assert ctypes.addressof(arr[2]) - ctypes.addressof(arr[0]) == 2 * ctypes.sizeof(cls)
        assert ctypes.sizeof(cls) % 8 == 0, cls          # every struct holds pointers: 8-byte aligned, so arrays need no tail padding


def test_integration_snippet_matches_the_header():
    """INTEGRATION.md section 2 shows the binding a reference maintainer would write; its struct must be the header's (VERDICT r03:
    the snippet had fallen 48 bytes behind).  The snippet's class definition is executed and compared with hiplib.RnnFwdArgs,
    which test_ctypes_structs_match_the_header_layout pins to the header itself."""
    import ctypes
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("class RnnFwdArgs(C.Structure):"):]
    block = block[:block.index("\nlib.mvae_rnn_fwd.restype")]
    ns = {"C": ctypes}
    exec(block, ns)
    snip = ns["RnnFwdArgs"]
    assert ctypes.sizeof(snip) == ctypes.sizeof(hl.RnnFwdArgs)
    assert [(n, getattr(snip, n).offset) for n, _ in snip._fields_] == [(n, getattr(hl.RnnFwdArgs, n).offset) for n, _ in hl.RnnFwdArgs._fields_]
    # the plan example names real entry points / fields
    assert "mvae_plan_add_call" in text and hasattr(hl.RnnFwdArgs, "wait_value")
