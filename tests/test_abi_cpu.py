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
    assert lib.mvae_abi_version() == 1
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
