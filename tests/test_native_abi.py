"""CPU: libdiagan_hip.so loads and exports every symbol include/diagan_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "diagan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(diagan_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_something():
    names = _declared()
    assert "diagan_ldr_scores_f64" in names and "diagan_last_error" in names


def test_library_loads_and_exports_all_symbols():
    from diagan import _native as nat
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing
    assert nat.lib().diagan_target_arch() == b"gfx950"
    assert nat.fn("diagan_abi_version")() >= 1


def test_binding_table_matches_header():
    from diagan import _native as nat
    import diagan.ops  # noqa: F401  (registers the op signatures)
    import diagan.trainer.compute_pr  # noqa: F401  (registers the precision/recall entry points)
    from diagan.trainer import distributed as D
    D._register_native()                 # the RCCL context entry points (csrc/comm.hip)
    declared = set(_declared()) - {"diagan_last_error", "diagan_target_arch"}
    assert declared == set(nat._SIGS), declared ^ set(nat._SIGS)


def test_missing_library_fails_loudly(monkeypatch):
    from diagan import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", "/nonexistent/libdiagan_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()
