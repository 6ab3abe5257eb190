"""CPU-only: the C-ABI library loads and exports every symbol include/*.h
declare; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="dorylus_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dory_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    import dorylus_amd
    assert sorted(dorylus_amd.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(dorylus_amd.LIB_PATH)
    for s in _declared() + _declared("dorylus_host.h") + _declared("dorylus_wire.h"):
        assert hasattr(lib, s), s


def test_fails_loudly_without_gpu():
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(dorylus_amd.DoryError):
        dorylus_amd.Context(0)
