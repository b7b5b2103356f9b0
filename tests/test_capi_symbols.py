"""CPU: the product library loads and exports every symbol include/pymes_amd.h declares
(no compute calls: there is no GPU here)."""
import os
import re

from pymes_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "pymes_amd.h")).read()
    declared = set(re.findall(r"\b(pymes_[a-zA-Z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.Library()                      # libpymes_amd.so (HIP); raises if missing
    assert lib.backend == "hip-gfx950"
    for name in declared:
        assert hasattr(lib.dll, name), name


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(_lib.PymesError, match="no CPU fallback"):
        _lib.Library(str(tmp_path / "nope.so"))


def test_library_override(tmp_path, monkeypatch):
    """PYMES_AMD_LIBRARY=<path>: the loader takes that build (a missing one fails loudly; the backend check still applies)."""
    import importlib
    import pytest
    monkeypatch.setenv("PYMES_AMD_LIBRARY", str(tmp_path / "other_build.so"))
    mod = importlib.reload(_lib)
    try:
        assert mod.DEFAULT_PATH.endswith("other_build.so")
        with pytest.raises(mod.PymesError, match="no CPU fallback"):
            mod.Library()
    finally:
        monkeypatch.delenv("PYMES_AMD_LIBRARY")
        importlib.reload(_lib)
