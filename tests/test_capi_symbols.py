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


def test_library_override(tmp_path):
    """PYMES_AMD_LIBRARY=<path>: the loader takes that build (a missing one fails loudly; the backend check still applies).
    In a process of its own: the variable is read when the module is imported."""
    import subprocess
    import sys
    code = ("from pymes_amd import _lib\n"
            "assert _lib.DEFAULT_PATH.endswith('other_build.so'), _lib.DEFAULT_PATH\n"
            "try:\n    _lib.Library()\nexcept _lib.PymesError as e:\n    assert 'no CPU fallback' in str(e); print('refused')\n")
    env = dict(os.environ, PYMES_AMD_LIBRARY=str(tmp_path / "other_build.so"), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "refused" in out.stdout, out.stderr
