"""CPU checks of the C-ABI boundary: the built library loads and exports every symbol include/petschip.h declares."""
import os
import re

from petsc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(phip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_symbols("petschip.h")
    assert len(names) >= 50
    for n in names:
        assert hasattr(lib, n), "libpetschip.so lacks %s" % n
    assert sorted(s[0] for s in capi.SIGNATURES) == names


def test_no_oracle_in_product():
    """The product package must not import, link or call the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "petsc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in src.lower() or f == "__init__.py", "%s mentions the oracle" % f
