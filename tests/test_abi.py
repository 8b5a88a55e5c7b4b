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


def test_restated_callers_live_in_the_harness_library():
    """The restated KSPSolve_CG / KSPGMRESCycle / KSPConvergedDefault are test and benchmark harness: libpetschip_host.so (product: types,
    ops tables, assembly, PCJACOBI) exports no KSP entry point, libpetschip_harness.so exports them and is built on the host library's
    public header alone."""
    import ctypes as C
    from petsc_amd import petsc as P
    P.lib()
    host = C.CDLL(P.HOST_LIB_PATH)
    harness = C.CDLL(P.HARNESS_LIB_PATH)
    for n in ("KSPSolve", "KSPCreate", "KSPSetUp", "KSPDestroy"):
        assert not hasattr(host, n), "libpetschip_host.so exports %s" % n
        assert hasattr(harness, n), "libpetschip_harness.so lacks %s" % n
    for n in ("PCApply", "PCSetUp", "MatMult", "VecMDot", "PetscHipSetError"):
        assert hasattr(host, n)
    src = open(os.path.join(ROOT, "harness", "ksp_restated.cpp")).read()
    incs = re.findall(r'#include\s+"([^"]+)"', src)
    assert incs == ["ksp_restated.h"], incs            # no internal header of the product
    assert "phip_" not in re.sub(r"/\*.*?\*/|//[^\n]*", "", src, flags=re.S)   # and no kernel-library call: public Vec*/Mat*/PC* only
    host_src = open(os.path.join(ROOT, "petsc_amd", "csrc", "host", "petsc_host.cpp")).read()
    host_code = re.sub(r"/\*.*?\*/|//[^\n]*", "", host_src, flags=re.S)   # comments may cite the callers
    assert "KSPSolve_CG" not in host_code and "KSPGMRESCycle" not in host_code and "KSPConvergedDefault" not in host_code
