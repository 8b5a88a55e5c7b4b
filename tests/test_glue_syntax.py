"""The PETSc-side glue (integration/petsc/petschip_glue.c: MatCreate_SeqAIJHIP/MPIAIJHIP, VecCreate_*HIP, PetscDLLibraryRegister_*) is
PARSED AND TYPE-CHECKED by a C compiler against the PETSc 3.6.1 headers of the reference tree -- public, private (vecimpl.h, matimpl.h)
and the AIJ/Vec implementation headers -- with the hand-written configuration stub of tests/native/petscconf_stub (real, double, 32-bit
indices, MPIUNI).  gcc -fsyntax-only: nothing is compiled to an object, linked or run; PETSc itself cannot be configured in this image
(DESIGN.md "Oracle").  What this pins: every struct member the glue touches (a->Mvctx->todata, Vec_MPI embedding, compressedrow.*,
B->ops->getvecs ...), every ops-table slot's function type, the pointer types handed to libpetschip (PetscInt* as int32_t*,
PetscMPIInt* procs), and that the glue calls no routine the headers do not declare.  CPU-only; skipped where the reference tree is absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GLUE = os.path.join(ROOT, "integration", "petsc", "petschip_glue.c")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")) or shutil.which("gcc") is None,
                                reason="needs the reference headers and gcc")


def _parse(extra):
    cmd = ["gcc", "-fsyntax-only", "-std=gnu99", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-sign-compare",
           "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration", "-Werror=int-conversion",
           "-I" + os.path.join(ROOT, "tests", "native", "petscconf_stub"), "-I" + os.path.join(REF, "include"),
           "-I" + os.path.join(REF, "include", "petsc", "mpiuni"), "-I" + REF, "-I" + os.path.join(ROOT, "include"), *extra, GLUE]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("extra", [[], ["-DPETSC_USE_DEBUG=1"]], ids=["optimized", "debug"])
def test_glue_parses_against_reference_headers(extra):
    r = _parse(extra)
    mine = [l for l in r.stderr.splitlines() if "petschip_glue.c" in l and ("error" in l or "warning" in l)]
    assert r.returncode == 0 and not mine, r.stderr[-4000:]


def test_glue_exports_the_boundary_symbols():
    """SURVEY.md 8(b) "must export": the constructors and the loader entry point are defined with PETSC_EXTERN, and the loader's name
    rule (dl.c:175-191: lib<basename>.so -> PetscDLLibraryRegister_<basename>) is met for both file names the glue may be installed under."""
    src = open(GLUE).read()
    for name in ("MatCreate_SeqAIJHIP", "MatCreate_MPIAIJHIP", "VecCreate_SeqHIP", "VecCreate_MPIHIP", "VecCreate_HIP",
                 "PetscDLLibraryRegister_petschip_glue", "PetscDLLibraryRegister_petschip"):
        assert re.search(r"PETSC_EXTERN\s+PetscErrorCode\s+%s\s*\(" % name, src), name
    for t in ('"seqaijhip"', '"mpiaijhip"', '"aijhip"', '"seqhip"', '"mpihip"', '"hip"'):
        assert t in src


def test_glue_calls_only_public_petsc_symbols():
    """An out-of-tree plug-in can link only PETSC_EXTERN symbols: the PETSC_INTERN (hidden) or file-static routines of the Seq/MPI types
    must be reached through the ops table the base constructor filled, never by name."""
    src = re.sub(r"/\*.*?\*/", "", open(GLUE).read(), flags=re.S)
    for hidden in ("VecDestroy_Seq", "VecDestroy_MPI", "VecCreate_Seq_Private", "VecCreate_MPI_Private", "VecPlaceArray_Seq", "VecPlaceArray_MPI",
                   "VecResetArray_Seq", "VecResetArray_MPI", "VecReplaceArray_Seq", "MatDestroy_SeqAIJ", "MatDestroy_MPIAIJ",
                   "MatAssemblyEnd_SeqAIJ", "MatAssemblyEnd_MPIAIJ"):
        assert not re.search(r"\b%s\s*\(" % hidden, src), hidden


def test_every_phip_call_of_the_glue_is_declared():
    """the glue binds exactly the C ABI: every phip_* it calls is declared in include/petschip.h (and so exported, tests/test_abi.py)"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "petschip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(phip_[a-z0-9_]+)\s*\(", hdr))
    used = set(re.findall(r"\b(phip_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(GLUE).read(), flags=re.S)))
    assert used and used <= declared, used - declared


def _petsc_extern_symbols():
    """names declared PETSC_EXTERN (functions and data) in the reference's public / private headers and the AIJ / Vec implementation
    headers and sources the constructors chain to; MPIUNI's Petsc_MPI_* come from its mpi.h"""
    names = set()
    pats = [os.path.join(REF, "include"), os.path.join(REF, "src", "mat", "impls", "aij"), os.path.join(REF, "src", "vec", "vec", "impls"),
            os.path.join(REF, "src", "sys", "objects"), os.path.join(REF, "src", "sys", "logging")]
    for top in pats:
        for root, _, files in os.walk(top):
            for f in files:
                if not f.endswith((".h", ".c")):
                    continue
                txt = open(os.path.join(root, f), errors="ignore").read()
                for m in re.finditer(r"PETSC_(?:EXTERN|VISIBILITY_PUBLIC)\s+([^;{]*?)[;{]", txt, flags=re.S):
                    decl = m.group(1)
                    pm = re.search(r"\(\s*\*\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)", decl)   # a function pointer: (*PetscTrMalloc)(...)
                    fm = re.search(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", decl)
                    if pm:
                        names.add(pm.group(1))
                    elif fm:
                        names.add(fm.group(1))
                    else:
                        names.update(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\b", decl)[-1:])
                if f == "mpi.h":
                    names.update(re.findall(r"\b(Petsc_MPI_[A-Za-z_]+|MPIUNI_[A-Za-z_]+)\b", txt))
    return names


def test_glue_object_links_only_against_the_two_boundaries(tmp_path):
    """Compiled to an object (same headers and stub as the parse test), the glue's undefined symbols are exactly: phip_* entry points
    declared in include/petschip.h, symbols the reference declares PETSC_EXTERN (what an out-of-tree library can link), and libc/libm.
    Anything else -- a hidden PETSc routine, a helper of the stand-alone mirror -- would fail at dlopen time inside a real PETSc."""
    obj = str(tmp_path / "glue.o")
    cmd = ["gcc", "-c", "-std=gnu99", "-O1", "-fPIC", "-I" + os.path.join(ROOT, "tests", "native", "petscconf_stub"), "-I" + os.path.join(REF, "include"),
           "-I" + os.path.join(REF, "include", "petsc", "mpiuni"), "-I" + REF, "-I" + os.path.join(ROOT, "include"), GLUE, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.split()
    und = {u for u in und if u not in ("U", "w")}
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "petschip.h")).read(), flags=re.S)
    phip = set(re.findall(r"\b(phip_[a-z0-9_]+)\s*\(", hdr))
    libc = {"sqrt", "memcpy", "memset", "strcmp", "strlen", "__stack_chk_fail", "_GLOBAL_OFFSET_TABLE_", "fabs", "snprintf", "getenv", "atoi"}
    petsc = _petsc_extern_symbols()
    stray = sorted(u for u in und if u not in phip and u not in libc and u not in petsc)
    assert not stray, stray
    assert any(u.startswith("phip_") for u in und) and "MatCreate_SeqAIJ" in und and "VecCreate_MPI" in und
    defined = subprocess.run(["nm", "--defined-only", "-g", obj], capture_output=True, text=True, check=True).stdout
    for name in ("MatCreate_SeqAIJHIP", "MatCreate_MPIAIJHIP", "VecCreate_SeqHIP", "VecCreate_MPIHIP", "VecCreate_HIP", "PetscDLLibraryRegister_petschip"):
        assert re.search(r"\bT %s\b" % name, defined), name
