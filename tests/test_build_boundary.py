"""The build boundary (SURVEY.md 8b "Build boundary"): integration/petsc/hip.py is the BuildSystem package behind --with-hip and
integration/petsc/WithHIP.cmake the OpenCMISS CMake switch.  Neither can run here (the BuildSystem needs Python 2, the CMake build Fortran +
MPI + BLAS), so they are checked the way the glue is: parsed, and every name they take from the reference looked up in the reference."""
import ast
import os
import py_compile
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIP_PY = os.path.join(ROOT, "integration", "petsc", "hip.py")
CMAKE = os.path.join(ROOT, "integration", "petsc", "WithHIP.cmake")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "config", "BuildSystem")), reason="needs the reference tree")


def test_hip_py_compiles_in_the_python2_python3_subset(tmp_path):
    py_compile.compile(HIP_PY, cfile=str(tmp_path / "hip.pyc"), doraise=True)
    tree = ast.parse(open(HIP_PY).read())
    for node in ast.walk(tree):
        # constructs Python 2 does not parse
        assert not isinstance(node, (ast.JoinedStr, ast.Nonlocal, ast.AnnAssign, ast.YieldFrom, ast.NamedExpr, ast.AsyncFunctionDef, ast.Await, ast.Starred)), ast.dump(node)[:80]
        if isinstance(node, ast.FunctionDef):
            assert not node.args.kwonlyargs and node.returns is None and all(a.annotation is None for a in node.args.args)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name):
            assert node.func.id != "print"   # a statement in Python 2
        if isinstance(node, ast.ExceptHandler):
            assert node.name is None         # `except X as e` vs `except X, e`
    src = open(HIP_PY).read()
    assert "from __future__ import generators" in src and "class Configure(config.package.Package)" in src


@needs_ref
def test_hip_py_reads_only_what_the_buildsystem_provides():
    """every attribute hip.py READS on self, and every base-class method it calls, exists in the reference's config/package.py /
    config/base.py / script.py / logger.py or is assigned by hip.py itself; the framework modules it requires exist; the options it
    declares are the ones it reads"""
    src = open(HIP_PY).read()
    tree = ast.parse(src)
    assigned, read = set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "self":
            (assigned if isinstance(node.ctx, ast.Store) else read).add(node.attr)
    own_methods = {n.name for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    base = ""
    for f in ("config/BuildSystem/config/package.py", "config/BuildSystem/config/base.py", "config/BuildSystem/script.py", "config/BuildSystem/logger.py",
              "config/BuildSystem/args.py"):
        base += open(os.path.join(REF, f), errors="ignore").read()
    provided = set(re.findall(r"self\.([A-Za-z_][A-Za-z0-9_]*)\s*=", base)) | set(re.findall(r"def ([A-Za-z_][A-Za-z0-9_]*)\(", base)) \
        | set(re.findall(r"^\s+([A-Za-z_][A-Za-z0-9_]*)\s*=\s*property\(", base, flags=re.M))
    missing = sorted(a for a in read if a not in assigned and a not in own_methods and a not in provided)
    assert not missing, missing
    # Package methods called through the class
    for m in re.findall(r"config\.package\.Package\.([A-Za-z_]+)\(self", src):
        assert re.search(r"def %s\(self" % m, open(os.path.join(REF, "config/BuildSystem/config/package.py")).read()), m
    # required framework modules exist
    for mod in re.findall(r"framework\.require\('([^']+)'", src):
        path = os.path.join(REF, "config", "BuildSystem", *mod.split(".")) + ".py" if mod.startswith("config.") else os.path.join(REF, "config", *mod.split(".")) + ".py"
        assert os.path.exists(path), mod
    # the attributes it reads on those modules exist there
    for mod, attrs in (("config/PETSc/options/scalarTypes.py", ("scalartype", "precision")), ("config/PETSc/options/indexTypes.py", ("integerSize",)),
                       ("config/BuildSystem/config/headers.py", ("toString",)), ("config/BuildSystem/config/compilers.py", ("CPPFLAGS",))):
        txt = open(os.path.join(REF, mod)).read()
        for a in attrs:
            assert re.search(r"\b%s\b" % a, txt), (mod, a)
    # what the generic consistency checks need from this package (package.py:648-653) and what PETSc/Configure.py hangs on it (:90-101)
    for a in ("double", "complex", "requires32bitint", "functions", "includes", "liblist"):
        assert a in assigned, a
    assert "indexProvider" in open(os.path.join(REF, "config/BuildSystem/config/package.py")).read()
    assert set(re.findall(r"'-with-(hip-[a-z]+)=", src)) == set(re.findall(r"argDB\['with-(hip-[a-z]+)'\]", src))
    # the library functions it asks configure to link are entry points of the C ABI
    hdr = open(os.path.join(ROOT, "include", "petschip.h")).read()
    fl = re.search(r"self\.functions\s*=\s*\[([^\]]*)\]", src).group(1)
    for fn in re.findall(r"'([a-z_0-9]+)'", fl):
        assert re.search(r"\bint\s+%s\s*\(" % fn, hdr), fn


@needs_ref
def test_cmake_fragment_names_what_the_reference_build_has():
    txt = open(CMAKE).read()
    assert re.search(r"option\(WITH_HIP\b", txt) and "PETSC_HAVE_HIP" in txt and "petschip_glue.c" in txt
    code = "\n".join(l.split("#")[0] for l in txt.splitlines())
    assert len(re.findall(r"(?<![a-z])if\s*\(", code)) == len(re.findall(r"endif\s*\(", code))
    ref = open(os.path.join(REF, "CMakeLists.txt")).read() + open(os.path.join(REF, "cmake", "OpenCMISSPETScConfig.cmake")).read()
    for var in ("PETSCCONF_HAVE_FLAGS", "PETSC_PACKAGE_LIBS", "PETSC_CONFIGINFO_STRING", "PETSCMAT_SRCS", "PETSC_USE_COMPLEX", "PETSC_USE_REAL_SINGLE"):
        assert var in txt and var in ref, var
    # the CMake build never sets 64-bit indices; the name guarded against is petscconf.h's (config/PETSc/options/indexTypes.py)
    assert "PETSC_USE_64BIT_INDICES" in txt and "USE_64BIT_INDICES" in open(os.path.join(REF, "config/PETSc/options/indexTypes.py")).read()
    assert "option(WITH_CUDA" in ref   # the option it goes beside


def test_cmake_fragment_runs(tmp_path):
    """the fragment itself, run by cmake in a scratch project: WITH_HIP=ON + PETSCHIP_DIR = this repository finds the header, the built
    library and the glue source, sets PETSC_HAVE_HIP and the petscconf.h lines; a wrong architecture or 64-bit indices stop the configure"""
    import shutil
    import subprocess
    if shutil.which("cmake") is None:
        pytest.skip("no cmake")
    if not os.path.exists(os.path.join(ROOT, "petsc_amd", "lib", "libpetschip.so")):
        pytest.skip("libpetschip.so not built")
    (tmp_path / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.10)\nproject(PETSc NONE)\nset(PETSCCONF_HAVE_FLAGS )\nset(PETSC_PACKAGE_LIBS )\nset(PETSC_CONFIGINFO_STRING )\n"
        "include(%s)\nmessage(STATUS \"HAVE=${PETSC_HAVE_HIP}\")\nmessage(STATUS \"GLUE=${PETSCHIP_GLUE_SOURCES}\")\n"
        "message(STATUS \"LIBS=${PETSC_PACKAGE_LIBS}\")\nstring(REPLACE \";\" \"\" F \"${PETSCCONF_HAVE_FLAGS}\")\nfile(WRITE ${CMAKE_BINARY_DIR}/flags.h \"${F}\")\n" % CMAKE)

    def run(*defs):
        b = tmp_path / ("b%d" % len(list(tmp_path.iterdir())))
        return b, subprocess.run(["cmake", "-S", str(tmp_path), "-B", str(b), "-DPETSCHIP_DIR=" + ROOT, *defs], capture_output=True, text=True)

    b, r = run("-DWITH_HIP=ON")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "HAVE=TRUE" in r.stdout and "petschip_glue.c" in r.stdout and "libpetschip.so" in r.stdout
    flags = (b / "flags.h").read_text()
    assert "#define PETSC_HAVE_HIP 1" in flags and '#define PETSC_HIP_ARCH "gfx950"' in flags
    b, r = run("-DWITH_HIP=OFF")
    assert r.returncode == 0 and "HAVE=TRUE" not in r.stdout
    _, r = run("-DWITH_HIP=ON", "-DPETSC_HIP_ARCH=gfx90a")
    assert r.returncode != 0 and "gfx950" in r.stderr
    _, r = run("-DWITH_HIP=ON", "-DPETSC_USE_64BIT_INDICES=YES")
    assert r.returncode != 0 and "32-bit" in r.stderr
