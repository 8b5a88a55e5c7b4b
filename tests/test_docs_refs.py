"""The documents name scripts and committed measurements; a name that no longer exists is a stale claim."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]
# named on purpose although gone: profiles/README.md says that round 3 deleted them with their kernel
DELETED = {"scripts/exp_rmat_lds.py", "scripts/pmc_rmat.sh"}


def _text(name):
    with open(os.path.join(ROOT, name), encoding="utf-8") as f:
        return f.read()


def test_scripts_named_in_the_documents_exist():
    missing = []
    for doc in DOCS:
        for s in sorted(set(re.findall(r"scripts/[A-Za-z0-9_/]+\.(?:py|sh|hip)", _text(doc)))):
            if s not in DELETED and not os.path.exists(os.path.join(ROOT, s)):
                missing.append((doc, s))
    assert not missing, missing


def test_measurement_files_named_in_the_documents_exist():
    have = set()
    for _, _, files in os.walk(os.path.join(ROOT, "profiles")):
        have.update(files)
    missing = []
    for doc in DOCS:
        for f in sorted(set(re.findall(r"\br0[123]_[A-Za-z0-9_]+\.(?:jsonl|json|csv|txt)\b", _text(doc)))):
            if f not in have:
                missing.append((doc, f))
    assert not missing, missing


def test_c_abi_names_in_the_documents_are_declared():
    header = _text(os.path.join("include", "petschip.h")) + _text(os.path.join("include", "petschip_host.h"))
    internal = {"phip_check_device_error"}  # petsc_amd/csrc/phip_internal.h, named in DESIGN.md as such
    missing = []
    for doc in DOCS[:3]:
        for n in sorted(set(re.findall(r"\bphip_[a-z0-9_]+\b(?!\.(?:hip|h|cpp)\b)", _text(doc)))):
            if n.endswith("_") or n in internal:
                continue
            if not re.search(r"\b%s\b" % re.escape(n), header):
                missing.append((doc, n))
    assert not missing, missing
