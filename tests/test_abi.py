"""CPU: the C-ABI library builds/loads and exports every symbol include/gomatching_hip.h declares (no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gomatching_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gom_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    import __graft_entry__ as entry
    entry.build()
    from gomatching_amd import lib
    handle = lib.load()
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(handle, n), "symbol %s declared in the header but not exported" % n
        assert n in lib.SIGNATURES, "symbol %s has no ctypes signature" % n
    for n in lib.SIGNATURES:
        assert n in names, "bound symbol %s is not declared in include/gomatching_hip.h" % n
    assert handle.gom_abi_version() == 1


def test_product_has_no_oracle_dependency():
    """The shipped package must never import the oracle (or /root/reference)."""
    pkg = os.path.join(ROOT, "gomatching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "smoke.py":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
                assert "/root/reference" not in re.sub(r'""".*?"""', "", src, flags=re.S), f


def test_missing_library_fails_loudly(monkeypatch):
    from gomatching_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libgomatching_hip.so")
    with pytest.raises(lib.GomError):
        lib.load()
