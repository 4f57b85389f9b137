"""CPU suite, part 2: the C-ABI library loads and exports every symbol that
include/pcd_engine.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from fenapack_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(ROOT, "include", "pcd_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcd_[a-z_0-9]+)\s*\(", text)))


def test_binding_covers_the_header():
    assert sorted(_cabi.DECLARED_SYMBOLS) == _declared_in_header()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_cabi.HIP_LIBRARY_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_cabi.HIP_LIBRARY_PATH)
    for name in _declared_in_header():
        assert hasattr(lib, name), name


def test_product_fails_loudly_without_the_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_cabi, "_hip_library", None)
    monkeypatch.setattr(_cabi, "HIP_LIBRARY_PATH",
                        str(tmp_path / "libpcd_hip.so"))
    with pytest.raises(_cabi.EngineError):
        _cabi.hip_library()


def test_oracle_is_not_reachable_from_the_product():
    """No module of the product package may import the test oracle."""
    pkg = os.path.join(ROOT, "fenapack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src,
                                     flags=re.M), f
