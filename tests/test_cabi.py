"""The C-ABI library loads without a GPU and exports every symbol include/dta.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    import torch  # noqa: F401  - torch's own libamdhip64 must be resident first (as _lib.lib() arranges): ONE HIP runtime per process
    from dynamictreeattn_amd.build import build_native
    lib = ctypes.CDLL(build_native())
    header = open(os.path.join(ROOT, "include", "dta.h")).read()
    declared = set(re.findall(r"^int\s+(dta_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dta_version() >= 100
    from dynamictreeattn_amd import _lib
    assert set(_lib.EXPORTS) <= declared | {"dta_version"}
    _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dynamictreeattn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
