"""The C-ABI library loads without a GPU and exports every symbol include/tirt.h declares
(no compute calls here)."""
import ctypes
import os
import re

from ti_raytrace_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tirt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tirt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    decl = declared_symbols()
    assert len(decl) >= 30
    assert sorted(_native.SIGNATURES) == decl


def test_library_exports_every_declared_symbol():
    lib = _native.lib()            # (through the package's loader: a bare CDLL here would map the system HIP runtime beside the one PyTorch ships)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.tirt_version.restype = ctypes.c_int
    assert lib.tirt_version() >= 100


def test_no_cpu_fallback_in_product_path():
    """The product package never references the oracle."""
    pkg = os.path.join(ROOT, "ti_raytrace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "oracle_api" not in src, f
