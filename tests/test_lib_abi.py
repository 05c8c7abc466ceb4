"""CPU-side checks of the drop-in boundary: libpimgpu.so loads without a GPU and exports every
symbol include/pimgpu.h declares; no compute call is made here."""
import ctypes
import os
import re

import pytest

from pim_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "pimgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pimgpu_[a-z_0-9]+)\s*\(", text)))


def test_header_and_python_mirror_agree():
    assert header_functions() == sorted(api.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    for name in header_functions():
        assert hasattr(lib, name), f"libpimgpu.so does not export {name}"


def test_struct_sizes_match_the_header():
    # sizes the header documents (PtRayHit 48 B, PtResult 36 B, Material 48 B)
    assert ctypes.sizeof(api.PimMaterial) == 48
    assert api.RAYHIT_DTYPE.itemsize == 48
    assert api.RESULT_DTYPE.itemsize == 36
    assert ctypes.sizeof(api.PimCamera) == 48
    assert ctypes.sizeof(api.PimDofInfo) == 28
    assert ctypes.sizeof(api.PimTraceParams) == 24


def test_calls_before_init_fail_loudly():
    lib = api.load_library()
    out = ctypes.c_void_p()
    rc = lib.pimgpu_trace_create(ctypes.c_int32(4), ctypes.c_int32(4), ctypes.byref(out))
    assert rc == -4 and b"pimgpu_init" in lib.pimgpu_last_error()


def test_product_never_touches_the_oracle():
    """the product path must not import, link or load anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pim_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liborc" not in text and "oracle/" not in text and "pim_oracle" not in text, os.path.join(dirpath, f)
