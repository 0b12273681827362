"""CPU: the C-ABI library builds, loads and exports every symbol include/kart_amd.h declares;
without a GPU the entry points refuse to run instead of falling back to a CPU path."""
import ctypes as C
import os
import re

from conftest import ROOT, SMALL_PREFIX


def test_header_symbols_are_exported(built_lib):
    from kart_amd import api
    hdr = open(os.path.join(ROOT, "include", "kart_amd.h")).read()
    declared = set(re.findall(r"\b(kg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.ABI_SYMBOLS), declared ^ set(api.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(built_lib, name), name


def test_host_library_symbols_are_exported(built_lib):
    """libkart_host.so (the host pipeline as a library) exports what include/kart_host.h declares"""
    from kart_amd import api
    hdr = open(os.path.join(ROOT, "include", "kart_host.h")).read()
    declared = set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.HOST_ABI_SYMBOLS), declared ^ set(api.HOST_ABI_SYMBOLS)
    lib = api.load_host_library()
    for name in declared:
        assert hasattr(lib, name), name
    if api.device_count() <= 0:       # no device: the session refuses to open (no CPU path behind it either)
        h = C.c_void_p()
        assert lib.kh_open(SMALL_PREFIX.encode(), 0, 2, C.byref(h)) != 0 and not h.value
        assert b"no HIP device" in lib.kh_last_error()


def test_no_cpu_fallback_without_device(built_lib):
    from kart_amd import api
    if api.device_count() > 0:
        return
    h = C.c_void_p()
    rc = built_lib.kg_index_load(SMALL_PREFIX.encode(), 0, 0, C.byref(h))
    assert rc == 1 and not h.value          # KG_ERR_NO_DEVICE
    assert b"no HIP device" in built_lib.kg_last_error()


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kart_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc", ".c", ".cc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f
