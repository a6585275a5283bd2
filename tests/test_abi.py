"""The C-ABI library loads and exports every symbol include/nohuman_engine.h declares; on a box
without a GPU every entry fails loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "nohuman_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nh_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from nohuman_amd import _lib
    declared = _declared_symbols()
    assert declared, "no symbols parsed from the header"
    assert sorted(_lib.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    from nohuman_amd import _lib
    L = _lib.lib()  # raises if the .so is missing or lacks a symbol
    for name in _declared_symbols():
        assert getattr(L, name) is not None
    assert L.nh_abi_version() == 5


def test_struct_layouts_match_header():
    from nohuman_amd import _lib
    from nohuman_amd.engine import RESULT_DTYPE
    assert C.sizeof(_lib.nh_result) == 16 == RESULT_DTYPE.itemsize
    assert C.sizeof(_lib.nh_stats) == 48
    assert C.sizeof(_lib.nh_options) == 16
    assert C.sizeof(_lib.nh_db_info) == 96
    assert C.sizeof(_lib.nh_db_check) == 32
    assert C.sizeof(_lib.nh_run_args) == 7 * 8 + 8 + 4 * 3 + 4 + 8 + 8


def _has_gpu():
    from nohuman_amd import _lib
    n = C.c_int(0)
    return _lib.lib().nh_device_count(C.byref(n)) == 0 and n.value > 0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_a_device(toy):
    """Product path must fail loudly when no gfx950 device is present."""
    import nohuman_amd
    from nohuman_amd import Engine, EngineError
    with pytest.raises(EngineError) as ei:
        nohuman_amd.probe()
    assert ei.value.code == -4  # NH_EDEVICE
    ob, tb, hb, _, _ = toy
    with pytest.raises(EngineError) as ei:
        Engine.from_images(ob, tb, hb)
    assert ei.value.code == -4
    with pytest.raises(EngineError):
        Engine.open(os.path.join(ROOT, "tests", "golden", "toy_db"))
    assert nohuman_amd.CommandRunner("kraken2").is_executable() is False


def test_product_does_not_import_the_oracle():
    """nothing under nohuman_amd/ or include/ may reference oracle/ (test infrastructure)."""
    for d, _, files in os.walk(os.path.join(ROOT, "nohuman_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in txt.lower().replace("/root/repo/oracle", ""), os.path.join(d, f)
