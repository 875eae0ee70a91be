"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/dsr_hip.h declares, with the argument counts the ctypes binding assumes.  No compute calls."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "deep-super-resolution_amd"


@pytest.fixture(scope="module")
def so():
    b = importlib.import_module(PKG + "._build")
    return b.build()


def declared():
    src = open(os.path.join(ROOT, "include", "dsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(dsr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_symbols_exported(so):
    lib = ctypes.CDLL(so)
    decl = declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/dsr_hip.h but not exported"


def test_ctypes_signatures_match_header(so):
    L = importlib.import_module(PKG + "._lib")
    decl = declared()
    assert set(L.SIGNATURES) == set(decl), set(L.SIGNATURES) ^ set(decl)
    for name, (_, args) in L.SIGNATURES.items():
        assert len(args) == decl[name], (name, len(args), decl[name])
    L.lib()
    assert L.lib().dsr_abi_version() == 1


def test_host_side_descriptor_checks(so):
    """Argument validation runs on the host before any launch, so it is testable without a GPU."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    oh, ow = ctypes.c_int(), ctypes.c_int()
    d = L.ConvDesc(L.BF16, 2, 16, 16, 64, 64, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_out_size(ctypes.byref(d), ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oh.value, ow.value) == (8, 8)
    assert lib.dsr_conv_stats_rows(ctypes.byref(d)) == 1
    assert lib.dsr_conv_packed_elems(ctypes.byref(d), 0) == 9 * 64 * 64
    bad = L.ConvDesc(L.BF16, 2, 16, 16, 64, 64, 11, 11, 1, 5, 0)      # 121 taps > 96
    assert lib.dsr_conv_out_size(ctypes.byref(bad), ctypes.byref(oh), ctypes.byref(ow)) < 0
    assert b"taps" in lib.dsr_last_error()
    refl = L.ConvDesc(L.BF16, 1, 1, 1, 8, 8, 3, 3, 1, 1, 1)           # reflect pad >= size: torch raises too
    assert lib.dsr_conv_out_size(ctypes.byref(refl), ctypes.byref(oh), ctypes.byref(ow)) < 0
    with pytest.raises(RuntimeError):
        L.check(-1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    pk = os.path.join(ROOT, PKG)
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    gen = importlib.import_module(PKG + ".models.GAN.generator")
    g = gen.Generator(4, 1)
    with pytest.raises(RuntimeError):
        g(torch.zeros(1, 3, 8, 8))
