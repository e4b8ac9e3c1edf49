"""CPU tier: the product library (gfx950 build, cross-compiled here) loads without a GPU and exports exactly the C ABI that
include/humor_amd.h declares; the ctypes binding covers every entry point.  No compute call is made."""
import ctypes
import os
import re

from conftest import ROOT
from humor_amd import _lib, build


def _header_functions():
    text = open(os.path.join(ROOT, 'include', 'humor_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    return sorted(set(re.findall(r'\b(ha_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_bound_entry_points():
    declared = _header_functions()
    assert declared, 'no ha_* functions found in include/humor_amd.h'
    assert sorted(_lib._SIGS) == declared


def test_product_library_exports_every_declared_symbol():
    path = build.build()                      # no-op when the in-tree build is current
    assert os.path.exists(path)
    dll = ctypes.CDLL(path)
    for name in _header_functions():
        assert hasattr(dll, name), f'{name} is declared in include/humor_amd.h but not exported by {path}'
    dll.ha_abi_version.restype = ctypes.c_int
    assert dll.ha_abi_version() == _lib.ABI_VERSION
    lib = _lib.Lib(path)
    assert lib.missing == []
