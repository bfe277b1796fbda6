"""CPU: the C-ABI shared library loads without a GPU and exports every function that
include/*.h declares (and nothing C++-mangled leaks into the documented surface)."""
import ctypes
import os
import re

import deciphon_amd
from dcp_testlib import ROOT

DECL = re.compile(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b((?:dcp|viterbi)_[a-z0-9_]+)\s*\(", re.M)


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(deciphon_amd.library_path())
    headers = [os.path.join(ROOT, "include", h) for h in sorted(os.listdir(os.path.join(ROOT, "include")))
               if h.endswith(".h")]
    assert headers
    total = 0
    for h in headers:
        names = declared_functions(h)
        assert names, h
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {os.path.basename(h)} but not exported"
            total += 1
    assert total >= 70


def test_device_count_needs_no_gpu():
    assert deciphon_amd.device_count() >= 0
