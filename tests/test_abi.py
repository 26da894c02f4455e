"""The C ABI boundary: libzkhip.so (the real gfx950 build) loads without a GPU and exports every function that
include/zkhip.h declares; the ctypes binding covers all of them; without a device the library fails loudly instead of
falling back to anything."""
import ctypes
import os
import re

import pytest

from zokrates_amd import build as zbuild
from zokrates_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    return zbuild.build_lib()          # hipcc cross-compiles for gfx950 without a GPU (incremental)


def test_header_symbols_are_exported(lib_path):
    names = declared_functions()
    assert len(names) >= 25
    dll = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing


def test_binding_covers_the_header():
    assert sorted(native.Library.SYMBOLS) == declared_functions()


def test_no_cpu_fallback(lib_path):
    """On a machine without a GPU the product library refuses to create a context (ZKHIP_ERR_DEVICE)."""
    lib = native.Library(lib_path)
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.ZkhipError) as e:
        native.Context(0, lib)
    assert e.value.code == -4
