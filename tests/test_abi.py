"""The C-ABI library loads and exports every symbol include/mcx.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    path = os.path.join(ROOT, "mapcaller_amd", "libmcx.so")
    if not os.path.exists(path):
        g.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_lists_the_same_symbols():
    from mapcaller_amd import api
    assert sorted(api.SYMBOLS) == _declared()


def test_binding_fails_loudly_without_gpu_or_index():
    from mapcaller_amd import api
    import pytest
    with pytest.raises(api.McxError):
        api.Index("/nonexistent/prefix")
