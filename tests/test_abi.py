"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
that include/dransac.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from differentiable_ransac_amd import _lib as L


def _declared():
    src = open(L.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "run python -m differentiable_ransac_amd.build"
    lib = L.lib()
    names = _declared()
    assert "dr_msac_score_f32" in names and len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_error_string():
    lib = L.lib()
    assert lib.dr_version() == 1
    assert isinstance(lib.dr_last_error(), bytes)


def test_bad_arguments_return_einval_without_touching_the_gpu():
    lib = L.lib()
    lib.dr_msac_score_f32.restype = ctypes.c_int
    rc = lib.dr_msac_score_f32(None, None, None, None, 1, 1, 1, None, None, None, None, None)
    assert rc == -1
    assert b"null" in lib.dr_last_error()
