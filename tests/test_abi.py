"""The C-ABI library builds, loads, and exports every symbol include/mxdet.h declares (no compute here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mxdet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mxdet_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from mxdetection_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_hip(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) > 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in mxdet.h but not exported: %s" % missing
    unbound = [n for n in names if n not in _lib.SIGNATURES]
    assert not unbound, "declared in mxdet.h but not bound in _lib.SIGNATURES: %s" % unbound
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, "bound but not declared: %s" % extra


def test_version_and_error_string():
    from mxdetection_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.mxdet_version()
    assert lib.mxdet_last_error() == b""
    # argument validation happens on the host before any launch: no GPU needed
    rc = lib.mxdet_nms_batched(None, None, None, 1, 5000, 0.5, 10, None, None, None, 0, None)
    assert rc == -2 and b"n_max" in lib.mxdet_last_error()
    rc = lib.mxdet_box_iou(None, 4, None, 4, None, None)
    assert rc == -1 and b"null" in lib.mxdet_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mxdetection_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.MxdetError, match="no CPU fallback"):
        _lib.load()
