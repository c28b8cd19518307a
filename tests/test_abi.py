"""The C-ABI library builds, loads, and exports every symbol include/mxdet.h declares (no compute here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="mxdet.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mxdet_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from mxdetection_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_hip(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    debug = _declared("mxdet_debug.h")
    assert len(names) > 30
    assert not [n for n in names if n.startswith("mxdet_debug_")], "debug hooks belong in mxdet_debug.h"
    assert sorted(debug) == sorted(_lib.DEBUG_SYMBOLS)
    missing = [n for n in names + debug if not hasattr(lib, n)]
    assert not missing, "declared in a header but not exported: %s" % missing
    unbound = [n for n in names + debug if n not in _lib.SIGNATURES]
    assert not unbound, "declared in a header but not bound in _lib.SIGNATURES: %s" % unbound
    extra = [n for n in _lib.SIGNATURES if n not in names and n not in debug]
    assert not extra, "bound but not declared: %s" % extra
    for n in ("mxdet_comm_unique_id", "mxdet_comm_create", "mxdet_comm_destroy", "mxdet_allreduce_bucket",
              "mxdet_comm_wait", "mxdet_comm_broadcast"):
        assert n in names          # the gradient exchange is part of the boundary


def test_library_does_not_read_the_environment():
    """'No global mutable state' hygiene: tuning goes through mxdet_debug_set_tuning, never getenv inside the library."""
    import subprocess
    from mxdetection_amd import _lib
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out
    lib = _lib.load()
    assert lib.mxdet_debug_set_tuning(99, 1) == -1 and b"unknown key" in lib.mxdet_last_error()
    assert lib.mxdet_debug_set_tuning(0, 123) == 0 and lib.mxdet_debug_set_tuning(0, -1) == 0


def test_comm_entries_validate_arguments():
    """No GPU and no second rank needed: argument checks happen before RCCL is touched."""
    import ctypes as C
    from mxdetection_amd import _lib
    lib = _lib.load()
    assert lib.mxdet_comm_unique_id(None) == -1
    out = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert lib.mxdet_comm_create(ident, 2, 5, C.byref(out)) == -1 and b"rank 5 of 2" in lib.mxdet_last_error()
    assert lib.mxdet_allreduce_bucket(None, None, 4, None, None) == -1
    assert lib.mxdet_comm_wait(None, -1, None) == -1
    assert lib.mxdet_comm_destroy(None) == 0


def test_version_and_error_string():
    from mxdetection_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.mxdet_version()
    assert lib.mxdet_debug_set_tuning(0, -1) == 0       # any successful entry clears this thread's error string
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
