"""Checkpoint container: MXNet 1.3.0 NDArray-list layout restated in
mxdetection_amd/utils/params_io.py. MXNet is not available offline, so the reader is pinned by bytes assembled by hand
from the documented layout (and by round trips); compatibility with MXNet-written files is otherwise unpinned."""
import struct

import numpy as np
import pytest

from mxdetection_amd.utils import fold_batchnorm, load_params, save_params
from mxdetection_amd.utils.params_io import ParamsFormatError


def _hand_file():
    w = np.arange(6, dtype=np.float32).reshape(2, 3)
    m = np.array([1, -2, 3], dtype=np.int32)
    b = struct.pack("<QQQ", 0x112, 0, 2)
    b += struct.pack("<IiI", 0xF993FAC9, 0, 2) + struct.pack("<qq", 2, 3) + struct.pack("<iii", 1, 0, 0) + w.tobytes()
    b += struct.pack("<II", 0xF993FAC8, 1) + struct.pack("<q", 3) + struct.pack("<iii", 2, 3, 4) + m.tobytes()   # V1, gpu(3) context
    b += struct.pack("<Q", 2)
    for name in (b"arg:fc_weight", b"aux:bn_moving_mean"):
        b += struct.pack("<Q", len(name)) + name
    return b, w, m


def test_reads_hand_assembled_v2_and_v1_records():
    b, w, m = _hand_file()
    p = load_params(b)
    assert list(p) == ["arg:fc_weight", "aux:bn_moving_mean"]
    assert p["arg:fc_weight"].dtype == np.float32 and np.array_equal(p["arg:fc_weight"], w)
    assert p["aux:bn_moving_mean"].dtype == np.int32 and np.array_equal(p["aux:bn_moving_mean"], m)


def test_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    params = {"arg:conv0_weight": rng.standard_normal((4, 3, 7, 7)).astype(np.float32),
              "aux:bn0_moving_var": rng.random(4).astype(np.float64), "arg:idx": np.arange(5, dtype=np.int64),
              "arg:half": rng.standard_normal((2, 2)).astype(np.float16)}
    fn = tmp_path / "model-0001.params"
    save_params(str(fn), params)
    raw = fn.read_bytes()
    assert struct.unpack_from("<QQQ", raw, 0) == (0x112, 0, 4)
    assert struct.unpack_from("<IiI", raw, 24) == (0xF993FAC9, 0, 4)            # first record: V2 magic, dense, rank 4
    assert struct.unpack_from("<4q", raw, 36) == (4, 3, 7, 7)
    got = load_params(str(fn))
    assert list(got) == list(params)
    for k in params:
        assert got[k].dtype == params[k].dtype and np.array_equal(got[k], params[k])
    assert save_params(None, params) == raw


def test_unnamed_list_and_errors():
    a = np.ones((2,), np.float32)
    raw = save_params(None, {"0": a})
    unnamed = raw[:len(raw) - (8 + 8 + 1)] + struct.pack("<Q", 0)        # same arrays, zero names
    assert list(load_params(unnamed)) == ["0"]
    with pytest.raises(ParamsFormatError):
        load_params(b"\x00" * 24)
    with pytest.raises(ParamsFormatError):
        load_params(raw[:40])
    with pytest.raises(ParamsFormatError):
        save_params(None, {"c": np.zeros(2, np.complex64)})


def test_fold_batchnorm_matches_definition():
    rng = np.random.default_rng(1)
    w = rng.standard_normal((5, 3, 3, 4)).astype(np.float32)
    gamma, beta = rng.random(5) + 0.5, rng.standard_normal(5)
    mean, var = rng.standard_normal(5), rng.random(5) + 0.1
    wf, bf = fold_batchnorm(w, gamma, beta, mean, var, eps=2e-5)
    x = rng.standard_normal((3, 3, 4)).astype(np.float32)
    conv = np.tensordot(w.astype(np.float64), x, axes=3)
    want = gamma * (conv - mean) / np.sqrt(var + 2e-5) + beta
    got = np.tensordot(wf.astype(np.float64), x, axes=3) + bf
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    wf1, _ = fold_batchnorm(w, gamma, beta, mean, var, fix_gamma=True)
    assert np.allclose(wf1, w / np.sqrt(var + 2e-5).reshape(-1, 1, 1, 1), rtol=1e-6)
