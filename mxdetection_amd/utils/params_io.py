"""MXNet `.params` container reader / writer and frozen-BatchNorm folding.

Slot: mxdetection/utils (/root/reference/README.md:25); the reference delegates checkpoints to MXNet 1.3.0
(`mx.nd.save` / `mx.nd.load`, README.md:37). The file layout restated here is MXNet 1.3.0's NDArray-list format
(src/ndarray/ndarray.cc, `NDArray::Save` / `NDArray::Load` list form) -- MXNet is not installed and cannot be fetched, so
compatibility with files written by MXNet itself is **unpinned** (checked only against bytes assembled by hand from that
layout in tests/test_params_io.py):

    uint64  0x112                      list magic (kMXAPINDArrayListMagic)
    uint64  0                          reserved
    uint64  n                          number of arrays, then n x NDArray:
        uint32  0xF993FAC9             NDARRAY_V2_MAGIC  (V1 = 0xF993FAC8 has no storage-type field; files older than
                                       V1 start directly with ndim -- all three are read, V2 is written)
        int32   stype                  0 = dense (the only one supported here)
        uint32  ndim, int64 dims[ndim]
        int32   dev_type (1 = cpu), int32 dev_id
        int32   type_flag              0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
        raw little-endian data
    uint64  n_names, then n_names x (uint64 length, bytes)      keys such as "arg:conv0_weight", "aux:bn0_moving_mean"

Pure numpy, no code is executed from the file. bf16 tensors are stored as float32 (MXNet 1.3.0 has no bf16 type).
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V2_MAGIC = 0xF993FAC9
V1_MAGIC = 0xF993FAC8
_TYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _TYPES.items()}


class ParamsFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b, self.o = memoryview(buf), 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.o + n > len(self.b):
            raise ParamsFormatError("truncated file at byte %d" % self.o)
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.o + n > len(self.b):
            raise ParamsFormatError("truncated tensor data at byte %d" % self.o)
        v = self.b[self.o:self.o + n]
        self.o += n
        return v


def _read_array(r):
    first = r.take("<I")
    if first == V2_MAGIC:
        stype = r.take("<i")
        if stype != 0:
            raise ParamsFormatError("sparse storage type %d is not supported" % stype)
        ndim = r.take("<I")
    elif first == V1_MAGIC:
        ndim = r.take("<I")
    else:                       # pre-V1 files: the first word is ndim and dims are uint32
        ndim = first
        if ndim > 32:
            raise ParamsFormatError("bad NDArray magic / rank 0x%x" % first)
        dims = [r.take("<I") for _ in range(ndim)]
        return _finish_array(r, dims)
    if ndim > 32:
        raise ParamsFormatError("implausible rank %d" % ndim)
    dims = [r.take("<q") for _ in range(ndim)]
    return _finish_array(r, dims)


def _finish_array(r, dims):
    if len(dims) == 0:
        return None             # MXNet writes nothing more for an empty NDArray
    if any(d < 0 for d in dims):
        raise ParamsFormatError("negative dimension in %r" % (dims,))
    r.take("<ii")               # context: device type, device id (ignored: everything is loaded to host memory)
    flag = r.take("<i")
    if flag not in _TYPES:
        raise ParamsFormatError("unknown type flag %d" % flag)
    dt = np.dtype(_TYPES[flag]).newbyteorder("<")
    n = int(np.prod(dims, dtype=np.int64))
    data = np.frombuffer(r.raw(n * dt.itemsize), dtype=dt).reshape(dims)
    return np.array(data, dtype=_TYPES[flag])      # own, writable, native-endian copy


def load_params(path_or_bytes):
    """Read an MXNet NDArray-list file. Returns an ordered dict name -> numpy array (names "0", "1", ... if the file
    carries none, as `mx.nd.save(fname, [a, b])` writes)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        buf = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            buf = f.read()
    r = _Reader(buf)
    magic, _reserved = r.take("<QQ")
    if magic != LIST_MAGIC:
        raise ParamsFormatError("not an MXNet NDArray list (magic 0x%x)" % magic)
    n = r.take("<Q")
    if n > (1 << 24):
        raise ParamsFormatError("implausible array count %d" % n)
    arrays = [_read_array(r) for _ in range(n)]
    nn = r.take("<Q")
    if nn not in (0, n):
        raise ParamsFormatError("%d names for %d arrays" % (nn, n))
    names = []
    for _ in range(nn):
        ln = r.take("<Q")
        names.append(bytes(r.raw(ln)).decode("utf-8"))
    if not names:
        names = [str(i) for i in range(n)]
    return {k: v for k, v in zip(names, arrays)}


def save_params(path, params):
    """Write name -> array (numpy, or anything `np.asarray` accepts; torch tensors via `.numpy()`) as an MXNet NDArray list."""
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(params))]
    for name, a in params.items():
        a = np.ascontiguousarray(a.detach().cpu().float().numpy() if hasattr(a, "detach") else a)
        if a.dtype not in _FLAGS:
            raise ParamsFormatError("%s: dtype %s has no MXNet 1.3.0 type flag" % (name, a.dtype))
        if a.ndim == 0:
            a = a.reshape(1)
        out.append(struct.pack("<IiI", V2_MAGIC, 0, a.ndim))
        out.append(struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))
        out.append(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    out.append(struct.pack("<Q", len(params)))
    for name in params:
        b = name.encode("utf-8")
        out.append(struct.pack("<Q", len(b)) + b)
    data = b"".join(out)
    if path is None:
        return data
    with open(path, "wb") as f:
        f.write(data)
    return None


def fold_batchnorm(weight, gamma, beta, moving_mean, moving_var, eps=2e-5, fix_gamma=False):
    """Fold a frozen BatchNorm (use_global_stats) that follows a bias-free convolution into the filter and a bias:
    y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta  ==  conv(x; w * s) + (beta - mean * s),  s = gamma / sqrt(var + eps).
    `weight` is [Cout, ...] (any layout with the output channel first). eps 2e-5 is the MXNet ResNet symbols' value;
    fix_gamma=True treats gamma as 1 (MXNet's BatchNorm default)."""
    w = np.asarray(weight, np.float64)
    g = np.ones_like(np.asarray(moving_var, np.float64)) if fix_gamma else np.asarray(gamma, np.float64)
    s = g / np.sqrt(np.asarray(moving_var, np.float64) + eps)
    wf = w * s.reshape((-1,) + (1,) * (w.ndim - 1))
    bf = np.asarray(beta, np.float64) - np.asarray(moving_mean, np.float64) * s
    return wf.astype(np.float32), bf.astype(np.float32)
