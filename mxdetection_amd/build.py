"""Build libmxdet_hip.so (gfx950) in-tree with hipcc, and the C oracle with gcc.

Run as ``python -m mxdetection_amd.build``. hipcc cross-compiles for gfx950 without a GPU.
Objects are cached by source mtime so an incremental rebuild only recompiles what changed.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
OBJ = os.path.join(ROOT, "_obj")
LIB = os.path.join(ROOT, "libmxdet_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
          "-Wno-unused-result", "-I", os.path.join(REPO, "include")]
# Detection/box math must not be FMA-contracted: it is compared bit-exact with the C oracle.
EXACT = ["-ffp-contract=off"]
# MFMA conv / elementwise kernels: default contraction is fine (tolerance-checked).
PER_FILE = {
    "conv.hip": [],
    "mask.hip": ["-ffp-contract=off"],
    "wgrad.hip": [],
    "dense_misc.hip": [],
}


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(REPO, "include", f) for f in os.listdir(os.path.join(REPO, "include"))]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force=False):
    name = os.path.basename(src)
    obj = os.path.join(OBJ, name + ".o")
    newest = max(os.path.getmtime(src), _deps(), os.path.getmtime(__file__))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, None
    flags = COMMON + PER_FILE.get(name, EXACT)
    cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return obj, "hipcc failed for %s:\n%s\n%s" % (name, r.stdout[-4000:], r.stderr[-8000:])
    return obj, None


def build_hip(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    errs = [e for _, e in res if e]
    if errs:
        raise RuntimeError("\n".join(errs))
    objs = [o for o, _ in res]
    if (not os.path.exists(LIB)) or force or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
    if verbose:
        print("built", LIB)
    return LIB


def build_oracle(force=False, verbose=True):
    odir = os.path.join(REPO, "oracle")
    r = subprocess.run(["make", "-C", odir] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-4000:] + r.stderr[-8000:])
    if verbose:
        print("built oracle")


if __name__ == "__main__":
    force = "--force" in sys.argv
    build_hip(force)
    if os.path.isdir(os.path.join(REPO, "oracle")):
        build_oracle(force)
