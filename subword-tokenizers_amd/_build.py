"""Build libswt_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch, no cmake: one `hipcc --offload-arch=gfx950 -shared` over csrc/*.hip.  hipcc cross-compiles without a
GPU, so this also runs in the build container; the .so is git-ignored but travels with the snapshot.
"""
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libswt_hip.so")
SOURCES = ["swt_core.hip", "swt_tile.hip", "swt_dedup.hip", "swt_bpe_encode.hip", "swt_wp.hip", "swt_words.hip", "swt_bpe_train.hip", "swt_dist.hip", "swt_lower.hip", "swt_metrics.hip"]
HEADERS = ["swt_common.h", "swt_tile.h", "swt_dedup.h", "swt_words.h", "swt_train.h", "unicode_classes.inc", "unicode_lower.inc", os.path.join("..", "..", "include", "swt.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# diagnostic builds (tools/ only): SWT_EXTRA_FLAGS="-DSWT_STAMPS" or "-DSWT_ABLATION"; the flags are part of the build stamp
FLAGS += os.environ.get("SWT_EXTRA_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libswt_hip.so cannot be built (set HIPCC or install ROCm)")


def _stamp():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libswt_hip.so.  Returns its path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp_path = os.path.join(LIB_DIR, "build.stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp_path):
        with open(stamp_path) as f:
            if f.read().strip() == stamp:
                return LIB_PATH
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp_path, "w") as f:
        f.write(stamp)
    return LIB_PATH


PYHOST_SRC = os.path.join(CSRC, "swt_pyhost.c")
PYHOST_PATH = os.path.join(LIB_DIR, "_swt_pyhost.so")


def build_pyhost(force=False):
    """Compile csrc/swt_pyhost.c (list[str] -> joined UTF-8, CPython C API; loaded with ctypes.PyDLL) with the C compiler.
    Returns its path, or None where no compiler / Python.h is to be had: the callers then join and encode in Python."""
    import sysconfig
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp_path = os.path.join(LIB_DIR, "pyhost.stamp")
    with open(PYHOST_SRC, "rb") as fh:
        stamp = hashlib.sha256(fh.read() + sysconfig.get_python_version().encode()).hexdigest()
    if not force and os.path.exists(PYHOST_PATH) and os.path.exists(stamp_path):
        with open(stamp_path) as f:
            if f.read().strip() == stamp:
                return PYHOST_PATH
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths().get("include")
    if not cc or not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    r = subprocess.run([cc, "-O2", "-Wall", "-shared", "-fPIC", "-pthread", "-I", inc, PYHOST_SRC, "-o", PYHOST_PATH],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        return None
    with open(stamp_path, "w") as f:
        f.write(stamp)
    return PYHOST_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_pyhost(force=True))
