"""Ahead-of-time build of librwkv6_amd.so with hipcc for gfx950 (no JIT, no torch headers).

The built library stays in-tree (rwkv_lm_ext_amd/librwkv6_amd.so) so that it travels with the
source snapshot; it is git-ignored.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librwkv6_amd.so")
SOURCES = ["wkv6_scan.hip", "wkv6_chunk.hip", "wkv6_chunk_bwd12k.hip", "wkv6_mix.hip", "wkv6_api.hip"]
HEADERS = ["wkv6_common.h", "wkv6_scan.h", "wkv6_chunk.h", os.path.join("..", "..", "include", "wkv6_amd.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link librwkv6_amd.so.  Returns its path.

    Safe under torchrun: an inter-process lock serialises the staleness check and the build, objects go to a private
    temporary directory and the finished library is moved into place atomically, so another rank can never dlopen a
    half-written file."""
    import fcntl
    import tempfile
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            with tempfile.TemporaryDirectory(prefix="rwkv6_build_", dir=PKG_DIR) as tmp:
                objs = []
                for src in SOURCES:
                    obj = os.path.join(tmp, src.replace(".hip", ".o"))
                    cmd = [_hipcc()] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
                    if verbose:
                        print(" ".join(cmd))
                    subprocess.check_call(cmd)
                    objs.append(obj)
                out = os.path.join(tmp, "librwkv6_amd.so")
                cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(out, LIB_PATH)
            return LIB_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
