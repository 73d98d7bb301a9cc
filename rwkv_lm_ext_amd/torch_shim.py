"""Builds and loads the C++ torch-extension shim (csrc/torch_shim/wkv6_torch_shim.cpp) -- INTEGRATION.md level 2: the
reference's `torch.utils.cpp_extension.load(name="wkv6", sources=[...op.cpp, ...cuda.cu])` (src/model.py:188-189) with
the CUDA source replaced by a link against librwkv6_amd.so."""
import os

from . import _build

_SHIM_SRC = os.path.join(_build.CSRC, "torch_shim", "wkv6_torch_shim.cpp")
_INCLUDE = os.path.join(os.path.dirname(_build.PKG_DIR), "include")
BUILD_DIR = os.path.join(_build.PKG_DIR, "torch_shim_build")      # in-tree, so that the built module travels with the repo


def load(prefix="shim", verbose=False, build_directory=BUILD_DIR):
    """Returns the extension module: `.wkv6.forward(B,T,C,H,r,k,v,ew,u,y)`, `.wkv6.backward(...)`, `.wkv6_bi`, `.wkv6state`,
    `.wkv6infctx`, `.rwkv6`; also registers torch.ops.<prefix>_wkv6 etc. (prefix=None: the reference's bare names, only in
    a process that has not imported rwkv_lm_ext_amd.wkv6_op, which defines torch.ops.wkv6* itself)."""
    from torch.utils.cpp_extension import ROCM_HOME, load as _load
    _build.build()
    rocm_inc = os.path.join(ROCM_HOME or "/opt/rocm", "include")
    flags = ["-O2"] + ([f"-DWKV6_SHIM_PREFIX={prefix}"] if prefix else [])
    if build_directory:
        os.makedirs(build_directory, exist_ok=True)
    return _load(name=f"wkv6_torch_shim_{prefix or 'bare'}", sources=[_SHIM_SRC], extra_include_paths=[_INCLUDE, rocm_inc],
                 extra_cflags=flags,
                 extra_ldflags=[f"-L{_build.PKG_DIR}", "-lrwkv6_amd", f"-Wl,-rpath,{_build.PKG_DIR}"],
                 build_directory=build_directory, verbose=verbose)
