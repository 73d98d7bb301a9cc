"""ctypes binding of librwkv6_amd.so (the C ABI declared in include/wkv6_amd.h).

There is deliberately no CPU fallback: if the HIP library cannot be built or loaded, importing the
operator fails loudly.
"""
import ctypes
import os
import threading

from . import _build

_VP = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_SZ = ctypes.c_size_t
_U = ctypes.c_uint

class SeqSet(ctypes.Structure):
    """wkv6_seq_set of include/wkv6_amd.h: one of the two problems of a pair launch."""
    _fields_ = [("r", _VP), ("k", _VP), ("v", _VP), ("w", _VP), ("y", _VP), ("gy", _VP), ("gr", _VP), ("gk", _VP), ("gv", _VP),
                ("gw", _VP), ("gu", _VP), ("ckpt", _VP), ("ckpt_bytes", _SZ), ("rev_n", _VP), ("rev_mask", _U)]


# name -> (restype, argtypes); must list every symbol of include/wkv6_amd.h
SIGNATURES = {
    "wkv6_cuda_forward": (_I, [_I] * 4 + [_VP] * 7),
    "wkv6_cuda_backward": (_I, [_I] * 4 + [_VP] * 12),
    "wkv6bi_cuda_forward": (_I, [_I] * 4 + [_VP] * 8),
    "wkv6bi_cuda_backward": (_I, [_I] * 4 + [_VP] * 13),
    "wkv6state_cuda_forward": (_I, [_I] * 4 + [_VP] * 8),
    "wkv6state_cuda_backward": (_I, [_I] * 4 + [_VP] * 14),
    "wkv6infctx_cuda_forward": (_I, [_I] * 4 + [_VP] * 8),
    "wkv6infctx_cuda_backward": (_I, [_I] * 4 + [_VP] * 14),
    "rwkv6_cuda_forward_bf16": (_I, [_I] * 4 + [_VP] * 8),
    "rwkv6_cuda_forward_fp16": (_I, [_I] * 4 + [_VP] * 8),
    "rwkv6_cuda_forward_fp32": (_I, [_I] * 4 + [_VP] * 8),
    "wkv6_backward_workspace_bytes": (_SZ, [_I] * 4),
    "wkv6bi_workspace_bytes": (_SZ, [_I] * 4),
    "wkv6bi_kept_bytes": (_SZ, [_I] * 4),
    "wkv6_forward_ex": (_I, [_I] * 4 + [_VP] * 8 + [_U, _VP]),
    "wkv6_forward_ckpt_ex": (_I, [_I] * 4 + [_VP] * 9 + [_SZ, _U, _VP]),
    "wkv6_forward_gn_ex": (_I, [_I] * 4 + [_VP] * 9 + [_SZ] + [_VP] * 3 + [_F] + [_VP] * 2 + [_U, _VP]),
    "wkv6_backward_ex": (_I, [_I] * 4 + [_VP] * 14 + [_SZ, _U, _VP]),
    "wkv6_forward_rev_ex": (_I, [_I] * 4 + [_VP] * 7 + [_SZ, _VP, _U, _U, _VP]),
    "wkv6_backward_rev_ex": (_I, [_I] * 4 + [_VP] * 12 + [_SZ, _VP, _U, _U, _VP]),
    "wkv6_forward_pair_ex": (_I, [_I] * 4 + [_VP, ctypes.POINTER(SeqSet), _U, _VP]),
    "wkv6_backward_pair_ex": (_I, [_I] * 4 + [_VP, ctypes.POINTER(SeqSet), _U, _VP]),
    "wkv6bi_forward_ex": (_I, [_I] * 4 + [_VP] * 9 + [_SZ, _U, _VP]),
    "wkv6bi_backward_ex": (_I, [_I] * 4 + [_VP] * 14 + [_SZ, _U, _VP]),
    "wkv6_ddlerp_forward": (_I, [_I] * 4 + [_VP] * 6),
    "wkv6_ddlerp_backward": (_I, [_I] * 4 + [_VP] * 8 + [_I, _VP]),
    "wkv6_ddlerp_rev_forward": (_I, [_I] * 4 + [_VP] * 7),
    "wkv6_ddlerp_rev_backward": (_I, [_I] * 4 + [_VP] * 9 + [_I, _VP]),
    "wkv6_gn_gate_forward": (_I, [_L, _I, _I] + [_VP] * 4 + [_F] + [_VP] * 3),
    "wkv6_gn_gate_backward": (_I, [_L, _I, _I] + [_VP] * 10 + [_I, _VP]),
    "wkv6_sqrelu_forward": (_I, [_L] + [_VP] * 3),
    "wkv6_sqrelu_backward": (_I, [_L] + [_VP] * 4),
    "wkv6_sigmul_forward": (_I, [_L] + [_VP] * 4),
    "wkv6_sigmul_backward": (_I, [_L] + [_VP] * 6),
    "wkv6_selftest": (_I, [_VP]),
    "wkv6_set_clock_ring": (None, [_VP, _I, _I]),
    "wkv6_set_clock_buffer": (None, [_VP, _I]),
    "wkv6_clock_ring_counts": (None, [ctypes.POINTER(_L), ctypes.POINTER(_L)]),
    "wkv6_pass_marker": (_I, [_VP]),
    "wkv6_amd_version": (ctypes.c_char_p, []),
}

# flags of include/wkv6_amd.h
W_EW_F32, W_RAW, IO_F32, S0_PER_BATCH, ALGO_SCAN, CKPT_VALID, BI_KEEP_CKPT, PARTIALS_F32 = 0, 1, 2, 4, 16, 32, 64, 128
REV_R, REV_K, REV_V, REV_W, REV_Y, REV_ALL = 1, 2, 4, 8, 16, 31      # wkv6_*_rev_ex: tensors held in reversed order

EUNSUPPORTED = -4
ERRORS = {-1: "WKV6_EINVAL (shape: need C == H*64 and B,T,C,H >= 1)", -2: "WKV6_ENULL (null pointer)",
          -3: "WKV6_EWORKSPACE (workspace too small / allocation failed)", -4: "WKV6_EUNSUPPORTED"}

# measurement aids: an explicit A/B library of an earlier round (RWKV_AMD_LIB) may lack them; has_symbol() says which are there
MEASUREMENT_AIDS = ("wkv6_set_clock_ring", "wkv6_set_clock_buffer", "wkv6_clock_ring_counts", "wkv6_pass_marker")

_lib = None
_missing = set()
_lock = threading.Lock()


def has_symbol(name):
    load()
    return name not in _missing


def lib_path():
    return _build.LIB_PATH


def load():
    """Load (building first if the sources are newer) and type the library."""
    global _lib
    with _lock:
        if _lib is None:
            path = os.environ.get("RWKV_AMD_LIB")        # explicit library (A/B experiments); normally unset
            if not path:
                path = _build.LIB_PATH
                if os.environ.get("RWKV_AMD_NO_BUILD", "0") != "1":
                    path = _build.build()
            if not os.path.exists(path):
                raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
            lib = ctypes.CDLL(path)
            explicit = bool(os.environ.get("RWKV_AMD_LIB"))
            for name, (res, args) in SIGNATURES.items():
                if explicit and name in MEASUREMENT_AIDS and not hasattr(lib, name):
                    _missing.add(name)           # an A/B library of an earlier round: the measurement aids are newer than it
                    continue
                fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


_selftested = set()


def selftest_once(device_index, stream_ptr):
    """Run wkv6_selftest the first time a device is used (cross-lane primitives, then the chunked MFMA kernels against
    the exact scan kernels): a mis-scheduled build fails loudly here instead of training silently wrong.
    RWKV_AMD_NO_SELFTEST=1 skips it."""
    if device_index in _selftested:
        return
    with _lock:
        if device_index in _selftested:
            return
        if os.environ.get("RWKV_AMD_NO_SELFTEST", "0") != "1":
            rc = _lib.wkv6_selftest(stream_ptr)
            if rc != 0:
                raise RuntimeError(f"librwkv6_amd.so failed its device self-test on cuda:{device_index} (code {rc}); "
                                   "the build is unusable on this device")
        _selftested.add(device_index)


def check(rc, what):
    if rc != 0:
        msg = ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"code {rc}")
        raise RuntimeError(f"{what} failed: {msg}")
