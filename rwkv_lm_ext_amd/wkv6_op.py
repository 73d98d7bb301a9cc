"""Operator layer: the torch-extension surface of cuda/wkv6_op.cpp, cuda/wkv6_bi_op.cpp,
cuda/wkv6state_op.cpp and cuda/wkv6infctx_op.cpp, served by librwkv6_amd.so.

The reference obtains four module objects from ``torch.utils.cpp_extension.load`` (src/model.py:80-81,
134-135, 188-189; cuda/wkv6_bi.py:7) and calls ``module.forward(...)`` / ``module.backward(...)`` on
them with caller-allocated outputs.  The four objects below keep those names, positional signatures,
in-place output convention and dtypes:

    wkv6_cuda.forward (B,T,C,H, r,k,v, ew[f32], u, y)                              cuda/wkv6_op.cpp:8-10
    wkv6_cuda.backward(B,T,C,H, r,k,v, ew[f32], u, gy, gr,gk,gv,gw, gu[B,C])       cuda/wkv6_op.cpp:11-13
    wkv6_bi_cuda.forward / backward: same with `mask` (int32 [B,T]) after H        cuda/wkv6_bi_op.cpp:8-13
    wkv6state_cuda / wkv6infctx_cuda.forward(B,T,C,H, r,k,v, w[bf16 raw], u, s, y) cuda/wkv6state_op.cpp:8-10
                                   .backward(..., s, gy, gr,gk,gv,gw, gu, gs)      cuda/wkv6state_op.cpp:11-13

They are also registered as ``torch.ops.wkv6.forward`` etc. (the reference's TORCH_LIBRARY blocks,
cuda/wkv6_op.cpp:19-22).  Unlike the reference shims (which check nothing) every call validates
device, dtype, contiguity and shape and launches on the current stream of the tensors' device.
"""
import torch

from . import _lib

HEAD_SIZE = 64


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _check_tensors(B, T, C, H, named, dtype=torch.bfloat16):
    dev = None
    for name, (t, shape, dt) in named.items():
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a tensor")
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be on the GPU (the WKV6 operator has no CPU path)")
        dev = dev or t.device
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
        if t.dtype != (dt or dtype):
            raise RuntimeError(f"{name} must be {dt or dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    if C != H * HEAD_SIZE:
        raise RuntimeError(f"C ({C}) must equal H*{HEAD_SIZE} ({H * HEAD_SIZE})")   # reference: assert(H*_N_ == C)
    _lib.load()
    if dev.index not in _lib._selftested:
        with torch.cuda.device(dev):
            _lib.selftest_once(dev.index, _stream_ptr())
    return dev


class _Wkv6:
    """Stand-in for the module object of `load(name="wkv6", ...)` (src/model.py:188-189)."""

    @staticmethod
    def forward(B, T, C, H, r, k, v, w, u, y):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(r=(r, btc, None), k=(k, btc, None), v=(v, btc, None),
                                              w=(w, btc, torch.float32), u=(u, None, None), y=(y, btc, None)))
        with torch.cuda.device(dev):
            rc = _lib.load().wkv6_cuda_forward(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(y),
                                               _stream_ptr())
        _lib.check(rc, "wkv6 forward")

    @staticmethod
    def backward(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(
            r=(r, btc, None), k=(k, btc, None), v=(v, btc, None), w=(w, btc, torch.float32), u=(u, None, None),
            gy=(gy, btc, None), gr=(gr, btc, None), gk=(gk, btc, None), gv=(gv, btc, None), gw=(gw, btc, None),
            gu=(gu, (B, C), None)))
        ws = torch.empty(_lib.load().wkv6_backward_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().wkv6_backward_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), None,
                                              _ptr(gy), _ptr(gr), _ptr(gk), _ptr(gv), _ptr(gw), _ptr(gu), None,
                                              _ptr(ws), ws.numel(), _lib.W_EW_F32, _stream_ptr())
        _lib.check(rc, "wkv6 backward")


class _Wkv6Bi:
    """Stand-in for `load(name="wkv6_bi", ...)` (cuda/wkv6_bi.py:7)."""

    @staticmethod
    def forward(B, T, C, H, mask, r, k, v, w, u, y):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(mask=(mask, (B, T), torch.int32), r=(r, btc, None), k=(k, btc, None),
                                              v=(v, btc, None), w=(w, btc, torch.float32), u=(u, None, None),
                                              y=(y, btc, None)))
        ws = torch.empty(_lib.load().wkv6bi_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().wkv6bi_forward_ex(B, T, C, H, _ptr(mask), None, _ptr(r), _ptr(k), _ptr(v), _ptr(w),
                                               _ptr(u), _ptr(y), _ptr(ws), ws.numel(), _lib.W_EW_F32, _stream_ptr())
        _lib.check(rc, "wkv6_bi forward")

    @staticmethod
    def backward(B, T, C, H, mask, r, k, v, w, u, gy, gr, gk, gv, gw, gu):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(
            mask=(mask, (B, T), torch.int32), r=(r, btc, None), k=(k, btc, None), v=(v, btc, None),
            w=(w, btc, torch.float32), u=(u, None, None), gy=(gy, btc, None), gr=(gr, btc, None),
            gk=(gk, btc, None), gv=(gv, btc, None), gw=(gw, btc, None), gu=(gu, (B, C), None)))
        ws = torch.empty(_lib.load().wkv6bi_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().wkv6bi_backward_ex(B, T, C, H, _ptr(mask), None, _ptr(r), _ptr(k), _ptr(v), _ptr(w),
                                                _ptr(u), _ptr(gy), _ptr(gr), _ptr(gk), _ptr(gv), _ptr(gw),
                                                _ptr(gu), _ptr(ws), ws.numel(), _lib.W_EW_F32, _stream_ptr())
        _lib.check(rc, "wkv6_bi backward")


class _Wkv6State:
    """Stand-in for `load(name="wkv6state", ...)` (src/model.py:134-135); s: bf16 [H,N,N]."""
    _per_batch = False
    _name = "wkv6state"

    @classmethod
    def _s_shape(cls, B, H):
        return (B, H, HEAD_SIZE, HEAD_SIZE) if cls._per_batch else (H, HEAD_SIZE, HEAD_SIZE)

    @classmethod
    def forward(cls, B, T, C, H, r, k, v, w, u, s, y):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(r=(r, btc, None), k=(k, btc, None), v=(v, btc, None), w=(w, btc, None),
                                              u=(u, None, None), s=(s, cls._s_shape(B, H), None), y=(y, btc, None)))
        fn = getattr(_lib.load(), cls._name + "_cuda_forward")
        with torch.cuda.device(dev):
            rc = fn(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(s), _ptr(y), _stream_ptr())
        _lib.check(rc, cls._name + " forward")

    @classmethod
    def backward(cls, B, T, C, H, r, k, v, w, u, s, gy, gr, gk, gv, gw, gu, gs):
        btc = (B, T, C)
        dev = _check_tensors(B, T, C, H, dict(
            r=(r, btc, None), k=(k, btc, None), v=(v, btc, None), w=(w, btc, None), u=(u, None, None),
            s=(s, cls._s_shape(B, H), None), gy=(gy, btc, None), gr=(gr, btc, None), gk=(gk, btc, None),
            gv=(gv, btc, None), gw=(gw, btc, None), gu=(gu, (B, C), None),
            gs=(gs, (B, H, HEAD_SIZE, HEAD_SIZE), None)))
        flags = _lib.W_RAW | (_lib.S0_PER_BATCH if cls._per_batch else 0)
        ws = torch.empty(_lib.load().wkv6_backward_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().wkv6_backward_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(s),
                                              _ptr(gy), _ptr(gr), _ptr(gk), _ptr(gv), _ptr(gw), _ptr(gu), _ptr(gs),
                                              _ptr(ws), ws.numel(), flags, _stream_ptr())
        _lib.check(rc, cls._name + " backward")


class _Wkv6Infctx(_Wkv6State):
    """Stand-in for `load(name="wkv6infctx", ...)` (src/model.py:80-81); s: bf16 [B,H,N,N], the forward
    overwrites it with the final state (cuda/wkv6infctx_cuda.cu:65-67)."""
    _per_batch = True
    _name = "wkv6infctx"


class _Rwkv6:
    """Stand-in for `load(name="rwkv6", ...)` (src/model_run.py:46-47): stateful forward-only kernel for inference.
    forward_bf16 / forward_fp32 (B,T,C,H, state[f32], r,k,v, eew[f32 decay], u, y); state updated in place."""

    @staticmethod
    def _call(B, T, C, H, state, r, k, v, w, u, y, io):
        btc = (B, T, C)
        sshape = None if state.numel() == B * H * HEAD_SIZE * HEAD_SIZE else (-1,)
        dev = _check_tensors(B, T, C, H, dict(state=(state, sshape, torch.float32), r=(r, btc, io), k=(k, btc, io),
                                              v=(v, btc, io), w=(w, btc, torch.float32), u=(u, None, io), y=(y, btc, io)),
                             dtype=io)
        lib = _lib.load()
        fn = {torch.bfloat16: lib.rwkv6_cuda_forward_bf16, torch.float16: lib.rwkv6_cuda_forward_fp16,
              torch.float32: lib.rwkv6_cuda_forward_fp32}[io]
        with torch.cuda.device(dev):
            rc = fn(B, T, C, H, _ptr(state), _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(y), _stream_ptr())
        _lib.check(rc, "rwkv6 forward")

    @staticmethod
    def forward_bf16(B, T, C, H, state, r, k, v, w, u, y):
        _Rwkv6._call(B, T, C, H, state, r, k, v, w, u, y, torch.bfloat16)

    @staticmethod
    def forward_fp32(B, T, C, H, state, r, k, v, w, u, y):
        _Rwkv6._call(B, T, C, H, state, r, k, v, w, u, y, torch.float32)

    @staticmethod
    def forward_fp16(B, T, C, H, state, r, k, v, w, u, y):
        """cuda/rwkv6_op.cpp:16-19: r, k, v, u, y in fp16; the kernel widens the inputs to fp32 (exact), computes and carries the
        state in fp32 and rounds y to fp16 once (cuda/rwkv6.cu:8-71)."""
        _Rwkv6._call(B, T, C, H, state, r, k, v, w, u, y, torch.float16)


rwkv6 = _Rwkv6
wkv6_cuda = _Wkv6
wkv6_bi_cuda = _Wkv6Bi
wkv6state_cuda = _Wkv6State
wkv6infctx_cuda = _Wkv6Infctx


# ---- generic entry used by the autograd layer (raw bf16 decay, optional fp32 I/O, explicit state) ----
def new_checkpoint(B, T, C, H, device):
    """Buffer for the forward-state checkpoints that `forward_ex(..., ckpt=)` fills and `backward_ex(..., ckpt=)`
    consumes (also usable as the backward workspace)."""
    return torch.empty(_lib.load().wkv6_backward_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=device)


def forward_ex(r, k, v, w, u, H, s0=None, s_out=None, w_is_ew=False, y=None, algo=None, ckpt=None):
    """y = WKV6(r,k,v,w,u[,s0]) with the I/O type of `r` (bf16, or fp32 for numerics tests).
    algo: None (library default: chunked MFMA kernel for bf16 I/O) or "scan" (exact token-serial kernels)."""
    B, T, C = r.shape
    io = r.dtype
    if io not in (torch.bfloat16, torch.float32):
        raise RuntimeError(f"unsupported I/O dtype {io}")
    btc = (B, T, C)
    wdt = torch.float32 if w_is_ew else io
    named = dict(r=(r, btc, io), k=(k, btc, io), v=(v, btc, io), w=(w, btc, wdt), u=(u, (H, HEAD_SIZE), io))
    flags = (_lib.W_EW_F32 if w_is_ew else _lib.W_RAW) | (_lib.IO_F32 if io == torch.float32 else 0)
    flags |= _lib.ALGO_SCAN if algo == "scan" else 0
    if s0 is not None:
        per_batch = s0.dim() == 4
        named["s0"] = (s0, (B, H, HEAD_SIZE, HEAD_SIZE) if per_batch else (H, HEAD_SIZE, HEAD_SIZE), io)
        flags |= _lib.S0_PER_BATCH if per_batch else 0
    if s_out is not None:
        named["s_out"] = (s_out, (B, H, HEAD_SIZE, HEAD_SIZE), io)
    if y is None:
        y = torch.empty(btc, device=r.device, dtype=io)
    named["y"] = (y, btc, io)
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    with torch.cuda.device(dev):
        if ckpt is not None:       # training forward: also store the per-group state checkpoints for the backward
            rc = _lib.load().wkv6_forward_ckpt_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(s0),
                                                  _ptr(s_out), _ptr(y), _ptr(ckpt), ckpt.numel(), flags, _stream_ptr())
        else:
            rc = _lib.load().wkv6_forward_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(s0),
                                             _ptr(s_out), _ptr(y), flags, _stream_ptr())
    _lib.check(rc, "wkv6 forward_ex")
    return y


def forward_gn_ex(r, k, v, w, u, H, gate, gamma, beta, eps, ckpt=None, want_y=True, want_stats=True):
    """The operator with the time-mix block's GroupNorm(H) * gate epilogue fused into its store (src/model.py:462-468, SURVEY.md
    row n1).  Returns (out, y or None, stats or None), or None where the library cannot fuse (so few (batch, head) pairs that
    two workgroups share one): the caller then runs the two kernels."""
    B, T, C = r.shape
    bf = torch.bfloat16
    btc = (B, T, C)
    named = dict(r=(r, btc, bf), k=(k, btc, bf), v=(v, btc, bf), w=(w, btc, bf), u=(u, (H, HEAD_SIZE), bf),
                 gate=(gate, btc, bf), gamma=(gamma, (C,), bf), beta=(beta, (C,), bf))
    dev = _check_tensors(B, T, C, H, named, dtype=bf)
    out = torch.empty(btc, device=dev, dtype=bf)
    y = torch.empty(btc, device=dev, dtype=bf) if want_y else None
    stats = torch.empty((B * T, H, 2), device=dev, dtype=torch.float32) if want_stats else None
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_forward_gn_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), None, None, _ptr(y),
                                            _ptr(ckpt), 0 if ckpt is None else ckpt.numel(), _ptr(gate), _ptr(gamma), _ptr(beta),
                                            float(eps), _ptr(out), _ptr(stats), _lib.W_RAW, _stream_ptr())
    if rc == _lib.EUNSUPPORTED:
        return None
    _lib.check(rc, "wkv6 forward_gn_ex")
    return out, y, stats


def backward_ex(r, k, v, w, u, gy, H, s0=None, w_is_ew=False, want_gs=False, algo=None, ckpt=None):
    """Returns (gr, gk, gv, gw, gu[B,C], gs[B,H,N,N] or None): gradients in the I/O type of `r`, the per-batch partials gu / gs
    in fp32 (WKV6_PARTIALS_F32)."""
    B, T, C = r.shape
    io = r.dtype
    btc = (B, T, C)
    wdt = torch.float32 if w_is_ew else io
    named = dict(r=(r, btc, io), k=(k, btc, io), v=(v, btc, io), w=(w, btc, wdt), u=(u, (H, HEAD_SIZE), io),
                 gy=(gy, btc, io))
    flags = (_lib.W_EW_F32 if w_is_ew else _lib.W_RAW) | (_lib.IO_F32 if io == torch.float32 else 0)
    flags |= _lib.ALGO_SCAN if algo == "scan" else 0
    if s0 is not None:
        per_batch = s0.dim() == 4
        named["s0"] = (s0, (B, H, HEAD_SIZE, HEAD_SIZE) if per_batch else (H, HEAD_SIZE, HEAD_SIZE), io)
        flags |= _lib.S0_PER_BATCH if per_batch else 0
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    gr, gk, gv, gw = (torch.empty(btc, device=dev, dtype=io) for _ in range(4))
    # gu / gs are per-batch partial sums the caller reduces: always fp32, so that the parameter gradient is rounded once
    flags |= _lib.PARTIALS_F32
    gu = torch.empty((B, C), device=dev, dtype=torch.float32)
    gs = torch.empty((B, H, HEAD_SIZE, HEAD_SIZE), device=dev, dtype=torch.float32) if want_gs else None
    if ckpt is not None:           # checkpoints written by forward_ex(..., ckpt=ckpt) on the same inputs
        ws = ckpt
        flags |= _lib.CKPT_VALID
    else:
        ws = torch.empty(_lib.load().wkv6_backward_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_backward_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(s0),
                                          _ptr(gy), _ptr(gr), _ptr(gk), _ptr(gv), _ptr(gw), _ptr(gu), _ptr(gs),
                                          _ptr(ws), ws.numel(), flags, _stream_ptr())
    _lib.check(rc, "wkv6 backward_ex")
    return gr, gk, gv, gw, gu, gs


REV_R, REV_K, REV_V, REV_W, REV_Y, REV_ALL = _lib.REV_R, _lib.REV_K, _lib.REV_V, _lib.REV_W, _lib.REV_Y, _lib.REV_ALL


def _check_rev(B, rev_n, rev_mask, dev):
    if not (isinstance(rev_n, torch.Tensor) and rev_n.dtype == torch.int32 and rev_n.is_contiguous() and
            tuple(rev_n.shape) == (B,) and rev_n.device == dev):
        raise RuntimeError("rev_n must be a contiguous int32 [B] tensor on the device of r")
    if rev_mask & ~REV_ALL:
        raise RuntimeError(f"rev_mask {rev_mask} has unknown bits")


def forward_rev_ex(r, k, v, w, u, H, rev_n, rev_mask, y=None, ckpt=None, algo=None):
    """WKV6 over partially reversed sequences without materialising the reversal (include/wkv6_amd.h, wkv6_forward_rev_ex):
    tokens [0, rev_n[b]) of the tensors named in `rev_mask` (REV_R | REV_K | REV_V | REV_W | REV_Y) are taken in reverse
    order, the rest in place.  I/O type of `r`: bf16 (chunked MFMA kernels; algo="scan" forces the exact ones) or fp32 (exact
    scan kernels)."""
    B, T, C = r.shape
    btc, io = (B, T, C), r.dtype
    if io not in (torch.bfloat16, torch.float32):
        raise RuntimeError(f"unsupported I/O dtype {io}")
    if y is None:
        y = torch.empty(btc, device=r.device, dtype=io)
    named = dict(r=(r, btc, io), k=(k, btc, io), v=(v, btc, io), w=(w, btc, io), u=(u, (H, HEAD_SIZE), io), y=(y, btc, io))
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    _check_rev(B, rev_n, rev_mask, dev)
    flags = _lib.W_RAW | (_lib.IO_F32 if io == torch.float32 else 0) | (_lib.ALGO_SCAN if algo == "scan" else 0)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_forward_rev_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(y), _ptr(ckpt),
                                             0 if ckpt is None else ckpt.numel(), _ptr(rev_n), rev_mask, flags,
                                             _stream_ptr())
    _lib.check(rc, "wkv6 forward_rev_ex")
    return y


def _pair_sets(B, T, C, H, sets, u, bwd):
    """Check and pack the two problems of a pair launch.  sets: two dicts with r, k, v, w (+ y | gy), ckpt, rev_n, rev_mask."""
    bf = torch.bfloat16
    btc = (B, T, C)
    arr = (_lib.SeqSet * 2)()
    keep = []                                   # tensors the packed pointers refer to
    dev = None
    for i, q in enumerate(sets):
        named = {n: (q[n], btc, bf) for n in ("r", "k", "v", "w") + (("gy",) if bwd else ("y",))}
        named["u"] = (u, (H, HEAD_SIZE), bf)
        dev = _check_tensors(B, T, C, H, named, dtype=bf)
        rev_n, rev_mask = q.get("rev_n"), q.get("rev_mask", 0)
        if rev_n is not None:
            _check_rev(B, rev_n, rev_mask, dev)
        ckpt = q.get("ckpt")
        if bwd and ckpt is None:
            raise RuntimeError("wkv6 pair backward: both problems need the checkpoints their forward wrote")
        e = arr[i]
        e.r, e.k, e.v, e.w = _ptr(q["r"]), _ptr(q["k"]), _ptr(q["v"]), _ptr(q["w"])
        if bwd:
            outs = [torch.empty(btc, device=dev, dtype=bf) for _ in range(4)] + [torch.empty((B, C), device=dev, dtype=torch.float32)]
            e.gy, e.gr, e.gk, e.gv, e.gw, e.gu = (_ptr(t) for t in [q["gy"]] + outs)
            keep.append(outs)
        else:
            e.y = _ptr(q["y"])
        e.ckpt, e.ckpt_bytes = _ptr(ckpt), 0 if ckpt is None else ckpt.numel()
        e.rev_n, e.rev_mask = _ptr(rev_n), rev_mask if rev_n is not None else 0
    return arr, keep, dev


def forward_pair_ex(H, u, sets):
    """Both operator calls of a bidirectional time-mix layer in one launch (include/wkv6_amd.h, wkv6_forward_pair_ex; SURVEY.md
    row n2): `sets` = two dicts {r, k, v, w, [y], [ckpt], [rev_n, rev_mask]} of one shape.  Returns (y0, y1)."""
    B, T, C = sets[0]["r"].shape
    for q in sets:
        if q.get("y") is None:
            q["y"] = torch.empty((B, T, C), device=q["r"].device, dtype=torch.bfloat16)
    arr, _, dev = _pair_sets(B, T, C, H, sets, u, bwd=False)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_forward_pair_ex(B, T, C, H, _ptr(u), arr, _lib.W_RAW, _stream_ptr())
    _lib.check(rc, "wkv6 forward_pair_ex")
    return sets[0]["y"], sets[1]["y"]


def backward_pair_ex(H, u, sets):
    """Gradients of forward_pair_ex: two tuples (gr, gk, gv, gw, gu[B,C] fp32), one per problem; `sets` as in the forward plus gy."""
    B, T, C = sets[0]["r"].shape
    arr, keep, dev = _pair_sets(B, T, C, H, sets, u, bwd=True)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_backward_pair_ex(B, T, C, H, _ptr(u), arr, _lib.W_RAW | _lib.PARTIALS_F32, _stream_ptr())
    _lib.check(rc, "wkv6 backward_pair_ex")
    return tuple(keep[0]), tuple(keep[1])


def backward_rev_ex(r, k, v, w, u, gy, H, rev_n, rev_mask, ckpt=None, algo=None):
    """Gradients of forward_rev_ex: (gr, gk, gv, gw, gu[B,C]); each gradient is laid out like its tensor."""
    B, T, C = r.shape
    btc, io = (B, T, C), r.dtype
    if io not in (torch.bfloat16, torch.float32):
        raise RuntimeError(f"unsupported I/O dtype {io}")
    named = dict(r=(r, btc, io), k=(k, btc, io), v=(v, btc, io), w=(w, btc, io), u=(u, (H, HEAD_SIZE), io), gy=(gy, btc, io))
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    _check_rev(B, rev_n, rev_mask, dev)
    gr, gk, gv, gw = (torch.empty(btc, device=dev, dtype=io) for _ in range(4))
    gu = torch.empty((B, C), device=dev, dtype=torch.float32)
    exact = io == torch.float32 or algo == "scan"
    flags = _lib.W_RAW | _lib.PARTIALS_F32 | (_lib.IO_F32 if io == torch.float32 else 0) | (_lib.ALGO_SCAN if algo == "scan" else 0)
    if ckpt is not None and not exact:
        ws = ckpt
        flags |= _lib.CKPT_VALID
    else:
        ws = torch.empty(_lib.load().wkv6_backward_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6_backward_rev_ex(B, T, C, H, _ptr(r), _ptr(k), _ptr(v), _ptr(w), _ptr(u), _ptr(gy), _ptr(gr),
                                              _ptr(gk), _ptr(gv), _ptr(gw), _ptr(gu), _ptr(ws), ws.numel(), _ptr(rev_n),
                                              rev_mask, flags, _stream_ptr())
    _lib.check(rc, "wkv6 backward_rev_ex")
    return gr, gk, gv, gw, gu


def bi_new_workspace(B, T, C, H, device):
    """Workspace of the wkv6_bi pair: the forward's fp32 y side buffer, the state checkpoints of both scans (kept from
    forward to backward when `ws` is passed to both calls) and the backward's four fp32 gradient side buffers."""
    return torch.empty(_lib.load().wkv6bi_workspace_bytes(B, T, C, H), dtype=torch.uint8, device=device)


def bi_new_kept(B, T, C, H, device):
    """The part of the wkv6_bi workspace that has to live from the forward to the backward (row lengths, order, the state
    checkpoints of both scans: 2 x 4 B per token-channel at 64-token checkpoint spacing); passed as `ws`, the fp32 side buffers
    (4 B per token-channel in the forward, 16 B in the backward) become per-call scratch of the library."""
    return torch.empty(_lib.load().wkv6bi_kept_bytes(B, T, C, H), dtype=torch.uint8, device=device)


def _mask_or_lens(mask, lens, B, T):
    """wkv6bi_*_ex take the reference's int32 mask [B,T] (row length = 1 + index of its first zero) or, instead, the row lengths
    themselves as int32 [B] (0 .. T): exactly one of the two."""
    if (mask is None) == (lens is None):
        raise RuntimeError("pass either mask or lens")
    return {"mask": (mask, (B, T), torch.int32)} if lens is None else {"lens": (lens, (B,), torch.int32)}


def bi_forward_ex(mask, r, k, v, w, u, H, w_is_ew=False, algo=None, ws=None, lens=None):
    """ws: a bi_new_workspace() buffer the caller keeps for bi_backward_ex(..., ws=ws): the forward then stores the state
    checkpoints of both scans in it and the backward skips its two state passes.  lens: row lengths instead of the mask."""
    B, T, C = r.shape
    io = r.dtype
    btc = (B, T, C)
    wdt = torch.float32 if w_is_ew else io
    named = dict(**_mask_or_lens(mask, lens, B, T), r=(r, btc, io), k=(k, btc, io), v=(v, btc, io),
                 w=(w, btc, wdt), u=(u, (H, HEAD_SIZE), io))
    flags = (_lib.W_EW_F32 if w_is_ew else _lib.W_RAW) | (_lib.IO_F32 if io == torch.float32 else 0)
    flags |= _lib.ALGO_SCAN if algo == "scan" else 0
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    y = torch.empty(btc, device=dev, dtype=io)
    if ws is not None:
        flags |= _lib.BI_KEEP_CKPT
    else:
        ws = bi_new_workspace(B, T, C, H, dev)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6bi_forward_ex(B, T, C, H, _ptr(mask), _ptr(lens), _ptr(r), _ptr(k), _ptr(v), _ptr(w),
                                           _ptr(u), _ptr(y), _ptr(ws), ws.numel(), flags, _stream_ptr())
    _lib.check(rc, "wkv6_bi forward_ex")
    return y


def bi_backward_ex(mask, r, k, v, w, u, gy, H, w_is_ew=False, algo=None, ws=None, lens=None):
    """ws: the workspace a preceding bi_forward_ex(..., ws=ws) on the same inputs filled (checkpoints valid)."""
    B, T, C = r.shape
    io = r.dtype
    btc = (B, T, C)
    wdt = torch.float32 if w_is_ew else io
    named = dict(**_mask_or_lens(mask, lens, B, T), r=(r, btc, io), k=(k, btc, io), v=(v, btc, io),
                 w=(w, btc, wdt), u=(u, (H, HEAD_SIZE), io), gy=(gy, btc, io))
    flags = (_lib.W_EW_F32 if w_is_ew else _lib.W_RAW) | (_lib.IO_F32 if io == torch.float32 else 0)
    flags |= _lib.ALGO_SCAN if algo == "scan" else 0
    dev = _check_tensors(B, T, C, H, named, dtype=io)
    gr, gk, gv, gw = (torch.empty(btc, device=dev, dtype=io) for _ in range(4))
    gu = torch.empty((B, C), device=dev, dtype=torch.float32)      # per-batch partials: fp32 (WKV6_PARTIALS_F32)
    flags |= _lib.PARTIALS_F32
    if ws is not None and io == torch.bfloat16 and algo != "scan":
        flags |= _lib.CKPT_VALID
    elif ws is None:
        ws = bi_new_workspace(B, T, C, H, dev)
    with torch.cuda.device(dev):
        rc = _lib.load().wkv6bi_backward_ex(B, T, C, H, _ptr(mask), _ptr(lens), _ptr(r), _ptr(k), _ptr(v), _ptr(w),
                                            _ptr(u), _ptr(gy), _ptr(gr), _ptr(gk), _ptr(gv), _ptr(gw), _ptr(gu),
                                            _ptr(ws), ws.numel(), flags, _stream_ptr())
    _lib.check(rc, "wkv6_bi backward_ex")
    return gr, gk, gv, gw, gu


def selftest():
    """Cross-lane primitive self-test on the current device (0 = pass)."""
    return _lib.load().wkv6_selftest(_stream_ptr())


def pass_marker():
    """Launch the empty kernel wkv6::pass_marker_kernel on the current stream (a phase boundary in a profiler's dispatch list).
    A no-op with an explicit A/B library (RWKV_AMD_LIB) that predates the symbol."""
    if _lib.has_symbol("wkv6_pass_marker"):
        _lib.check(_lib.load().wkv6_pass_marker(_stream_ptr()), "wkv6_pass_marker")


class ClockProbe:
    """In-run shader clock and duration of the chunked kernels' launches (wkv6_set_clock_ring, include/wkv6_amd.h): while active, wave 0
    of the first `n_slots` workgroups of every chunked forward / backward launch stamps {s_memtime, s_memrealtime} at its start and end
    into a ring of the last `n_launches` launches of each kind.

        with ClockProbe(dev, n_slots=64, n_launches=4096) as probe:
            ... launches ...
            rec = probe.read()

    read() -> {"fwd_ghz", "bwd_ghz"} (median over the workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz for the LAST launch of
    each kind, None where nothing was stamped) plus, per kind, the lists "fwd_ghz_launches" / "fwd_us_launches" (oldest first: every
    launch still in the ring; us = max(end) - min(start) of s_memrealtime over the stamped workgroups) and "fwd_count".
    The probe owns the device buffer for as long as the library holds its address: close() (also on __exit__ / __del__) switches the
    stamps off first.  With an explicit A/B library (RWKV_AMD_LIB) that predates the symbols everything is a no-op and read() returns
    None values."""

    def __init__(self, device, n_slots=256, n_launches=1):
        self.n, self.nl = int(n_slots), int(n_launches)
        self.lib = _lib.load()
        self.ring = _lib.has_symbol("wkv6_set_clock_ring")
        self.active = self.ring or _lib.has_symbol("wkv6_set_clock_buffer")
        self.buf = None
        if not self.active:
            return
        if not self.ring:
            self.nl = 1
        self.buf = torch.zeros(2 * self.nl * self.n * 4, dtype=torch.int64, device=device)
        if self.ring:
            self.lib.wkv6_set_clock_ring(self.buf.data_ptr(), self.n, self.nl)
        else:
            self.lib.wkv6_set_clock_buffer(self.buf.data_ptr(), self.n)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        if not (self.active and self.ring):
            return None, None
        import ctypes
        f, b = ctypes.c_long(0), ctypes.c_long(0)
        self.lib.wkv6_clock_ring_counts(ctypes.byref(f), ctypes.byref(b))
        return f.value, b.value

    def read(self):
        out = {}
        names = ("fwd", "bwd")
        if not self.active or self.buf is None:
            for nm in names:
                out[nm + "_ghz"] = None
                out[nm + "_ghz_launches"], out[nm + "_us_launches"], out[nm + "_count"] = [], [], None
            return out
        d = self.buf.view(2, self.nl, self.n, 4).cpu().double()
        cnt = self.counts()
        for i, nm in enumerate(names):
            n = cnt[i]
            if n is None:                       # plain buffer (no ring in the library): one launch, the last
                order = [0]
            else:
                order = [j % self.nl for j in range(max(0, n - self.nl), n)]
            ghz, us = [], []
            for j in order:
                s = d[i, j]
                dc, dr = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
                ok = (dr > 0) & (s[:, 0] > 0)
                if bool(ok.any()):
                    ghz.append(round(float((dc[ok] / dr[ok]).median()) * 0.1, 3))
                    us.append(round(float(s[ok, 3].max() - s[ok, 1].min()) * 0.01, 2))
                else:
                    ghz.append(None)
                    us.append(None)
            out[nm + "_ghz_launches"], out[nm + "_us_launches"], out[nm + "_count"] = ghz, us, n
            out[nm + "_ghz"] = ghz[-1] if ghz else None
        return out

    def close(self):
        if getattr(self, "buf", None) is None:
            return
        if self.ring:
            self.lib.wkv6_set_clock_ring(None, 0, 0)
        else:
            self.lib.wkv6_set_clock_buffer(None, 0)
        self.buf = None


# ---- torch.ops registration: the TORCH_LIBRARY(wkv6|wkv6bi|wkv6state|wkv6infctx, m) blocks ------------
def _register():
    T9 = "Tensor r, Tensor k, Tensor v, Tensor w, Tensor u"
    defs = {
        "wkv6": (f"(int B, int T, int C, int H, {T9}, Tensor(a!) y) -> ()",
                 f"(int B, int T, int C, int H, {T9}, Tensor gy, Tensor(a!) gr, Tensor(b!) gk, Tensor(c!) gv, "
                 f"Tensor(d!) gw, Tensor(e!) gu) -> ()", _Wkv6),
        "wkv6bi": (f"(int B, int T, int C, int H, Tensor mask, {T9}, Tensor(a!) y) -> ()",
                   f"(int B, int T, int C, int H, Tensor mask, {T9}, Tensor gy, Tensor(a!) gr, Tensor(b!) gk, "
                   f"Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu) -> ()", _Wkv6Bi),
        "wkv6state": (f"(int B, int T, int C, int H, {T9}, Tensor s, Tensor(a!) y) -> ()",
                      f"(int B, int T, int C, int H, {T9}, Tensor s, Tensor gy, Tensor(a!) gr, Tensor(b!) gk, "
                      f"Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu, Tensor(f!) gs) -> ()", _Wkv6State),
        "wkv6infctx": (f"(int B, int T, int C, int H, {T9}, Tensor(z!) s, Tensor(a!) y) -> ()",
                       f"(int B, int T, int C, int H, {T9}, Tensor s, Tensor gy, Tensor(a!) gr, Tensor(b!) gk, "
                       f"Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu, Tensor(f!) gs) -> ()", _Wkv6Infctx),
    }
    libs = []
    rw = torch.library.Library("rwkv6", "DEF")          # TORCH_LIBRARY(rwkv6, m), cuda/rwkv6_op.cpp:30-34
    for name in ("forward_bf16", "forward_fp16", "forward_fp32"):
        rw.define(name + "(int B, int T, int C, int H, Tensor(s!) state, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, "
                         "Tensor(a!) y) -> ()")
        rw.impl(name, getattr(_Rwkv6, name), "CUDA")
    libs.append(rw)
    for ns, (fwd_schema, bwd_schema, impl) in defs.items():
        lib = torch.library.Library(ns, "DEF")
        lib.define("forward" + fwd_schema)
        lib.define("backward" + bwd_schema)
        lib.impl("forward", impl.forward, "CUDA")
        lib.impl("backward", impl.backward, "CUDA")
        libs.append(lib)
    return libs


_LIBRARIES = _register()
