// C++ torch-extension shim over the C ABI of librwkv6_amd.so (include/wkv6_amd.h): the pybind module and the
// TORCH_LIBRARY registrations that the reference builds from cuda/wkv6_op.cpp:8-22, cuda/wkv6_bi_op.cpp:8-22,
// cuda/wkv6state_op.cpp:8-22, cuda/wkv6infctx_op.cpp:8-22 and cuda/rwkv6_op.cpp:12-34 -- same operator names and
// positional signatures (caller-allocated outputs, in-place writes, void return), one extension instead of five.
// What the reference shims do not do and this one does: device guard, launch on the current stream of the tensors'
// device, dtype / contiguity / shape checks, error codes turned into exceptions.
//
// Build (INTEGRATION.md level 2): torch.utils.cpp_extension.load(name=..., sources=[this file],
//     extra_include_paths=[<repo>/include], extra_ldflags=["-L<repo>/rwkv_lm_ext_amd", "-lrwkv6_amd", "-Wl,-rpath,..."]).
// -DWKV6_SHIM_PREFIX=foo registers the libraries as foo_wkv6, foo_wkv6bi, ... (lets a process that has already imported
// rwkv_lm_ext_amd.wkv6_op, which defines torch.ops.wkv6*, load the shim as well).
#include <torch/extension.h>
// ROCm builds of PyTorch keep the "cuda" device type for HIP tensors; the matching guard and stream live in these headers
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <limits>

#include "wkv6_amd.h"

namespace {

using torch::Tensor;

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;

void need(const Tensor& t, const char* name, at::ScalarType dt, const Tensor& like)
{
    TORCH_CHECK(t.is_cuda(), name, " must be on the GPU (the WKV6 operator has no CPU path)");
    TORCH_CHECK(t.device() == like.device(), name, " is on ", t.device(), ", expected ", like.device());
    TORCH_CHECK(t.scalar_type() == dt, name, " must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
void need_btc(const Tensor& t, const char* name, int64_t B, int64_t T, int64_t C, at::ScalarType dt, const Tensor& like)
{
    need(t, name, dt, like);
    TORCH_CHECK(t.dim() == 3 && t.size(0) == B && t.size(1) == T && t.size(2) == C, name, " must be [B,T,C]");
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with code ", rc, " (include/wkv6_amd.h)"); }
// the C ABI takes B, T, C, H as int (the reference's own prototypes, cuda/wkv6_op.cpp:5-6): refuse sizes that would be narrowed
void need_sizes(int64_t B, int64_t T, int64_t C, int64_t H)
{
    constexpr int64_t lim = std::numeric_limits<int>::max();
    TORCH_CHECK(B >= 1 && T >= 1 && C >= 1 && H >= 1 && B <= lim && T <= lim && C <= lim && H <= lim, "B, T, C, H must be positive ints");
}

constexpr auto BF = at::kBFloat16;
constexpr auto F32 = at::kFloat;

// ---- wkv6 (cuda/wkv6_op.cpp:8-13): w is the fp32 tensor ew = -exp(w_raw)
void wkv6_forward(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& r, Tensor& k, Tensor& v, Tensor& w, Tensor& u, Tensor& y)
{
    need_sizes(B, T, C, H);
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", BF, r); need_btc(y, "y", B, T, C, BF, r);
    const DeviceGuard guard(r.device());
    ok(wkv6_cuda_forward(B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(), u.data_ptr(), y.data_ptr(),
                         stream_of(r)), "wkv6 forward");
}
void wkv6_backward(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& r, Tensor& k, Tensor& v, Tensor& w, Tensor& u, Tensor& gy,
                   Tensor& gr, Tensor& gk, Tensor& gv, Tensor& gw, Tensor& gu)
{
    need_sizes(B, T, C, H);
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", BF, r); need_btc(gy, "gy", B, T, C, BF, r);
    need_btc(gr, "gr", B, T, C, BF, r); need_btc(gk, "gk", B, T, C, BF, r); need_btc(gv, "gv", B, T, C, BF, r);
    need_btc(gw, "gw", B, T, C, BF, r); need(gu, "gu", BF, r);
    TORCH_CHECK(gu.numel() == B * C, "gu must be [B,C]");
    const DeviceGuard guard(r.device());
    ok(wkv6_cuda_backward(B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(), u.data_ptr(), gy.data_ptr(),
                          gr.data_ptr(), gk.data_ptr(), gv.data_ptr(), gw.data_ptr(), gu.data_ptr(), stream_of(r)), "wkv6 backward");
}

// ---- wkv6_bi (cuda/wkv6_bi_op.cpp:8-13): mask int32 [B,T] after H
void wkv6bi_forward(int64_t B, int64_t T, int64_t C, int64_t H, const Tensor& mask, Tensor& r, Tensor& k, Tensor& v, Tensor& w,
                    Tensor& u, Tensor& y)
{
    need_sizes(B, T, C, H);
    need(mask, "mask", at::kInt, r);
    TORCH_CHECK(mask.numel() == B * T, "mask must be [B,T]");
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", BF, r); need_btc(y, "y", B, T, C, BF, r);
    const DeviceGuard guard(r.device());
    ok(wkv6bi_cuda_forward(B, T, C, H, mask.data_ptr<int>(), r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(),
                           u.data_ptr(), y.data_ptr(), stream_of(r)), "wkv6_bi forward");
}
void wkv6bi_backward(int64_t B, int64_t T, int64_t C, int64_t H, const Tensor& mask, Tensor& r, Tensor& k, Tensor& v, Tensor& w,
                     Tensor& u, Tensor& gy, Tensor& gr, Tensor& gk, Tensor& gv, Tensor& gw, Tensor& gu)
{
    need_sizes(B, T, C, H);
    need(mask, "mask", at::kInt, r);
    TORCH_CHECK(mask.numel() == B * T, "mask must be [B,T]");
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", BF, r); need_btc(gy, "gy", B, T, C, BF, r);
    need_btc(gr, "gr", B, T, C, BF, r); need_btc(gk, "gk", B, T, C, BF, r); need_btc(gv, "gv", B, T, C, BF, r);
    need_btc(gw, "gw", B, T, C, BF, r); need(gu, "gu", BF, r);
    TORCH_CHECK(gu.numel() == B * C, "gu must be [B,C]");
    const DeviceGuard guard(r.device());
    ok(wkv6bi_cuda_backward(B, T, C, H, mask.data_ptr<int>(), r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(),
                            u.data_ptr(), gy.data_ptr(), gr.data_ptr(), gk.data_ptr(), gv.data_ptr(), gw.data_ptr(), gu.data_ptr(),
                            stream_of(r)), "wkv6_bi backward");
}

// ---- wkv6state / wkv6infctx (cuda/wkv6state_op.cpp:8-13, cuda/wkv6infctx_op.cpp:8-13): w is the raw bf16 decay
template <bool INFCTX>
void state_forward(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& r, Tensor& k, Tensor& v, Tensor& w, Tensor& u, Tensor& s,
                   Tensor& y)
{
    need_sizes(B, T, C, H);
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, BF, r); need(u, "u", BF, r); need(s, "s", BF, r); need_btc(y, "y", B, T, C, BF, r);
    TORCH_CHECK(s.numel() == (INFCTX ? B : 1) * H * 64 * 64, "s must be ", INFCTX ? "[B,H,N,N]" : "[H,N,N]");
    const DeviceGuard guard(r.device());
    if (INFCTX)
        ok(wkv6infctx_cuda_forward(B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr(), u.data_ptr(), s.data_ptr(),
                                   y.data_ptr(), stream_of(r)), "wkv6infctx forward");
    else
        ok(wkv6state_cuda_forward(B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr(), u.data_ptr(), s.data_ptr(),
                                  y.data_ptr(), stream_of(r)), "wkv6state forward");
}
template <bool INFCTX>
void state_backward(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& r, Tensor& k, Tensor& v, Tensor& w, Tensor& u, Tensor& s,
                    Tensor& gy, Tensor& gr, Tensor& gk, Tensor& gv, Tensor& gw, Tensor& gu, Tensor& gs)
{
    need_sizes(B, T, C, H);
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, BF, r); need(u, "u", BF, r); need(s, "s", BF, r); need_btc(gy, "gy", B, T, C, BF, r);
    need_btc(gr, "gr", B, T, C, BF, r); need_btc(gk, "gk", B, T, C, BF, r); need_btc(gv, "gv", B, T, C, BF, r);
    need_btc(gw, "gw", B, T, C, BF, r); need(gu, "gu", BF, r); need(gs, "gs", BF, r);
    TORCH_CHECK(gs.numel() == B * H * 64 * 64, "gs must be [B,H,N,N]");
    const DeviceGuard guard(r.device());
    auto fn = INFCTX ? wkv6infctx_cuda_backward : wkv6state_cuda_backward;
    ok(fn(B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr(), u.data_ptr(), s.data_ptr(), gy.data_ptr(),
          gr.data_ptr(), gk.data_ptr(), gv.data_ptr(), gw.data_ptr(), gu.data_ptr(), gs.data_ptr(), stream_of(r)),
       INFCTX ? "wkv6infctx backward" : "wkv6state backward");
}

// ---- rwkv6 (cuda/rwkv6_op.cpp:12-23): stateful forward-only kernel, w is the fp32 decay exp(-exp(w_raw))
void rwkv6_forward_bf16(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& state, Tensor& r, Tensor& k, Tensor& v, Tensor& w,
                        Tensor& u, Tensor& y)
{
    need_sizes(B, T, C, H);
    need(state, "state", F32, r);
    need_btc(r, "r", B, T, C, BF, r); need_btc(k, "k", B, T, C, BF, r); need_btc(v, "v", B, T, C, BF, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", BF, r); need_btc(y, "y", B, T, C, BF, r);
    TORCH_CHECK(state.numel() == B * H * 64 * 64, "state must be [B,H,N,N] ([H,N,N] for B = 1)");
    const DeviceGuard guard(r.device());
    ok(rwkv6_cuda_forward_bf16(B, T, C, H, state.data_ptr<float>(), r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(),
                               u.data_ptr(), y.data_ptr(), stream_of(r)), "rwkv6 forward_bf16");
}
void rwkv6_forward_fp32(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& state, Tensor& r, Tensor& k, Tensor& v, Tensor& w,
                        Tensor& u, Tensor& y)
{
    need_sizes(B, T, C, H);
    need(state, "state", F32, r);
    need_btc(r, "r", B, T, C, F32, r); need_btc(k, "k", B, T, C, F32, r); need_btc(v, "v", B, T, C, F32, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", F32, r); need_btc(y, "y", B, T, C, F32, r);
    TORCH_CHECK(state.numel() == B * H * 64 * 64, "state must be [B,H,N,N] ([H,N,N] for B = 1)");
    const DeviceGuard guard(r.device());
    ok(rwkv6_cuda_forward_fp32(B, T, C, H, state.data_ptr<float>(), r.data_ptr<float>(), k.data_ptr<float>(), v.data_ptr<float>(),
                               w.data_ptr<float>(), u.data_ptr<float>(), y.data_ptr<float>(), stream_of(r)), "rwkv6 forward_fp32");
}

// cuda/rwkv6_op.cpp:16-19: r, k, v, u, y in fp16; the kernel widens the inputs to fp32 (exact), computes and carries the state in
// fp32 and rounds y to fp16 once (cuda/rwkv6.cu:8-71)
void rwkv6_forward_fp16(int64_t B, int64_t T, int64_t C, int64_t H, Tensor& state, Tensor& r, Tensor& k, Tensor& v, Tensor& w,
                        Tensor& u, Tensor& y)
{
    constexpr auto F16 = at::kHalf;
    need_sizes(B, T, C, H);
    need(state, "state", F32, r);
    need_btc(r, "r", B, T, C, F16, r); need_btc(k, "k", B, T, C, F16, r); need_btc(v, "v", B, T, C, F16, r);
    need_btc(w, "w", B, T, C, F32, r); need(u, "u", F16, r); need_btc(y, "y", B, T, C, F16, r);
    TORCH_CHECK(state.numel() == B * H * 64 * 64, "state must be [B,H,N,N] ([H,N,N] for B = 1)");
    const DeviceGuard guard(r.device());
    ok(rwkv6_cuda_forward_fp16(B, T, C, H, state.data_ptr<float>(), r.data_ptr(), k.data_ptr(), v.data_ptr(), w.data_ptr<float>(),
                               u.data_ptr(), y.data_ptr(), stream_of(r)), "rwkv6 forward_fp16");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    // one python module with one sub-namespace per reference extension
    auto wkv6 = m.def_submodule("wkv6");
    wkv6.def("forward", &wkv6_forward, "wkv6 forward");
    wkv6.def("backward", &wkv6_backward, "wkv6 backward");
    auto bi = m.def_submodule("wkv6_bi");
    bi.def("forward", &wkv6bi_forward, "wkv6_bi forward");
    bi.def("backward", &wkv6bi_backward, "wkv6_bi backward");
    auto st = m.def_submodule("wkv6state");
    st.def("forward", &state_forward<false>, "wkv6state forward");
    st.def("backward", &state_backward<false>, "wkv6state backward");
    auto ic = m.def_submodule("wkv6infctx");
    ic.def("forward", &state_forward<true>, "wkv6infctx forward");
    ic.def("backward", &state_backward<true>, "wkv6infctx backward");
    auto rw = m.def_submodule("rwkv6");
    rw.def("forward_bf16", &rwkv6_forward_bf16, "rwkv6 forward_bf16");
    rw.def("forward_fp16", &rwkv6_forward_fp16, "rwkv6 forward_fp16");
    rw.def("forward_fp32", &rwkv6_forward_fp32, "rwkv6 forward_fp32");
}

#ifndef WKV6_SHIM_PREFIX
#define WKV6_SHIM_LIB(name) name
#else
#define WKV6_SHIM_CAT2(a, b) a##_##b
#define WKV6_SHIM_CAT(a, b) WKV6_SHIM_CAT2(a, b)
#define WKV6_SHIM_LIB(name) WKV6_SHIM_CAT(WKV6_SHIM_PREFIX, name)
#endif

// TORCH_LIBRARY pastes its first argument: expand the (possibly prefixed) name first
#define WKV6_SHIM_TORCH_LIBRARY2(ns, m) TORCH_LIBRARY(ns, m)
#define WKV6_SHIM_TORCH_LIBRARY(ns, m) WKV6_SHIM_TORCH_LIBRARY2(ns, m)

WKV6_SHIM_TORCH_LIBRARY(WKV6_SHIM_LIB(wkv6), m) { m.def("forward", wkv6_forward); m.def("backward", wkv6_backward); }
WKV6_SHIM_TORCH_LIBRARY(WKV6_SHIM_LIB(wkv6bi), m) { m.def("forward", wkv6bi_forward); m.def("backward", wkv6bi_backward); }
WKV6_SHIM_TORCH_LIBRARY(WKV6_SHIM_LIB(wkv6state), m) { m.def("forward", state_forward<false>); m.def("backward", state_backward<false>); }
WKV6_SHIM_TORCH_LIBRARY(WKV6_SHIM_LIB(wkv6infctx), m) { m.def("forward", state_forward<true>); m.def("backward", state_backward<true>); }
WKV6_SHIM_TORCH_LIBRARY(WKV6_SHIM_LIB(rwkv6), m) { m.def("forward_bf16", rwkv6_forward_bf16); m.def("forward_fp16", rwkv6_forward_fp16); m.def("forward_fp32", rwkv6_forward_fp32); }
