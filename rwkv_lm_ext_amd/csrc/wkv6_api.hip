// extern "C" entry points of librwkv6_amd.so (declared in include/wkv6_amd.h).
#include "../../include/wkv6_amd.h"
#include "wkv6_scan.h"

#include <mutex>
#include <vector>
#include <cstring>
#include <cmath>

using namespace wkv6;

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

int check_shape(int B, int T, int C, int H)
{
    if (B < 1 || T < 1 || C < 1 || H < 1) return WKV6_EINVAL;
    if ((long)H * HEAD != (long)C) return WKV6_EINVAL;      // reference: assert(H*_N_ == C)
    if (((long)T + 64) * C >= (1L << 31)) return WKV6_EUNSUPPORTED; // one sequence (rounded up to whole groups) must stay 32-bit
                                                                  // addressable: per-lane byte offsets of bf16 tensors are 32-bit
    return WKV6_OK;
}

// Scratch for callers that pass no workspace (the reference-signature entry points have no such argument): a stream-ordered
// allocation (hipMallocFromPoolAsync on the caller's stream, hipFreeAsync right behind the launches that use it), so concurrent
// streams never share a buffer and nothing synchronises the device.  The memory comes from a pool this library owns (one per
// device, created on first use), not from the device's default pool, whose settings belong to the host application; the pool
// keeps up to SCRATCH_KEEP bytes between calls (the steady state of ordinary shapes is an O(1) pool hit) and gives the rest back.
// Under stream capture the allocation becomes part of the graph like any other stream-ordered allocation; callers that
// replay graphs should pass a workspace instead (every *_ex entry point takes one).
constexpr unsigned long long SCRATCH_KEEP = 1ull << 30;
struct StreamScratch {
    void* ptr = nullptr;
    hipStream_t st = nullptr;
    static hipMemPool_t pool_of(int dev)
    {
        static std::mutex mu;
        static hipMemPool_t pools[64] = {};
        if (dev < 0 || dev >= 64) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (!pools[dev]) {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t pool = nullptr;
            if (hipMemPoolCreate(&pool, &props) != hipSuccess) return nullptr;
            unsigned long long keep = SCRATCH_KEEP;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            pools[dev] = pool;
        }
        return pools[dev];
    }
    void* get(size_t bytes, hipStream_t stream)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        st = stream;
        hipMemPool_t pool = pool_of(dev);
        const hipError_t e = pool ? hipMallocFromPoolAsync(&ptr, bytes, pool, st) : hipMallocAsync(&ptr, bytes, st);
        if (e != hipSuccess) ptr = nullptr;
        return ptr;
    }
    ~StreamScratch()
    {
        if (ptr) (void)hipFreeAsync(ptr, st);
    }
};

// lens[b] = 1 + index of the first zero of mask[b][:], or T when the row has no zero
// (cuda/wkv6_bi_cuda.cu:21-69 breaks AFTER processing the first masked token).
__global__ void mask_to_lens_kernel(const int* __restrict__ mask, int* __restrict__ lens, int T)
{
    const int b = blockIdx.x;
    int first = T;                                           // "no zero" sentinel
    for (int t = threadIdx.x; t < T; t += blockDim.x)
        if (mask[(long)b * T + t] == 0) { first = t; break; }
    __shared__ int red[256];
    red[threadIdx.x] = first;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) lens[b] = min(red[0] + 1, T);
}

// order[rank] = b with the rows ranked by decreasing length (ties: by index): workgroups are dispatched in blockIdx order, so the
// longest sequences start first and the short ones fill the tail (list scheduling; ~9 % on ragged batches of 1536 workgroups)
__global__ void length_order_kernel(const int* __restrict__ lens, int* __restrict__ order, int B)
{
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int lb = lens[b];
        int rank = 0;
        for (int o = 0; o < B; ++o) {
            const int lo = lens[o];
            rank += (lo > lb) || (lo == lb && o < b);
        }
        order[rank] = b;
    }
}

int to_rc(hipError_t e) { return e == hipSuccess ? WKV6_OK : (int)e; }

// ---- forward over few, long sequences (inference prefill: B*H << CUs): two-level scan over T.  The sequence is cut into S
// segments that run as S times as many workgroups:
//   1. state pass per segment from a zero state: A_seg = the segment's own contribution to the state, and the per-channel sum
//      of its log-decays (ScanArgs::dsum);
//   2. one small kernel chains the segments:  S_in(seg+1) = 2^{dsum_seg} (.) S_in(seg) + A_seg  (the recurrence of
//      cuda/wkv6_cuda.cu:44-57 applied to whole segments), giving every segment its entry state and the final state;
//   3. the ordinary forward per segment from its entry state.
// 1.7x the work of one pass at S times the parallelism.  The backward still walks whole sequences; the forward's checkpoints
// for it keep their whole-sequence layout (ScanArgs::ckpt_segs).
__global__ void tsplit_combine_kernel(const void* s0, int s_f32, long s0_bstride, const float* __restrict__ A,
                                      const float* __restrict__ dsum, float* __restrict__ Sin, void* s_out, int H, int S)
{
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    for (int m = 0; m < HEAD * HEAD / 256; ++m) {
        const int e = threadIdx.x + 256 * m, i = e & (HEAD - 1);       // state layout [j][i]: i = key channel
        float cur = 0.f;
        if (s0) {
            const long o = (long)b * s0_bstride + (long)h * HEAD * HEAD + e;
            cur = s_f32 ? reinterpret_cast<const float*>(s0)[o] : bf_lo((uint32_t)reinterpret_cast<const bf16_t*>(s0)[o]);
        }
        for (int seg = 0; seg < S; ++seg) {
            const long bp = ((long)b * S + seg) * H + h;
            Sin[bp * HEAD * HEAD + e] = cur;
            float dl = 0.f;
            for (int q = 0; q < 4; ++q) dl += dsum[(bp * 4 + q) * HEAD + i];
            cur = fmaf(__builtin_amdgcn_exp2f(dl), cur, A[bp * HEAD * HEAD + e]);
        }
        if (s_out) {
            const long o = ((long)b * H + h) * HEAD * HEAD + e;
            if (s_f32) reinterpret_cast<float*>(s_out)[o] = cur;
            else reinterpret_cast<bf16_t*>(s_out)[o] = (bf16_t)(pack_bf2(cur, 0.f) & 0xffffu);
        }
    }
}

int tsplit_segments(const ScanArgs& a)
{
    if (a.lens || a.reverse || a.rev_n || a.order || a.accumulate || a.y_f32 || a.zero_tail || a.dsum || a.gn_out) return 1;
    if (a.g_f32[0] || a.g_f32[1] || a.g_f32[2] || a.g_f32[3]) return 1;
    int want = 0;
    if (const char* e = getenv("WKV6_TSPLIT")) {       // A/B switch: 0 / 1 = off, n = exactly n segments (if T divides)
        want = atoi(e);
        if (want <= 1) return 1;
        return a.T % (64 * want) == 0 ? want : 1;
    }
    const int cus = cu_count();
    int S = 1;
    while (2 * S <= 16 && (long)a.B * a.H * 2 * S <= cus && a.T % (64 * 2 * S) == 0 && a.T / (2 * S) >= 512) S *= 2;
    return S >= 4 ? S : 1;      // two segments do not pay for the extra state pass (two workgroups per pair serve that case)
}
// segment entry states of a two-level forward: state pass per segment from zero (A, dsum), chained by tsplit_combine_kernel
hipError_t tsplit_entry_states(const ScanArgs& a, int S, float* A, float* Sin, float* dsum, hipStream_t st)
{
    ScanArgs p = a;
    p.B = a.B * S; p.T = a.T / S;
    p.s0 = nullptr; p.s0_bstride = 0; p.s_out = A; p.state_f32 = 1; p.y = nullptr; p.dsum = dsum; p.ckpt = nullptr;
    if (hipError_t e = launch_chunk_state_pass(p, st)) return e;
    hipLaunchKernelGGL(tsplit_combine_kernel, dim3(a.B * a.H), dim3(256), 0, st, a.s0, a.state_f32, a.s0_bstride, A, dsum, Sin,
                       a.s_out, a.H, S);
    return hipGetLastError();
}

hipError_t chunk_forward(const ScanArgs& a_, hipStream_t st)
{
    ScanArgs a = a_;
    const int S = tsplit_segments(a);
    if (S <= 1) return launch_chunk_fwd(a, st);
    const size_t nstate = (size_t)a.B * S * a.H * HEAD * HEAD;                 // floats
    const size_t ndsum = (size_t)a.B * S * a.H * 4 * HEAD;
    StreamScratch scratch;
    float* const buf = reinterpret_cast<float*>(scratch.get((2 * nstate + ndsum) * sizeof(float), st));
    if (!buf) return hipErrorOutOfMemory;
    float* const A = buf, * const Sin = buf + nstate, * const dsum = buf + 2 * nstate;
    if (hipError_t e = tsplit_entry_states(a, S, A, Sin, dsum, st)) return e;
    ScanArgs p = a;
    p.B = a.B * S; p.T = a.T / S;
    p.s0 = Sin; p.s0_bstride = (long)a.H * HEAD * HEAD; p.state_f32 = 1; p.s_out = nullptr;
    p.ckpt_segs = S;                               // checkpoints (training forward) land in the whole sequences' slots
    return launch_chunk_fwd(p, st);
}

// ---- backward over few, long sequences as a two-level scan over T: the S segments of a sequence run as S
// workgroups.  What a segment needs from beyond its end is closed-form:
//   * the adjoint state G entering it from the future.  G_t = d_t (.) G_{t+1} + r_t gy_t^T is the forward recurrence
//     S_{t+1} = d_t (.) S_t + k_t v_t^T read backwards with k := r, v := gy: the state-only pass of the forward kernel over the
//     reversed segment gives its own contribution from a zero end state (and the decay sums), and the chaining kernel walks
//     the segments last to first, G_start(seg) = 2^{dsum_seg} (.) G_end(seg) + A_seg;  G_start(first) is gs;
//   * the gw suffix sum beyond its end: a_s - b_s = Phi_s - Phi_{s+1} with Phi_s[i] = sum_j G_s[i][j] S_s[i][j] (G_s the adjoint of
//     the state S_s in front of token s), so sum_{s >= end} (a_s - b_s) = Phi_end, from the entering G and the forward checkpoint at
//     the boundary;
//   * gu sums over the segments.
__global__ void tsplit_combine_rev_kernel(const float* __restrict__ A, const float* __restrict__ dsum, const float* __restrict__ ckpt,
                                          float* __restrict__ Gin, float* __restrict__ phi, void* gs, int gs_f32, int H, int S,
                                          long nslots_seg, int C)
{
    const int bh = blockIdx.x, b = bh / H, h = bh % H, tid = threadIdx.x;
    __shared__ float red[256];
    float cur[HEAD * HEAD / 256];
    for (int m = 0; m < HEAD * HEAD / 256; ++m) cur[m] = 0.f;
    for (int seg = S - 1; seg >= 0; --seg) {
        const long bp = ((long)b * S + seg) * H + h;
        float part = 0.f;
        // forward state at this segment's end = the checkpoint at the next segment's start (row order, wkv6_scan.h)
        const float* const ck = ckpt + (((long)(b * H + h) * S + (seg + 1)) * nslots_seg) * (HEAD * HEAD);
        for (int m = 0; m < HEAD * HEAD / 256; ++m) {
            const int e = tid + 256 * m, i = e & (HEAD - 1), j = e >> 6;          // state layout [j][i]
            Gin[bp * HEAD * HEAD + e] = cur[m];
            if (seg < S - 1) {
                const int jt = 2 * (j >> 5) + ((j >> 2) & 1), gb = (j >> 3) & 3, qb = j & 3;
                part = fmaf(cur[m], ck[((((i >> 4) * 4 + jt) * 64 + 16 * gb + (i & 15)) << 2) + qb], part);
            }
        }
        red[tid] = part;                                  // the four threads tid, tid + 64, ... share key row i = tid & 63
        __syncthreads();
        if (tid < HEAD) phi[((long)b * S + seg) * C + h * HEAD + tid] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
        __syncthreads();
        for (int m = 0; m < HEAD * HEAD / 256; ++m) {
            const int e = tid + 256 * m, i = e & (HEAD - 1);
            float dl = 0.f;
            for (int q = 0; q < 4; ++q) dl += dsum[(bp * 4 + q) * HEAD + i];
            cur[m] = fmaf(__builtin_amdgcn_exp2f(dl), cur[m], A[bp * HEAD * HEAD + e]);
        }
    }
    if (gs) {
        for (int m = 0; m < HEAD * HEAD / 256; ++m) {
            const long o = ((long)b * H + h) * HEAD * HEAD + tid + 256 * m;
            if (gs_f32) reinterpret_cast<float*>(gs)[o] = cur[m];
            else reinterpret_cast<bf16_t*>(gs)[o] = (bf16_t)(pack_bf2(cur[m], 0.f) & 0xffffu);
        }
    }
}
__global__ void tsplit_gu_kernel(const float* __restrict__ part, void* gu, int gu_f32, int S, int C)
{
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int seg = 0; seg < S; ++seg) s += part[((long)b * S + seg) * C + c];
        if (gu_f32) reinterpret_cast<float*>(gu)[(long)b * C + c] = s;
        else reinterpret_cast<bf16_t*>(gu)[(long)b * C + c] = (bf16_t)(pack_bf2(s, 0.f) & 0xffffu);
    }
}

hipError_t chunk_backward(const ScanArgs& a_, hipStream_t st)
{
    ScanArgs a = a_;
    const int S = tsplit_segments(a);
    if (S <= 1) return launch_chunk_bwd(a_, st);
    const int Ts = a.T / S;
    const size_t nstate = (size_t)a.B * S * a.H * HEAD * HEAD, ndsum = (size_t)a.B * S * a.H * 4 * HEAD, nrow = (size_t)a.B * S * a.C;
    StreamScratch scratch;
    float* const buf = reinterpret_cast<float*>(scratch.get((2 * nstate + ndsum + 2 * nrow) * sizeof(float), st));
    if (!buf) return hipErrorOutOfMemory;
    float* const A = buf, * const Gin = buf + nstate, * const dsum = buf + 2 * nstate, * const phi = dsum + ndsum, * const gup = phi + nrow;
    if (!a.ckpt_valid) {                               // self-contained: the forward's states first (two-level, like chunk_forward)
        ScanArgs f = a;
        f.y = nullptr; f.s_out = nullptr; f.gy = nullptr;
        if (hipError_t e = tsplit_entry_states(f, S, A, Gin, dsum, st)) return e;
        ScanArgs p = f;
        p.B = a.B * S; p.T = Ts;
        p.s0 = Gin; p.s0_bstride = (long)a.H * HEAD * HEAD; p.state_f32 = 1;
        p.ckpt_segs = S;
        if (hipError_t e = launch_chunk_state_pass(p, st)) return e;
    }
    {   // the segments' own adjoint contributions: state-only pass over the reversed segments with k := r, v := gy
        ScanArgs p = a;
        p.B = a.B * S; p.T = Ts;
        p.k = a.r; p.v = a.gy; p.reverse = 1;
        p.s0 = nullptr; p.s0_bstride = 0; p.s_out = A; p.state_f32 = 1; p.y = nullptr; p.dsum = dsum; p.ckpt = nullptr;
        if (hipError_t e = launch_chunk_state_pass(p, st)) return e;
    }
    hipLaunchKernelGGL(tsplit_combine_rev_kernel, dim3(a.B * a.H), dim3(256), 0, st, A, dsum, a.ckpt, Gin, phi, a.gs, a.part_f32, a.H, S,
                       (long)(Ts / 64), a.C);
    if (hipError_t e = hipGetLastError()) return e;
    ScanArgs p = a;
    p.B = a.B * S; p.T = Ts;
    p.s0 = nullptr; p.s0_bstride = 0;
    p.g_in = Gin; p.rc_in = phi; p.ckpt_segs = S; p.ckpt_valid = 1; p.split = 0;
    p.gu = a.gu ? gup : nullptr; p.gs = nullptr; p.part_f32 = 1;
    if (hipError_t e = launch_chunk_bwd12k(p, st)) return e;
    if (a.gu) {
        hipLaunchKernelGGL(tsplit_gu_kernel, dim3(a.B), dim3(256), 0, st, gup, a.gu, a.part_f32, S, a.C);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

// forward dispatch: chunked MFMA kernel for bf16 I/O unless the caller forces the exact scan
hipError_t run_fwd(const ScanArgs& a, unsigned flags, hipStream_t st)
{
    if ((flags & WKV6_IO_F32) || (flags & WKV6_ALGO_SCAN)) return launch_scan_fwd(a, (flags & WKV6_IO_F32) ? IO_F32 : IO_BF16, st);
    return chunk_forward(a, st);
}

hipError_t run_bwd(ScanArgs& a, unsigned flags, float* scratch, hipStream_t st)
{
    if ((flags & WKV6_IO_F32) || (flags & WKV6_ALGO_SCAN)) {
        a.aux = scratch;
        return launch_scan_bwd(a, flags & WKV6_IO_F32, st);
    }
    a.ckpt = scratch;
    a.ckpt_valid = (flags & WKV6_CKPT_VALID) ? 1 : 0;
    return chunk_backward(a, st);
}

ScanArgs base_args(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                   const void* w, const void* u, unsigned flags)
{
    ScanArgs a = {};
    a.B = B; a.T = T; a.C = C; a.H = H;
    a.r = r; a.k = k; a.v = v; a.w = w; a.u = u;
    a.wkind = (flags & WKV6_W_RAW) ? 1 : 0;
    a.part_f32 = (flags & WKV6_PARTIALS_F32) ? 1 : 0;
    a.use_u = 1;
    return a;
}

}  // namespace

#if defined(WKV6_STAMP) || defined(WKV6_CLOCK) || defined(WKV6_DEBUG)
namespace wkv6 { unsigned long long* g_stamp_buffer = nullptr; }
extern "C" void wkv6_set_debug_buffer(void* p) { wkv6::g_stamp_buffer = reinterpret_cast<unsigned long long*>(p); }
#endif

namespace wkv6 {
namespace {
__global__ void pass_marker_kernel() {}
// the clock ring: [2 kinds][n_launches][n_slots][4] uint64 of caller-owned device memory; the launchers of one process may run on
// several threads (the autograd engine's), so the ring's description and the launch counts are read and advanced under a lock
struct ClockRing {
    std::mutex mu;
    unsigned long long* buf = nullptr;
    int slots = 0, launches = 0;
    long count[2] = {0, 0};
} g_clock;
}
unsigned long long* clock_claim(int kind, int* slots)
{
    std::lock_guard<std::mutex> lk(g_clock.mu);
    *slots = g_clock.slots;
    if (!g_clock.buf) return nullptr;
    const long n = g_clock.count[kind]++;
    return g_clock.buf + ((size_t)kind * g_clock.launches + (size_t)(n % g_clock.launches)) * g_clock.slots * 4;
}
}

extern "C" {

const char* wkv6_amd_version(void) { return "0.1"; }

void wkv6_set_clock_ring(void* buf, int n_slots, int n_launches)
{
    std::lock_guard<std::mutex> lk(wkv6::g_clock.mu);
    const bool on = buf && n_slots > 0 && n_launches > 0;
    wkv6::g_clock.buf = on ? reinterpret_cast<unsigned long long*>(buf) : nullptr;
    wkv6::g_clock.slots = on ? n_slots : 0;
    wkv6::g_clock.launches = on ? n_launches : 0;
    wkv6::g_clock.count[0] = wkv6::g_clock.count[1] = 0;
}
void wkv6_set_clock_buffer(void* buf, int n_slots) { wkv6_set_clock_ring(buf, n_slots, 1); }
void wkv6_clock_ring_counts(long* fwd, long* bwd)
{
    std::lock_guard<std::mutex> lk(wkv6::g_clock.mu);
    if (fwd) *fwd = wkv6::g_clock.count[0];
    if (bwd) *bwd = wkv6::g_clock.count[1];
}
int wkv6_pass_marker(void* stream)
{
    hipLaunchKernelGGL(wkv6::pass_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return to_rc(hipGetLastError());
}

size_t wkv6_backward_workspace_bytes(int B, int T, int C, int H)
{
    // scan path: one fp32 [B,T,C] array; chunked path: one fp32 64x64 state per CKPT_TOK tokens and head
    const size_t scan = (size_t)B * T * C * sizeof(float);
    const size_t chunk = chunk_ckpt_floats(B, T, H) * sizeof(float);
    return align_up(scan > chunk ? scan : chunk);
}
// wkv6_bi workspace: lens, order | checkpoints / scratch of the forward-direction scan | ... of the reverse-direction scan |
//                    4 fp32 [B,T,C] side buffers (the forward uses the first one for y, the backward all four)
struct BiWorkspace {
    int* lens;
    int* order;
    float* scan[2];
    float* side[4];
};
static size_t bi_side_bytes(int B, int T, int C) { return align_up((size_t)B * T * C * sizeof(float)); }
// the part that lives from the forward to the backward (lens, order, both scans' checkpoints); the four side buffers behind it
// are per-call scratch
size_t wkv6bi_kept_bytes(int B, int T, int C, int H)
{
    return align_up((size_t)2 * B * sizeof(int)) + 2 * wkv6_backward_workspace_bytes(B, T, C, H);
}
static BiWorkspace bi_carve(void* workspace, void* side, int B, int T, int C, int H)
{
    char* p = reinterpret_cast<char*>(workspace);
    BiWorkspace w;
    w.lens = reinterpret_cast<int*>(p);
    w.order = w.lens + B;
    p += align_up((size_t)2 * B * sizeof(int));
    for (int i = 0; i < 2; ++i) { w.scan[i] = reinterpret_cast<float*>(p); p += wkv6_backward_workspace_bytes(B, T, C, H); }
    if (side) p = reinterpret_cast<char*>(side);
    for (int i = 0; i < 4; ++i) { w.side[i] = reinterpret_cast<float*>(p); p += bi_side_bytes(B, T, C); }
    return w;
}
size_t wkv6bi_workspace_bytes(int B, int T, int C, int H)
{
    return wkv6bi_kept_bytes(B, T, C, H) + 4 * bi_side_bytes(B, T, C);
}
// workspace == NULL: everything is stream-ordered scratch for the call; >= wkv6bi_workspace_bytes(): everything is carved from
// it; EXACTLY wkv6bi_kept_bytes(): the caller keeps only the part that must survive until the backward and the `nside` fp32 side
// buffers this call needs are stream-ordered scratch (an explicit choice: any other short size is an undersized buffer and is
// refused, so that a caller sized for an older layout gets WKV6_EWORKSPACE instead of hidden allocations).  Returns false when
// the workspace is refused or the allocation fails.
static bool bi_workspace(void* workspace, size_t workspace_bytes, int nside, int B, int T, int C, int H, hipStream_t st,
                         StreamScratch& scratch, BiWorkspace& ws)
{
    const size_t full = wkv6bi_workspace_bytes(B, T, C, H), kept = wkv6bi_kept_bytes(B, T, C, H);
    if (!workspace) {
        workspace = scratch.get(full, st);
        if (!workspace) return false;
        ws = bi_carve(workspace, nullptr, B, T, C, H);
    } else if (workspace_bytes >= full) {
        ws = bi_carve(workspace, nullptr, B, T, C, H);
    } else if (workspace_bytes == kept) {
        void* const side = scratch.get((size_t)nside * bi_side_bytes(B, T, C), st);
        if (!side) return false;
        ws = bi_carve(workspace, side, B, T, C, H);
    } else {
        return false;
    }
    return true;
}

int wkv6_forward_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                    const void* w, const void* u, const void* s0, void* s_out, void* y,
                    unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !y) return WKV6_ENULL;
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.s0 = s0;
    a.s0_bstride = (flags & WKV6_S0_PER_BATCH) ? (long)H * HEAD * HEAD : 0;
    a.s_out = s_out;
    a.y = y;
    return to_rc(run_fwd(a, flags, (hipStream_t)stream));
}

int wkv6_forward_ckpt_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                         const void* w, const void* u, const void* s0, void* s_out, void* y,
                         void* ckpt, size_t ckpt_bytes, unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !y || !ckpt) return WKV6_ENULL;
    if (flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN)) return WKV6_EUNSUPPORTED;
    if (ckpt_bytes < wkv6_backward_workspace_bytes(B, T, C, H)) return WKV6_EWORKSPACE;
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.s0 = s0;
    a.s0_bstride = (flags & WKV6_S0_PER_BATCH) ? (long)H * HEAD * HEAD : 0;
    a.s_out = s_out;
    a.y = y;
    a.ckpt = reinterpret_cast<float*>(ckpt);
    return to_rc(chunk_forward(a, (hipStream_t)stream));
}

int wkv6_forward_gn_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w, const void* u,
                       const void* s0, void* s_out, void* y, void* ckpt, size_t ckpt_bytes, const void* gate, const void* gamma,
                       const void* beta, float eps, void* out, float* stats, unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !gate || !gamma || !beta || !out) return WKV6_ENULL;
    if (flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN)) return WKV6_EUNSUPPORTED;
    if (ckpt && ckpt_bytes < wkv6_backward_workspace_bytes(B, T, C, H)) return WKV6_EWORKSPACE;
    if (want_split(B * H)) return WKV6_EUNSUPPORTED;      // two workgroups per (batch, head): a head's statistics span both
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.s0 = s0;
    a.s0_bstride = (flags & WKV6_S0_PER_BATCH) ? (long)H * HEAD * HEAD : 0;
    a.s_out = s_out;
    a.y = y;
    a.ckpt = reinterpret_cast<float*>(ckpt);
    a.gn_gate = gate; a.gn_gamma = gamma; a.gn_beta = beta; a.gn_eps = eps; a.gn_out = out; a.gn_stats = stats;
    return to_rc(launch_chunk_fwd(a, (hipStream_t)stream));
}

int wkv6_backward_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                     const void* w, const void* u, const void* s0, const void* gy, void* gr,
                     void* gk, void* gv, void* gw, void* gu, void* gs, void* workspace,
                     size_t workspace_bytes, unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !gy || !gr || !gk || !gv || !gw) return WKV6_ENULL;
    const size_t need = wkv6_backward_workspace_bytes(B, T, C, H);
    StreamScratch scratch;                     // released (stream-ordered) when this call returns
    if (!workspace) {
        workspace = scratch.get(need, (hipStream_t)stream);
        if (!workspace) return WKV6_EWORKSPACE;
    } else if (workspace_bytes < need) {
        return WKV6_EWORKSPACE;
    }
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.s0 = s0;
    a.s0_bstride = (flags & WKV6_S0_PER_BATCH) ? (long)H * HEAD * HEAD : 0;
    a.gy = gy; a.gr = gr; a.gk = gk; a.gv = gv; a.gw = gw; a.gu = gu; a.gs = gs;
    return to_rc(run_bwd(a, flags, reinterpret_cast<float*>(workspace), (hipStream_t)stream));
}

int wkv6_forward_rev_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w,
                        const void* u, void* y, void* ckpt, size_t ckpt_bytes, const int* rev_n, unsigned rev_mask,
                        unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !y || !rev_n) return WKV6_ENULL;
    if (rev_mask & ~(unsigned)REV_ALL) return WKV6_EUNSUPPORTED;
    if (ckpt && ckpt_bytes < wkv6_backward_workspace_bytes(B, T, C, H)) return WKV6_EWORKSPACE;
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.y = y;
    a.rev_n = rev_n;
    a.rev_mask = rev_mask;
    if (flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN))        // exact scan kernels (fp32 I/O, or forced): same index maps, no checkpoints
        return to_rc(launch_scan_fwd(a, (flags & WKV6_IO_F32) ? IO_F32 : IO_BF16, (hipStream_t)stream));
    a.ckpt = reinterpret_cast<float*>(ckpt);
    return to_rc(launch_chunk_fwd(a, (hipStream_t)stream));
}

int wkv6_backward_rev_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w,
                         const void* u, const void* gy, void* gr, void* gk, void* gv, void* gw, void* gu,
                         void* workspace, size_t workspace_bytes, const int* rev_n, unsigned rev_mask, unsigned flags,
                         void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !gy || !gr || !gk || !gv || !gw || !rev_n) return WKV6_ENULL;
    if (rev_mask & ~(unsigned)REV_ALL) return WKV6_EUNSUPPORTED;
    const size_t need = wkv6_backward_workspace_bytes(B, T, C, H);
    StreamScratch scratch;                     // released (stream-ordered) when this call returns
    if (!workspace) {
        workspace = scratch.get(need, (hipStream_t)stream);
        if (!workspace) return WKV6_EWORKSPACE;
    } else if (workspace_bytes < need) {
        return WKV6_EWORKSPACE;
    }
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.gy = gy; a.gr = gr; a.gk = gk; a.gv = gv; a.gw = gw; a.gu = gu;
    a.rev_n = rev_n;
    a.rev_mask = rev_mask;
    return to_rc(run_bwd(a, flags, reinterpret_cast<float*>(workspace), (hipStream_t)stream));
}

static int pair_args(int B, int T, int C, int H, const void* u, const wkv6_seq_set* s, unsigned flags, bool bwd, ScanArgs (&a)[2])
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!u || !s) return WKV6_ENULL;
    if (flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN)) return WKV6_EUNSUPPORTED;
    const size_t need = wkv6_backward_workspace_bytes(B, T, C, H);
    for (int i = 0; i < 2; ++i) {
        const wkv6_seq_set& q = s[i];
        if (!q.r || !q.k || !q.v || !q.w) return WKV6_ENULL;
        if (bwd ? (!q.gy || !q.gr || !q.gk || !q.gv || !q.gw || !q.ckpt) : !q.y) return WKV6_ENULL;
        if (q.rev_mask & ~(unsigned)REV_ALL) return WKV6_EUNSUPPORTED;
        if (q.ckpt && q.ckpt_bytes < need) return WKV6_EWORKSPACE;
        a[i] = base_args(B, T, C, H, q.r, q.k, q.v, q.w, u, flags);
        a[i].ckpt = reinterpret_cast<float*>(q.ckpt);
        a[i].rev_n = q.rev_n;
        a[i].rev_mask = q.rev_n ? q.rev_mask : 0u;
        if (bwd) {
            a[i].gy = q.gy; a[i].gr = q.gr; a[i].gk = q.gk; a[i].gv = q.gv; a[i].gw = q.gw; a[i].gu = q.gu;
            a[i].ckpt_valid = 1;
        } else {
            a[i].y = q.y;
        }
    }
    return WKV6_OK;
}

int wkv6_forward_pair_ex(int B, int T, int C, int H, const void* u, const wkv6_seq_set* s, unsigned flags, void* stream)
{
    ScanArgs a[2];
    if (int rc = pair_args(B, T, C, H, u, s, flags, false, a)) return rc;
    return to_rc(launch_chunk_fwd_pair(a[0], a[1], (hipStream_t)stream));
}

int wkv6_backward_pair_ex(int B, int T, int C, int H, const void* u, const wkv6_seq_set* s, unsigned flags, void* stream)
{
    ScanArgs a[2];
    if (int rc = pair_args(B, T, C, H, u, s, flags, true, a)) return rc;
    return to_rc(launch_chunk_bwd_pair(a[0], a[1], (hipStream_t)stream));
}

// The chunked halves of wkv6_bi address their fp32 side buffers with 32-bit byte offsets (buffer resources): (T + 128) C fp32 elements must
// stay below 2^32 bytes (where two workgroups serve a pair the side buffers are [B,T,C] arrays; the one-launch path's are [T][64]).
// Longer rows take the exact scan kernels, whose indices are 64-bit -- decided before the call enqueues anything.
static unsigned bi_route(unsigned flags, int T, int C)
{
    if (!(flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN)) && ((long)T + 128) * C >= (1L << 30)) flags |= WKV6_ALGO_SCAN;
    return flags;
}

int wkv6bi_forward_ex(int B, int T, int C, int H, const int* mask, const int* lens, const void* r,
                      const void* k, const void* v, const void* w, const void* u, void* y,
                      void* workspace, size_t workspace_bytes, unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !y || (!mask && !lens)) return WKV6_ENULL;
    flags = bi_route(flags, T, C);             // (before anything is enqueued)
    hipStream_t st = (hipStream_t)stream;
    StreamScratch scratch;                     // released (stream-ordered) when this call returns
    BiWorkspace ws;
    if (!bi_workspace(workspace, workspace_bytes, 1, B, T, C, H, st, scratch, ws)) return WKV6_EWORKSPACE;
    if (!lens) {
        hipLaunchKernelGGL(mask_to_lens_kernel, dim3(B), dim3(256), 0, st, mask, ws.lens, T);
        lens = ws.lens;
    }
    const bool chunked = !(flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN));
    const bool keep = (flags & WKV6_BI_KEEP_CKPT) && chunked;
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    if (chunked && B > 1 && B <= 4096) {
        hipLaunchKernelGGL(length_order_kernel, dim3(1), dim3(256), 0, st, lens, ws.order, B);
        a.order = ws.order;
    }
    a.y = y;
    a.y_f32 = ws.side[0];                     // the two halves are summed in fp32 and rounded once
    a.lens = lens;
    a.zero_tail = 1;                          // y[t > L_b] = 0 (the reference leaves it uninitialised, Q2)
    a.ckpt = keep ? ws.scan[0] : nullptr;
    ScanArgs a2 = a;
    a2.reverse = 1; a2.use_u = 0; a2.accumulate = 1; a2.zero_tail = 0;   // cuda/wkv6_bi_cuda.cu:71-111
    a2.ckpt = keep ? ws.scan[1] : nullptr;
    if (chunked) {          // both halves in one persistent launch (the fp32 partial of a row stays in the slot's scratch: L2 / Infinity Cache)
        const hipError_t e = launch_chunk_fwd_bi(a, a2, nullptr, st);
        if (e != hipErrorNotSupported) return to_rc(e);
    }
    if (hipError_t e = run_fwd(a, flags, st)) return (int)e;
    return to_rc(run_fwd(a2, flags, st));
}

int wkv6bi_backward_ex(int B, int T, int C, int H, const int* mask, const int* lens, const void* r,
                       const void* k, const void* v, const void* w, const void* u, const void* gy,
                       void* gr, void* gk, void* gv, void* gw, void* gu, void* workspace,
                       size_t workspace_bytes, unsigned flags, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!r || !k || !v || !w || !u || !gy || !gr || !gk || !gv || !gw || (!mask && !lens)) return WKV6_ENULL;
    flags = bi_route(flags, T, C);             // (before anything is enqueued)
    hipStream_t st = (hipStream_t)stream;
    StreamScratch scratch;                     // released (stream-ordered) when this call returns
    BiWorkspace ws;
    if (!bi_workspace(workspace, workspace_bytes, 4, B, T, C, H, st, scratch, ws)) return WKV6_EWORKSPACE;
    if (!lens) {
        hipLaunchKernelGGL(mask_to_lens_kernel, dim3(B), dim3(256), 0, st, mask, ws.lens, T);
        lens = ws.lens;
    }
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, flags);
    a.gy = gy; a.gr = gr; a.gk = gk; a.gv = gv; a.gw = gw; a.gu = gu;
    a.lens = lens;
    a.zero_tail = 1;
    if (!(flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN)) && B > 1 && B <= 4096) {
        hipLaunchKernelGGL(length_order_kernel, dim3(1), dim3(256), 0, st, lens, ws.order, B);
        a.order = ws.order;
    }
    if (!(flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN))) {
        // chunked bf16 path: the first half goes to fp32 side buffers, the second adds it and rounds once
        // (the reference accumulates `_gr[t] += F(gr)` in bf16, cuda/wkv6_bi_cuda.cu:199-200)
        for (int i = 0; i < 4; ++i) a.g_f32[i] = ws.side[i];
    }
    if (!(flags & (WKV6_IO_F32 | WKV6_ALGO_SCAN))) {
        // both halves in one persistent launch: the first half's four fp32 partials of a row stay in the slot's scratch
        ScanArgs a1 = a, a2 = a;
        a1.ckpt = ws.scan[0]; a2.ckpt = ws.scan[1];
        a1.ckpt_valid = a2.ckpt_valid = (flags & WKV6_CKPT_VALID) ? 1 : 0;
        a2.reverse = 1; a2.use_u = 0; a2.accumulate = 1; a2.zero_tail = 0; a2.gu = nullptr;
        const hipError_t e = launch_chunk_bwd_bi(a1, a2, nullptr, st);
        if (e != hipErrorNotSupported) return to_rc(e);
    }
    if (hipError_t e = run_bwd(a, flags, ws.scan[0], st)) return (int)e;             // adjoint of the forward scan
    a.reverse = 1; a.use_u = 0; a.accumulate = 1; a.zero_tail = 0; a.gu = nullptr;   // adjoint of the reverse scan
    return to_rc(run_bwd(a, flags, ws.scan[1], st));
}

// ---- reference-signature entry points ------------------------------------------------------------
int wkv6_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                      const float* w, const void* u, void* y, void* stream)
{
    return wkv6_forward_ex(B, T, C, H, r, k, v, w, u, nullptr, nullptr, y, WKV6_W_EW_F32, stream);
}
int wkv6_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                       const float* w, const void* u, const void* gy, void* gr, void* gk, void* gv,
                       void* gw, void* gu, void* stream)
{
    return wkv6_backward_ex(B, T, C, H, r, k, v, w, u, nullptr, gy, gr, gk, gv, gw, gu, nullptr,
                            nullptr, 0, WKV6_W_EW_F32, stream);
}
int wkv6bi_cuda_forward(int B, int T, int C, int H, const int* mask, const void* r, const void* k,
                        const void* v, const float* w, const void* u, void* y, void* stream)
{
    return wkv6bi_forward_ex(B, T, C, H, mask, nullptr, r, k, v, w, u, y, nullptr, 0, WKV6_W_EW_F32, stream);
}
int wkv6bi_cuda_backward(int B, int T, int C, int H, const int* mask, const void* r, const void* k,
                         const void* v, const float* w, const void* u, const void* gy, void* gr,
                         void* gk, void* gv, void* gw, void* gu, void* stream)
{
    return wkv6bi_backward_ex(B, T, C, H, mask, nullptr, r, k, v, w, u, gy, gr, gk, gv, gw, gu,
                              nullptr, 0, WKV6_W_EW_F32, stream);
}
int wkv6state_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                           const void* w, const void* u, const void* s, void* y, void* stream)
{
    if (!s) return WKV6_ENULL;
    return wkv6_forward_ex(B, T, C, H, r, k, v, w, u, s, nullptr, y, WKV6_W_RAW, stream);
}
int wkv6state_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                            const void* w, const void* u, const void* s, const void* gy, void* gr,
                            void* gk, void* gv, void* gw, void* gu, void* gs, void* stream)
{
    if (!s) return WKV6_ENULL;
    return wkv6_backward_ex(B, T, C, H, r, k, v, w, u, s, gy, gr, gk, gv, gw, gu, gs, nullptr, 0,
                            WKV6_W_RAW, stream);
}
int wkv6infctx_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                            const void* w, const void* u, void* s, void* y, void* stream)
{
    if (!s) return WKV6_ENULL;
    return wkv6_forward_ex(B, T, C, H, r, k, v, w, u, s, s, y, WKV6_W_RAW | WKV6_S0_PER_BATCH, stream);
}
int wkv6infctx_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                             const void* w, const void* u, const void* s, const void* gy, void* gr,
                             void* gk, void* gv, void* gw, void* gu, void* gs, void* stream)
{
    if (!s) return WKV6_ENULL;
    return wkv6_backward_ex(B, T, C, H, r, k, v, w, u, s, gy, gr, gk, gv, gw, gu, gs, nullptr, 0,
                            WKV6_W_RAW | WKV6_S0_PER_BATCH, stream);
}

static int rwkv6_infer(int B, int T, int C, int H, float* state, const void* r, const void* k, const void* v,
                       const float* w, const void* u, void* y, int io, void* stream)
{
    if (int rc = check_shape(B, T, C, H)) return rc;
    if (!state || !r || !k || !v || !w || !u || !y) return WKV6_ENULL;
    ScanArgs a = base_args(B, T, C, H, r, k, v, w, u, 0);
    a.wkind = 2;                                  // w is the decay itself
    a.state_f32 = 1;
    a.s0 = state; a.s_out = state;
    a.s0_bstride = (long)H * HEAD * HEAD;
    a.y = y;
    // prefill-sized calls in bf16 go through the chunked MFMA kernel (log of the given decay, fp32 state I/O); decode
    // (a few tokens), fp32 I/O and fp16 I/O (r, k, v are not exact in bf16) use the exact scan
    if (io == IO_BF16 && T >= 32) return to_rc(chunk_forward(a, (hipStream_t)stream));
    return to_rc(launch_scan_fwd(a, io, (hipStream_t)stream));
}
int rwkv6_cuda_forward_bf16(int B, int T, int C, int H, float* state, const void* r, const void* k, const void* v,
                            const float* w, const void* u, void* y, void* stream)
{
    return rwkv6_infer(B, T, C, H, state, r, k, v, w, u, y, IO_BF16, stream);
}
int rwkv6_cuda_forward_fp16(int B, int T, int C, int H, float* state, const void* r, const void* k, const void* v,
                            const float* w, const void* u, void* y, void* stream)
{
    return rwkv6_infer(B, T, C, H, state, r, k, v, w, u, y, IO_F16, stream);
}
int rwkv6_cuda_forward_fp32(int B, int T, int C, int H, float* state, const float* r, const float* k, const float* v,
                            const float* w, const float* u, float* y, void* stream)
{
    return rwkv6_infer(B, T, C, H, state, r, k, v, w, u, y, IO_F32, stream);
}

static int selftest_problem(int B, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int T = 83, H = 2, C = H * HEAD;
    const size_t n = (size_t)B * T * C, nu = (size_t)C, ngu = (size_t)B * C;
    const size_t ws = wkv6_backward_workspace_bytes(B, T, C, H);
    // layout of one device allocation (bf16 elements): r k v w gy | y[2] gr[2] gk[2] gv[2] gw[2] | u | gu[2] ; then workspace
    const size_t elems = 5 * n + 10 * n + nu + 2 * ngu;
    const size_t bytes = align_up(elems * 2);
    char* dev = nullptr;
    if (hipMalloc(&dev, bytes + ws) != hipSuccess) return WKV6_EWORKSPACE;
    std::vector<unsigned short> host(elems, 0);
    unsigned lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f; };   // U(-0.5, 0.5)
    auto to_bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
    for (size_t i = 0; i < n; ++i) {
        host[i] = to_bf(rnd());                      // r
        host[n + i] = to_bf(rnd());                  // k
        host[2 * n + i] = to_bf(rnd());              // v
        host[3 * n + i] = to_bf(-6.f + 7.f * (rnd() + 0.5f));   // w in [-6, 1]
        host[4 * n + i] = to_bf(rnd());              // gy
    }
    for (size_t i = 0; i < nu; ++i) host[15 * n + i] = to_bf(rnd());
    bf16_t* const p = reinterpret_cast<bf16_t*>(dev);
    hipError_t e = hipMemcpyAsync(dev, host.data(), elems * 2, hipMemcpyHostToDevice, st);
    int rc = e == hipSuccess ? WKV6_OK : (int)e;
    const unsigned base_flags = WKV6_W_RAW;
    for (int alg = 0; alg < 2 && rc == WKV6_OK; ++alg) {   // 0: chunked, 1: scan
        const unsigned fl = base_flags | (alg ? WKV6_ALGO_SCAN : 0u);
        rc = wkv6_forward_ex(B, T, C, H, p, p + n, p + 2 * n, p + 3 * n, p + 15 * n, nullptr, nullptr, p + (5 + alg) * n, fl, stream);
        if (rc == WKV6_OK)
            rc = wkv6_backward_ex(B, T, C, H, p, p + n, p + 2 * n, p + 3 * n, p + 15 * n, nullptr, p + 4 * n, p + (7 + alg) * n,
                                  p + (9 + alg) * n, p + (11 + alg) * n, p + (13 + alg) * n, p + 15 * n + nu + alg * ngu, nullptr,
                                  dev + bytes, ws, fl, stream);
    }
    if (rc == WKV6_OK) {
        e = hipMemcpyAsync(host.data(), dev, elems * 2, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = (int)e;
    }
    (void)hipFree(dev);
    if (rc != WKV6_OK) return rc;
    auto bf = [&](size_t i) { unsigned u = (unsigned)host[i] << 16; float f; memcpy(&f, &u, 4); return f; };
    auto differ = [&](size_t a0, size_t b0, size_t cnt, float ulps) {   // max |a-b| in bf16 ulps of the tensor scale
        float scale = 0.f, diff = 0.f;
        for (size_t i = 0; i < cnt; ++i) {
            scale = fmaxf(scale, fabsf(bf(b0 + i)));
            diff = fmaxf(diff, fabsf(bf(a0 + i) - bf(b0 + i)));
        }
        return !(diff <= ulps * scale * 0.0078125f);   // also catches NaN
    };
    int bad = 0;
    bad += differ(5 * n, 6 * n, n, 2.f);          // y
    bad += differ(7 * n, 8 * n, n, 2.f);          // gr
    bad += differ(9 * n, 10 * n, n, 2.f);         // gk
    bad += differ(11 * n, 12 * n, n, 2.f);        // gv
    bad += differ(13 * n, 14 * n, n, 4.f);        // gw
    bad += differ(15 * n + nu, 15 * n + nu + ngu, ngu, 2.f);   // gu
    return bad ? WKV6_ESELFTEST : WKV6_OK;
}

// Device self-test: (1) the cross-lane primitives, (2) the chunked MFMA kernels against the exact scan kernels on a
// fixed pseudo-random problem (B=2, T=83, H=2: ragged last block and stage) -- catches a miscompiled or mis-scheduled
// build (e.g. the mixed-shape MFMA accumulation hazard) at load time instead of in training.
// Returns 0, a positive count of failed primitive checks, or WKV6_ESELFTEST.
int wkv6_selftest(void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    {
        int* d = nullptr;
        if (hipMalloc(&d, sizeof(int)) != hipSuccess) return WKV6_EWORKSPACE;
        int host = -1;
        hipError_t e = hipMemsetAsync(d, 0, sizeof(int), st);
        if (e == hipSuccess) e = launch_selftest(d, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&host, d, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d);
        if (e != hipSuccess) return (int)e;
        if (host != 0) return host;
    }
    // two problems of T = 83 tokens (ragged last block and stage, an odd and an even stage of a 64-token pair), H = 2: B = 2 -- few
    // (batch, head) pairs: the launchers put two workgroups on each -- and the smallest B
    // that gets one workgroup per pair on this device (the mode of real shapes): both modes of wkv6_chunk_bwd12k.hip run
    const int cus = cu_count();
    const int batches[2] = {2, (cus > 0 ? cus : 256) / 4 + 1};
    for (int prob = 0; prob < 2; ++prob) {
        if (int rc = selftest_problem(batches[prob], stream)) return rc;
    }
    return WKV6_OK;
}

}  // extern "C"
