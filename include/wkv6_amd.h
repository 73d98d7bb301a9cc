/*
 * wkv6_amd.h -- C ABI of librwkv6_amd.so: the MI355X (gfx950) implementation of the RWKV-6 WKV
 * operator family of yynil/RWKV_LM_EXT.
 *
 * Every entry point takes plain device pointers and sizes (no torch types) plus the HIP stream to
 * launch on, and returns 0 on success or a negative WKV6_E* / positive hipError_t code; nothing is
 * launched when arguments are rejected.  All tensors are contiguous, layouts as in the reference:
 *   r,k,v,w,y,gy,gr,gk,gv,gw : [B,T,C]     u : [H,N]     gu : [B,C] (per-batch partials)
 *   N = C/H = 64 (the reference build's -D_N_, src/model.py:189)
 * bf16 buffers are passed as void* (raw bfloat16 bits).
 *
 * The first four pairs are drop-in replacements for the `cuda_forward` / `cuda_backward` C symbols
 * that the reference's torch-extension shims call; each has the reference's parameter list with
 * one trailing `stream`.  The *_ex entry points expose what the reference cannot express (bf16 raw
 * decay without the fp32 `ew` pass, fp32 I/O for numerics tests, separate in/out state, caller-owned
 * workspace, per-row lengths).  INTEGRATION.md shows the binding a maintainer would add.
 */
#ifndef WKV6_AMD_H
#define WKV6_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    WKV6_OK = 0,
    WKV6_EINVAL = -1,      /* bad shape: C != H*64, B/T/C/H < 1 (reference: assert(H*_N_ == C), cuda/wkv6_cuda.cu:231) */
    WKV6_ENULL = -2,       /* a required pointer is NULL */
    WKV6_EWORKSPACE = -3,  /* workspace too small / allocation failed */
    WKV6_EUNSUPPORTED = -4,
    WKV6_ESELFTEST = -5
};

/* ---- wkv6: replaces cuda_forward / cuda_backward of cuda/wkv6_op.cpp:5-6 (cuda/wkv6_cuda.cu:229-242).
 * `w` is the fp32 tensor ew = -exp(w_raw) that src/model.py:210 builds. */
int wkv6_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                      const float* w, const void* u, void* y, void* stream);
int wkv6_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                       const float* w, const void* u, const void* gy, void* gr, void* gk, void* gv,
                       void* gw, void* gu, void* stream);

/* ---- wkv6_bi: replaces cuda_forward / cuda_backward of cuda/wkv6_bi_op.cpp:5-6
 * (cuda/wkv6_bi_cuda.cu:363-377).  mask: int32 [B,T]; both scans cover tokens 0..L_b where L_b is
 * the first t with mask[b][t]==0 (T-1 if the row has no zero); y and the gradients are 0 for
 * t > L_b; the backward is the exact adjoint of the forward (deviations Q1-Q3). */
int wkv6bi_cuda_forward(int B, int T, int C, int H, const int* mask, const void* r, const void* k,
                        const void* v, const float* w, const void* u, void* y, void* stream);
int wkv6bi_cuda_backward(int B, int T, int C, int H, const int* mask, const void* r, const void* k,
                         const void* v, const float* w, const void* u, const void* gy, void* gr,
                         void* gk, void* gv, void* gw, void* gu, void* stream);

/* ---- wkv6state: replaces cuda/wkv6state_op.cpp:5-6.  `w` is the RAW bf16 decay parameter
 * (cuda/wkv6state_cuda.cu:30), s: bf16 [H,N,N] (value-major: s[h][j][i]), gs: bf16 [B,H,N,N]. */
int wkv6state_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                           const void* w, const void* u, const void* s, void* y, void* stream);
int wkv6state_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                            const void* w, const void* u, const void* s, const void* gy, void* gr,
                            void* gk, void* gv, void* gw, void* gu, void* gs, void* stream);

/* ---- wkv6infctx: replaces cuda/wkv6infctx_op.cpp:5-6.  s: bf16 [B,H,N,N]; the forward overwrites
 * it with the final state (cuda/wkv6infctx_cuda.cu:65-67).  The backward must be given the INITIAL
 * state (the reference hands it the overwritten one; the python wrapper keeps a copy). */
int wkv6infctx_cuda_forward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                            const void* w, const void* u, void* s, void* y, void* stream);
int wkv6infctx_cuda_backward(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                             const void* w, const void* u, const void* s, const void* gy, void* gr,
                             void* gk, void* gv, void* gw, void* gu, void* gs, void* stream);

/* ---- rwkv6 (stateful forward-only inference): replaces cuda_forward_bf16 / cuda_forward_fp16 / cuda_forward_fp32 of
 * cuda/rwkv6_op.cpp:8-10 (cuda/rwkv6.cu:8-87).  `w` is the fp32 DECAY exp(-exp(w_raw)) (src/model_run.py:64),
 * `state` is fp32 [B,H,N,N] (value-major, like s above; [H,N,N] for B = 1 as the reference uses it) and is updated
 * in place.  The reference indexes the state without the batch (`wrong if B > 1`, cuda/rwkv6.cu:17); here every
 * batch row has its own state.  The fp16 flavour takes r, k, v, u, y in IEEE half: inputs are widened to fp32 in the
 * kernel (exact), the arithmetic and the state are fp32 and y is rounded to nearest even, as cuda/rwkv6.cu:8-71. */
int rwkv6_cuda_forward_bf16(int B, int T, int C, int H, float* state, const void* r, const void* k, const void* v,
                            const float* w, const void* u, void* y, void* stream);
int rwkv6_cuda_forward_fp16(int B, int T, int C, int H, float* state, const void* r, const void* k, const void* v,
                            const float* w, const void* u, void* y, void* stream);
int rwkv6_cuda_forward_fp32(int B, int T, int C, int H, float* state, const float* r, const float* k, const float* v,
                            const float* w, const float* u, float* y, void* stream);

/* ---- extended entry points -------------------------------------------------------------------------
 * flags (OR together): */
enum {
    WKV6_W_EW_F32 = 0,      /* w is fp32 ew = -exp(w_raw)                       (default) */
    WKV6_W_RAW = 1,         /* w is the raw decay in the I/O type               */
    WKV6_IO_F32 = 2,        /* every bf16 tensor is fp32 instead (numerics tests) */
    WKV6_S0_PER_BATCH = 4,  /* s0 is [B,H,N,N] (infctx) instead of [H,N,N] (state) */
    WKV6_ALGO_SCAN = 16,    /* force the exact token-serial kernels            */
    WKV6_CKPT_VALID = 32,   /* backward: `workspace` already holds the checkpoints written by wkv6_forward_ckpt_ex
                               (wkv6_bi: by wkv6bi_forward_ex with WKV6_BI_KEEP_CKPT) for the same inputs, so the backward
                               skips its own state pass(es) */
    WKV6_BI_KEEP_CKPT = 64, /* wkv6bi_forward_ex: also store the state checkpoints of both scans in `workspace` (which the
                               caller then hands to wkv6bi_backward_ex with WKV6_CKPT_VALID) */
    WKV6_PARTIALS_F32 = 128 /* backward: gu [B,C] and gs [B,H,N,N] are fp32 buffers -- they are per-batch partial sums that the
                               caller reduces over the batch, so keeping them unrounded lets the parameter gradient be rounded
                               once (the reference ABI, bf16 partials, rounds twice: src/model.py:181, 232) */
};
/* Bytes of scratch the backward needs (the forward needs none). */
size_t wkv6_backward_workspace_bytes(int B, int T, int C, int H);

/* s0 may be NULL (zero initial state); s_out may be NULL; s_out may alias s0. */
int wkv6_forward_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                    const void* w, const void* u, const void* s0, void* s_out, void* y,
                    unsigned flags, void* stream);
/* Same as wkv6_forward_ex, and additionally stores the forward state every 64 tokens (fp32, 4 B per token-channel,
 * wkv6_backward_workspace_bytes() bytes in all) into `ckpt` -- the activation checkpoint a following
 * wkv6_backward_ex(..., workspace = ckpt, flags | WKV6_CKPT_VALID) consumes.  bf16 I/O, chunked kernels only;
 * returns WKV6_EUNSUPPORTED for WKV6_IO_F32 / WKV6_ALGO_SCAN. */
int wkv6_forward_ckpt_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                         const void* w, const void* u, const void* s0, void* s_out, void* y,
                         void* ckpt, size_t ckpt_bytes, unsigned flags, void* stream);
/* wkv6_forward_ckpt_ex with the per-head GroupNorm and the gate multiply that follow the operator in the time-mix block
 * (src/model.py:462-468: x = ln_x(x.view(B*T, C)).view(B, T, C); output(x * g)) fused into its store:
 *     out = GroupNorm_H(y; gamma, beta, eps) * gate,       y = the operator's bf16 output, exactly what nn.GroupNorm would see.
 * y may be NULL (inference: only `out` is written); ckpt may be NULL; stats (fp32 [B*T, H, 2]: mean, rstd per token and head,
 * what wkv6_gn_gate_backward consumes) may be NULL.  gate [B,T,C], gamma / beta [C], bf16.  bf16 I/O, chunked kernels only.
 * Returns WKV6_EUNSUPPORTED where the fusion does not apply (WKV6_IO_F32 / WKV6_ALGO_SCAN; so few (batch, head) pairs that two
 * workgroups share one): the caller then runs wkv6_forward_ckpt_ex + wkv6_gn_gate_forward. */
int wkv6_forward_gn_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w, const void* u,
                       const void* s0, void* s_out, void* y, void* ckpt, size_t ckpt_bytes, const void* gate, const void* gamma,
                       const void* beta, float eps, void* out, float* stats, unsigned flags, void* stream);
/* gu, gs may be NULL (skipped).  workspace: wkv6_backward_workspace_bytes() bytes, or NULL: the library takes a stream-ordered
 * allocation on `stream` for the duration of the call (hipMallocAsync / hipFreeAsync; safe from any number of streams). */
int wkv6_backward_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v,
                     const void* w, const void* u, const void* s0, const void* gy, void* gr,
                     void* gk, void* gv, void* gw, void* gu, void* gs, void* workspace,
                     size_t workspace_bytes, unsigned flags, void* stream);
/* lens: int32 [B] device array, number of leading tokens both scans cover (NULL: derive from mask).
 * workspace: at least wkv6bi_workspace_bytes() bytes (NULL: stream-ordered allocation for the duration of the call), or EXACTLY
 * wkv6bi_kept_bytes() bytes -- the part that must live from a WKV6_BI_KEEP_CKPT forward to its backward (row lengths and the two
 * scans' checkpoints); the fp32 [B,T,C] side buffers (one in the forward, four in the backward) are then stream-ordered scratch
 * of the call.  Any other size is refused with WKV6_EWORKSPACE. */
int wkv6bi_forward_ex(int B, int T, int C, int H, const int* mask, const int* lens, const void* r,
                      const void* k, const void* v, const void* w, const void* u, void* y,
                      void* workspace, size_t workspace_bytes, unsigned flags, void* stream);
int wkv6bi_backward_ex(int B, int T, int C, int H, const int* mask, const int* lens, const void* r,
                       const void* k, const void* v, const void* w, const void* u, const void* gy,
                       void* gr, void* gk, void* gv, void* gw, void* gu, void* workspace,
                       size_t workspace_bytes, unsigned flags, void* stream);
size_t wkv6bi_workspace_bytes(int B, int T, int C, int H);
size_t wkv6bi_kept_bytes(int B, int T, int C, int H);

/* ---- partially reversed sequences: replaces the torch.gather round trips around the operator in the
 * bidirectional compositions -- src/model_bi.py:331-348 (k, v reversed, y un-reversed) and src/model_ext.py:410-437 (every
 * tensor reversed).  For batch row b, tokens [0, rev_n[b]) of the tensors named in rev_mask are read in reverse order (scan
 * position p < rev_n[b] <-> token rev_n[b]-1-p, exactly reverse_x_idx of src/model_ext.py:410-417); positions >= rev_n[b]
 * keep their place and ARE scanned (the sentence-embedding position sits right behind the reversed span).  WKV6_REV_Y applies
 * to y (written through the map) and, in the backward, to gy; each gradient follows its tensor's bit.  rev_n: int32 [B] on
 * the device.  bf16 I/O on the chunked kernels; WKV6_IO_F32 / WKV6_ALGO_SCAN run the exact scan kernels with the same index
 * maps (ckpt is ignored there).  ckpt may be NULL. */
enum { WKV6_REV_R = 1, WKV6_REV_K = 2, WKV6_REV_V = 4, WKV6_REV_W = 8, WKV6_REV_Y = 16 };
int wkv6_forward_rev_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w,
                        const void* u, void* y, void* ckpt, size_t ckpt_bytes, const int* rev_n, unsigned rev_mask,
                        unsigned flags, void* stream);
int wkv6_backward_rev_ex(int B, int T, int C, int H, const void* r, const void* k, const void* v, const void* w,
                         const void* u, const void* gy, void* gr, void* gk, void* gv, void* gw, void* gu,
                         void* workspace, size_t workspace_bytes, const int* rev_n, unsigned rev_mask, unsigned flags,
                         void* stream);

/* ---- both operator calls of a bidirectional time-mix layer in ONE launch (second half): the
 * forward-direction call and the reversed-direction call of src/model_bi.py:331-348 (composition B: same r, w, reversed k, v, y)
 * and src/model_ext.py:421-437 (composition C: separately projected, fully reversed) are two problems of one shape; a grid of
 * 2 B H workgroups serves problem s[0] with its first B H slots and s[1] with the rest -- one launch, one tail.  Each set names its
 * own tensors, reversal map (rev_n NULL = none) and checkpoint buffer (wkv6_backward_workspace_bytes() bytes each; required by
 * the backward, which must find them filled by wkv6_forward_pair_ex; may be NULL in a forward nobody differentiates).  u [H,N]
 * is shared.  bf16 I/O, chunked kernels; flags: WKV6_W_RAW / WKV6_PARTIALS_F32 as elsewhere (WKV6_EUNSUPPORTED with WKV6_IO_F32 /
 * WKV6_ALGO_SCAN).  Results are bit-identical to two wkv6_forward_rev_ex / wkv6_backward_rev_ex calls. */
typedef struct wkv6_seq_set {
    const void *r, *k, *v, *w;          /* [B,T,C] inputs */
    void* y;                            /* forward: [B,T,C] output */
    const void* gy;                     /* backward: [B,T,C] */
    void *gr, *gk, *gv, *gw;            /* backward: [B,T,C] gradients */
    void* gu;                           /* backward: [B,C] per-batch partials of this problem (bf16, or fp32 with WKV6_PARTIALS_F32) */
    void* ckpt;                         /* state checkpoints written by the forward, read by the backward */
    size_t ckpt_bytes;
    const int* rev_n;                   /* int32 [B] on the device, or NULL */
    unsigned rev_mask;                  /* WKV6_REV_* */
} wkv6_seq_set;
int wkv6_forward_pair_ex(int B, int T, int C, int H, const void* u, const wkv6_seq_set* s, unsigned flags, void* stream);
int wkv6_backward_pair_ex(int B, int T, int C, int H, const void* u, const wkv6_seq_set* s, unsigned flags, void* stream);

/* ---- elementwise neighbours of the operator in the RWKV-6 time-mix block; bf16 only ----
 * ddlerp (src/model.py:435-448): xx = shift(x) - x; out[s] = x + xx * (maa[s] + m[s]), s < NS.
 *   x [B,T,C]; shifted0 [B,C] = token in front of each row (NULL: zero, nn.ZeroPad2d((0,0,1,-1))); m [NS,B,T,C] or NULL;
 *   maa [NS,C]; out [NS,B,T,C].  Supported: (NS=1, m NULL or not), (NS=5, m given).
 * backward: dx [B,T,C], dm [NS,B,T,C] (NULL iff m NULL), dmaa_part fp32 [nparts,NS,C] partial sums (caller adds them). */
int wkv6_ddlerp_forward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                        void* out, void* stream);
int wkv6_ddlerp_backward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                         const void* dout, void* dx, void* dm, float* dmaa_part, int nparts, void* stream);
/* The same with the shift taken over the stream "first rev_n[b] tokens of row b reversed, the rest in place" while x, m, out
 * stay in the original token order (row n2: the reversed half of src/model_ext.py:421-437 without gathering x).  rev_n: int32
 * [B] on the device, NULL = plain shift. */
int wkv6_ddlerp_rev_forward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                            const int* rev_n, void* out, void* stream);
int wkv6_ddlerp_rev_backward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                             const int* rev_n, const void* dout, void* dx, void* dm, float* dmaa_part, int nparts, void* stream);
/* gn_gate (src/model.py:462-468): out = GroupNorm_H(y; gamma, beta, eps) * g on rows of C = 64 H channels (nn.GroupNorm(H, C)
 * applied to [rows, C]); stats fp32 [rows,H,2] (mean, rstd) is written for the backward (may be NULL in inference).
 * backward: dy, dg [rows,C]; dgamma_part, dbeta_part fp32 [nparts,C] partial sums. */
int wkv6_gn_gate_forward(long rows, int C, int H, const void* y, const void* g, const void* gamma, const void* beta,
                         float eps, void* out, float* stats, void* stream);
int wkv6_gn_gate_backward(long rows, int C, int H, const void* y, const void* g, const void* gamma, const void* beta,
                          const float* stats, const void* dout, void* dy, void* dg, float* dgamma_part, float* dbeta_part,
                          int nparts, void* stream);

/* Elementwise neighbours of the channel-mix FFN's GEMMs (src/model.py:636-644), bf16, n elements (a multiple of 8), one pass each:
 * sqrelu: out = relu(x)^2, dx = 2 relu(x) dout;  sigmul: out = sigmoid(r) * kv, dr = dout kv s (1 - s), dkv = dout s.
 * (The FFN's token shift and its two lerps are wkv6_ddlerp_* with NS = 2, m = NULL.) */
int wkv6_sqrelu_forward(long n, const void* x, void* out, void* stream);
int wkv6_sqrelu_backward(long n, const void* x, const void* dout, void* dx, void* stream);
int wkv6_sigmul_forward(long n, const void* r, const void* kv, void* out, void* stream);
int wkv6_sigmul_backward(long n, const void* r, const void* kv, const void* dout, void* dr, void* dkv, void* stream);

/* Device self-test: the cross-lane primitives, then the chunked MFMA kernels against the exact scan kernels on two fixed
 * pseudo-random problems -- one small enough that two workgroups serve a (batch, head) pair, one with one workgroup per pair, so
 * that both backward kernels run -- (forward and backward, all outputs within 2 bf16 ulps of the tensor scale, 4 for gw).
 * Returns 0 when it passes, WKV6_ESELFTEST (or the number of failed primitive checks) otherwise. */
int wkv6_selftest(void* stream);
/* Measurement aids (bench.py; no effect on results).
 * wkv6_set_clock_ring: while `buf` (device memory, 2 * n_launches * n_slots * 4 uint64) is set, wave 0 of the first n_slots workgroups of
 * the n-th chunked forward launch since the call writes {s_memtime, s_memrealtime} at its start and its end into
 * buf[((n % n_launches) * n_slots + slot) * 4 .. + 3], and of the n-th chunked backward launch into the second half of buf likewise:
 * the in-kernel shader clock of a launch is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6), its
 * duration max(end s_memrealtime) - min(start s_memrealtime) over the slots.  buf = NULL (the default) switches it off: the kernels
 * then execute one scalar branch for it and no stamp.  Process-wide; the caller keeps `buf` alive until it has switched the probe off.
 * wkv6_set_clock_buffer(buf, n_slots) = wkv6_set_clock_ring(buf, n_slots, 1): every launch overwrites the one before it.
 * wkv6_clock_ring_counts: chunked forward / backward launches since the ring was set.
 * wkv6_pass_marker: launches an empty kernel named wkv6::pass_marker_kernel on `stream`: a phase boundary in a profiler's
 * dispatch list. */
void wkv6_set_clock_ring(void* buf, int n_slots, int n_launches);
void wkv6_set_clock_buffer(void* buf, int n_slots);
void wkv6_clock_ring_counts(long* fwd, long* bwd);
int wkv6_pass_marker(void* stream);
/* "major.minor" of the library. */
const char* wkv6_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WKV6_AMD_H */
