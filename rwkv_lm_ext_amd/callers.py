"""Host-side callers of the WKV6 operator: what sits immediately on either side of the hot path in the reference.

Parameter names equal the reference's state_dict keys, so a reference checkpoint loads with ``load_state_dict``.

  reverse helpers      src/model_ext.py:398-419 (create_mask, reverse_x_idx, reverse_x)            SURVEY a12
  Tmix_x060            src/model.py:376-477 == src/model_encoder_run.py:96-186 (time-mix)          SURVEY a13
  CMix_x060            src/model.py:616-644 == src/model_encoder_run.py:189-219 (channel-mix)      SURVEY a14
  Tmix_x060.forward_bi_c   composition C: (WKV(x) + unrev(WKV(rev x))) / 2, src/model_ext.py:421-437    SURVEY a15
  Tmix_x060.forward_bi_b   composition B: WKV(r,k,v,w,u) + unrev(WKV(r, rev k, rev v, w, u)), src/model_bi.py:325-350  SURVEY a16
  BiBlock / RwkvEncoder    src/model_encoder_run.py:222-348 (encoder with sentence embedding at the first emb_id)  SURVEY a15
  pooling / info_nce_loss  src/model_ext.py:1708-1738, 1882-1911                                       SURVEY a17

The WKV call itself is `wkv(B, T, C, H, r, k, v, w, u) -> y` (default: rwkv_lm_ext_amd.wkv.RUN_CUDA_RWKV6, the HIP
kernels; bf16 on the GPU).  Everything else is plain PyTorch (the GEMMs ride rocBLAS), as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- a12: padding mask and in-row reversal ---------------------------------------------------------------------
def create_mask(x, emb_id=1, pad_id=0):
    """1 for ordinary tokens, 0 for pad and for the embedding marker (src/model_encoder_run.py:7-11)."""
    return ((x != pad_id) & (x != emb_id)).to(torch.int)


def reverse_x_idx(mask, max_len):
    """Per row: indices that reverse the first sum(mask) positions and keep the rest (src/model_ext.py:410-417).
    Vectorised (the reference loops over the batch in python)."""
    n = mask.sum(dim=1, keepdim=True).to(torch.long)                      # [B,1]
    pos = torch.arange(max_len, device=mask.device).unsqueeze(0)         # [1,T]
    return torch.where(pos < n, n - 1 - pos, pos)


def reverse_x(x, rev_idx):
    return torch.gather(x, 1, rev_idx.to(x.device).unsqueeze(-1).expand(-1, -1, x.size(-1)))


def _default_wkv(B, T, C, H, r, k, v, w, u):
    from .wkv import RUN_CUDA_RWKV6
    bf = torch.bfloat16
    y = RUN_CUDA_RWKV6(B, T, C, H, *(t.to(bf).contiguous() for t in (r, k, v, w, u)))
    return y.to(r.dtype)


class Tmix_x060(nn.Module):
    """RWKV-6 time-mix around the WKV operator (src/model.py:376-477)."""

    def __init__(self, n_embd, dim_att, head_size=64, head_size_divisor=8, wkv=None, fused=None):
        """fused: None = the HIP elementwise kernels of mix_op whenever the input is a bf16 GPU tensor; True / False force."""
        super().__init__()
        self.n_head = dim_att // head_size
        self.wkv = wkv or _default_wkv
        self.fused = fused
        # forward(): GroupNorm * gate inside the operator's forward kernel (wkv.WKV_6_GN) instead of a second kernel.  Off by
        # default: it saves the y round trip through HBM but the layer is not faster for it -- B x T = 48 x 512, C = 2048 on
        # MI355X: fwd+bwd 6.644 vs 6.626 ms, forward only 1.612 vs 1.593 ms (profiles/r03_tmix_layer_epilogue.txt): the 56 us
        # GroupNorm kernel runs at 5.4 TB/s, and the statistics exchange lengthens the forward's consumer waves by as much.
        self.fuse_epilogue = False
        # forward_bi_b / forward_bi_c with in-kernel reversal: both operator calls of the layer in one launch per pass
        # (wkv.WKV_6_PAIR) instead of two
        self.pair_launch = True
        d_mix = 64 if n_embd == 4096 else 32                              # TIME_MIX_EXTRA_DIM
        d_decay = 128 if n_embd == 4096 else 64                           # TIME_DECAY_EXTRA_DIM
        z = lambda *s: nn.Parameter(torch.zeros(*s))
        for n in ("x", "w", "k", "v", "r", "g"):
            setattr(self, "time_maa_" + n, z(1, 1, n_embd))
        self.time_maa_w1 = z(n_embd, d_mix * 5)
        self.time_maa_w2 = z(5, d_mix, n_embd)
        self.time_decay = z(1, 1, dim_att)
        self.time_decay_w1 = z(n_embd, d_decay)
        self.time_decay_w2 = z(d_decay, dim_att)
        self.time_faaaa = z(self.n_head, head_size)
        self.receptance = nn.Linear(n_embd, dim_att, bias=False)
        self.key = nn.Linear(n_embd, dim_att, bias=False)
        self.value = nn.Linear(n_embd, dim_att, bias=False)
        self.output = nn.Linear(dim_att, n_embd, bias=False)
        self.gate = nn.Linear(n_embd, dim_att, bias=False)
        self.ln_x = nn.GroupNorm(self.n_head, dim_att, eps=1e-5 * head_size_divisor ** 2)

    def _maa5(self):
        """[5,C]: the static lerp weights of the decay, key, value, receptance and gate inputs, in the order the
        low-rank correction tensor is laid out (src/model.py:441-442: mw, mk, mv, mr, mg)."""
        return torch.cat([self.time_maa_w, self.time_maa_k, self.time_maa_v, self.time_maa_r, self.time_maa_g], 0).view(5, -1)

    def _use_fused(self, x):
        """The fused HIP kernels serve bf16 GPU activations with bf16 GPU parameters and rows of at most 4096 channels in
        multiples of 64 (csrc/wkv6_mix.hip: check_rows); anything else (fp32 parameters under autocast, a partly cast model,
        wider rows) takes the eager path unless `fused` forces a choice."""
        if self.fused is None:
            C = x.shape[-1]
            ok = lambda t: t.is_cuda and t.dtype == torch.bfloat16
            return (ok(x) and ok(self.time_maa_x) and ok(self.time_maa_w) and ok(self.ln_x.weight)
                    and C % 64 == 0 and C <= 4096)
        return self.fused

    def jit_func(self, x, shifted=None, rev_n=None):
        """Inputs of the WKV operator from the block input (src/model.py:435-459): every projection reads its own
        data-dependent blend of x_t and x_{t-1},  x + (x_{t-1} - x) * (maa_s + m_s),  where the five corrections m_s come
        from one shared low-rank pair (tanh(blend_x @ W1) -> per-stream W2).  Then r, k, v = Linear(blend), g = silu(Linear),
        w = time_decay + tanh(blend_w @ D1) @ D2.
        `shifted`: x delayed by one token; default zero-padded (nn.ZeroPad2d((0,0,1,-1))), the infctx path passes the
        previous chunk's last token in front (src/model.py:740-741).
        On bf16 GPU tensors the two blend stages are one HIP kernel each (mix_op.ddlerp).
        `rev_n` (fused path only, int32 [B]): the token shift runs over the stream whose first rev_n[b] tokens are reversed
        while every tensor stays in the original token order (row n2)."""
        B, T, C = x.size()
        if self._use_fused(x):
            from . import mix_op
            first = None if shifted is None else shifted[:, 0].contiguous()
            lead = mix_op.ddlerp(x, self.time_maa_x.view(1, C), None, first, rev_n)[0]
            low = torch.tanh(lead @ self.time_maa_w1).view(B * T, 5, -1).transpose(0, 1)
            corr = torch.bmm(low, self.time_maa_w2).view(5, B, T, C)
            xw, xk, xv, xr, xg = mix_op.ddlerp(x, self._maa5(), corr, first, rev_n).unbind(0)
        else:
            assert rev_n is None, "the reversed-stream shift exists in the fused (HIP) path only"
            prev = F.pad(x, (0, 0, 1, -1)) if shifted is None else shifted
            delta = prev - x
            lead = torch.addcmul(x, delta, self.time_maa_x)
            low = torch.tanh(lead @ self.time_maa_w1).view(B * T, 5, -1).transpose(0, 1)
            corr = torch.bmm(low, self.time_maa_w2).view(5, B, T, C)
            xw, xk, xv, xr, xg = (x + delta * (self._maa5().view(5, 1, 1, C) + corr)).unbind(0)
        decay = self.time_decay + torch.tanh(xw @ self.time_decay_w1) @ self.time_decay_w2
        return self.receptance(xr), self.key(xk), self.value(xv), F.silu(self.gate(xg)), decay

    def jit_func_2(self, x, g):
        """per-head GroupNorm, gate, output projection (src/model.py:462-468); on bf16 GPU tensors the GroupNorm and the
        gate multiply are one HIP kernel (mix_op.group_norm_gate)."""
        B, T, C = x.size()
        if self._use_fused(x):
            from . import mix_op
            gated = mix_op.group_norm_gate(x.reshape(B * T, C), g.reshape(B * T, C), self.ln_x.weight, self.ln_x.bias,
                                           self.n_head, self.ln_x.eps).view(B, T, C)
            return self.output(gated)
        return self.output(self.ln_x(x.view(B * T, C)).view(B, T, C) * g)

    def _run(self, r, k, v, w):
        B, T, C = r.shape
        return self.wkv(B, T, C, self.n_head, r, k, v, w, self.time_faaaa)

    def forward(self, x):
        """causal time-mix (src/model.py:470-477).  With the HIP operator on bf16 GPU tensors the operator, the per-head
        GroupNorm and the gate multiply are ONE kernel (wkv.WKV_6_GN): y never leaves the chip on its way to
        the normalisation."""
        r, k, v, g, w = self.jit_func(x)
        if self.wkv is _default_wkv and self._use_fused(x) and self.fuse_epilogue:
            from .wkv import RUN_CUDA_RWKV6_GN
            B, T, C = r.shape
            gated = RUN_CUDA_RWKV6_GN(B, T, C, self.n_head, *(t.contiguous() for t in (r, k, v, w)), self.time_faaaa, g,
                                      self.ln_x.weight, self.ln_x.bias, self.ln_x.eps)
            return self.output(gated)
        return self.jit_func_2(self._run(r, k, v, w), g)

    def _rev_wkv(self, r, k, v, w, rev_n, rev_mask):
        from .wkv import WKV_6_REV
        B, T, C = r.shape
        bf = torch.bfloat16
        return WKV_6_REV.apply(B, T, C, self.n_head, *(t.to(bf).contiguous() for t in (r, k, v, w, self.time_faaaa)),
                               rev_n, rev_mask).to(r.dtype)

    def _pair_wkv(self, fwd, rev, rev_n, rev_mask):
        """Both operator calls of a bidirectional composition in one launch per pass (wkv.WKV_6_PAIR, row n2)."""
        from .wkv import WKV_6_PAIR
        B, T, C = fwd[0].shape
        bf = torch.bfloat16
        y, ry = WKV_6_PAIR.apply(B, T, C, self.n_head, *(t.to(bf).contiguous() for t in (*fwd, *rev)),
                                 self.time_faaaa.to(bf).contiguous(), rev_n, rev_mask)
        return y.to(fwd[0].dtype), ry.to(fwd[0].dtype)

    def _in_kernel_reversal(self, x):
        """Rows n2: with the HIP operator on bf16 GPU tensors the reversed half of the bidirectional compositions is
        addressed inside the kernels (wkv6_*_rev_ex, ddlerp rev_n) instead of through torch.gather round trips."""
        return self.wkv is _default_wkv and self._use_fused(x)

    def forward_bi_c(self, x, rev_idx, mask=None):
        """composition C (src/model_ext.py:421-437): reverse the hidden states, project twice, average."""
        r, k, v, g, w = self.jit_func(x)
        if mask is not None and self._in_kernel_reversal(x):
            from .wkv6_op import REV_ALL
            rev_n = mask.sum(dim=1).to(torch.int32)
            rr, rk, rv, _, rw = self.jit_func(x, rev_n=rev_n)      # the reversed stream's projections, in original order
            if self.pair_launch:
                y, ry = self._pair_wkv((r, k, v, w), (rr, rk, rv, rw), rev_n, REV_ALL)
            else:
                y, ry = self._run(r, k, v, w), self._rev_wkv(rr, rk, rv, rw, rev_n, REV_ALL)
        else:
            y = self._run(r, k, v, w)
            rr, rk, rv, _, rw = self.jit_func(reverse_x(x, rev_idx))
            ry = reverse_x(self._run(rr, rk, rv, rw), rev_idx)
        return self.jit_func_2((y + ry) / 2, g)

    def forward_bi_b(self, x, mask=None):
        """composition B (src/model_bi.py:325-350): only k and v are reversed, outputs are added."""
        B, T, C = x.size()
        if mask is None:
            mask = torch.ones(B, T, device=x.device)
        r, k, v, g, w = self.jit_func(x)
        if self._in_kernel_reversal(x):
            from .wkv6_op import REV_K, REV_V, REV_Y
            rev_n = mask.sum(dim=1).to(torch.int32)
            if self.pair_launch:
                y, ry = self._pair_wkv((r, k, v, w), (r, k, v, w), rev_n, REV_K | REV_V | REV_Y)
            else:
                y, ry = self._run(r, k, v, w), self._rev_wkv(r, k, v, w, rev_n, REV_K | REV_V | REV_Y)
            return self.jit_func_2(y + ry, g)
        y = self._run(r, k, v, w)
        rev_idx = reverse_x_idx(mask, T)
        ry = self._run(r, reverse_x(k, rev_idx), reverse_x(v, rev_idx), w)
        return self.jit_func_2(y + reverse_x(ry, rev_idx), g)


class CMix_x060(nn.Module):
    """RWKV-6 channel-mix FFN (src/model.py:616-644): squared-ReLU key, sigmoid receptance gate.  On bf16 GPU tensors the
    elementwise glue between the three GEMMs runs as HIP kernels (mix_op: token shift + both lerps in one pass, relu^2, and
    sigmoid * value in one pass each) instead of ten eager kernels; `fused` = None picks by tensor type, True / False force."""

    def __init__(self, n_embd, dim_ffn, fused=None):
        super().__init__()
        self.fused = fused
        self.time_maa_k = nn.Parameter(torch.zeros(1, 1, n_embd))
        self.time_maa_r = nn.Parameter(torch.zeros(1, 1, n_embd))
        self.key = nn.Linear(n_embd, dim_ffn, bias=False)
        self.receptance = nn.Linear(n_embd, n_embd, bias=False)
        self.value = nn.Linear(dim_ffn, n_embd, bias=False)

    def _use_fused(self, x):
        if self.fused is None:
            from . import mix_op
            C = x.shape[-1]
            return (mix_op.fusable(x, self.time_maa_k, self.time_maa_r) and x.dim() == 3 and C % 64 == 0 and C <= 4096
                    and self.key.weight.dtype == torch.bfloat16)
        return self.fused

    def forward(self, x):
        if self._use_fused(x):
            from . import mix_op
            xk, xr = mix_op.ddlerp(x, torch.cat([self.time_maa_k, self.time_maa_r], 0).view(2, -1))
            k = mix_op.sqrelu(self.key(xk))
            return mix_op.sigmoid_mul(self.receptance(xr), self.value(k))
        xx = F.pad(x, (0, 0, 1, -1)) - x
        k = torch.relu(self.key(x + xx * self.time_maa_k)) ** 2
        return torch.sigmoid(self.receptance(x + xx * self.time_maa_r)) * self.value(k)


class BiBlock(nn.Module):
    """src/model_encoder_run.py:222-259 (pre-LN residual block, ln0 on the first layer)."""

    def __init__(self, n_embd, dim_att, dim_ffn, layer_id, wkv=None):
        super().__init__()
        self.layer_id = layer_id
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        if layer_id == 0:
            self.ln0 = nn.LayerNorm(n_embd)
        self.att = Tmix_x060(n_embd, dim_att, wkv=wkv)
        self.ffn = CMix_x060(n_embd, dim_ffn)

    def forward(self, x, rev_idx, mask):
        if self.layer_id == 0:
            x = self.ln0(x)
        x = x + self.att.forward_bi_c(self.ln1(x), rev_idx, mask)
        return x + self.ffn(self.ln2(x))


class RwkvEncoder(nn.Module):
    """Bidirectional RWKV-6 encoder (src/model_encoder_run.py:262-348, share_emb, no head_qk, no dropout)."""

    def __init__(self, vocab_size, n_embd, n_layer, dim_att=None, dim_ffn=None, emb_id=1, pad_id=0, wkv=None):
        super().__init__()
        self.emb_id, self.pad_id = emb_id, pad_id
        self.emb = nn.Embedding(vocab_size, n_embd)
        self.blocks = nn.ModuleList([BiBlock(n_embd, dim_att or n_embd, dim_ffn or 4 * n_embd, i, wkv=wkv)
                                     for i in range(n_layer)])
        self.ln_out = nn.LayerNorm(n_embd)

    def forward(self, idx, return_logits=False):
        B, T = idx.size()
        mask = create_mask(idx, emb_id=self.emb_id, pad_id=self.pad_id)
        rev_idx = reverse_x_idx(mask, T)
        x = self.emb(idx)
        for block in self.blocks:
            x = block(x, rev_idx, mask)
        hidden = self.ln_out(x)
        logits = torch.matmul(hidden, self.emb.weight.t())
        return (logits, hidden) if return_logits else logits

    def encode_sentence(self, idx):
        _, hidden = self.forward(idx, True)
        position = torch.eq(idx, self.emb_id).int().argmax(-1)
        return hidden[torch.arange(hidden.size(0)), position]


# ---- a17: embedding head ---------------------------------------------------------------------------------------
def pooling(x, actual_len, pooling_type="weightedmean"):
    """src/model_ext.py:1708-1738.  actual_len[b] = index of the first emb_id token of row b."""
    T = x.size(1)
    if pooling_type == "weightedmean":
        mask = torch.arange(T, device=x.device) <= actual_len.unsqueeze(1)
        weights = torch.arange(1, T + 1, device=x.device).unsqueeze(0).float() / actual_len.unsqueeze(1).float()
        weights = weights * mask.float()
        x = torch.sum(x * weights.unsqueeze(-1), dim=1) / actual_len.unsqueeze(1).float()
        return x.bfloat16()
    if pooling_type == "lasttoken":
        return x[torch.arange(x.size(0)), actual_len]
    if pooling_type == "avg":
        mask = (torch.arange(T, device=x.device).unsqueeze(0) < actual_len.unsqueeze(1)).to(x.dtype)
        return (torch.sum(x * mask.unsqueeze(-1), dim=1) / actual_len.unsqueeze(1).float()).bfloat16()
    raise ValueError(pooling_type)


def cos_sim(a, b):
    """sentence_transformers.util.cos_sim: all pairs, [len(a), len(b)]."""
    return F.normalize(a, p=2, dim=1) @ F.normalize(b, p=2, dim=1).t()


def pairwise_cos_sim(a, b):
    """sentence_transformers.util.pairwise_cos_sim: row i of a with row i of b."""
    return (F.normalize(a, p=2, dim=1) * F.normalize(b, p=2, dim=1)).sum(-1)


def info_nce_loss(query, positive, negative=None, scale=20.0):
    """In-batch-negative loss of RwkvForSequenceEmbedding.training_step (src/model_ext.py:1897-1911):
    CE([cos_sim(q, p) * 20 | pairwise_cos_sim(q, n) * 20], arange(bs)); negatives are per rank only."""
    scores = cos_sim(query, positive) * scale
    if negative is not None:
        scores = torch.cat([scores, pairwise_cos_sim(query, negative).unsqueeze(1) * scale], dim=1)
    labels = torch.arange(scores.shape[0], dtype=torch.long, device=scores.device)
    return F.cross_entropy(scores, labels)
