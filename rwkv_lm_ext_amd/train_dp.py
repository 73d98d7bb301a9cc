"""Data-parallel LoRA training step of the sentence-embedding model around the WKV6 operator (BASELINE configs[3]).

What the reference runs (peft_train/peft_train_bi_encoder.py): RWKV-x060 backbone, LoRA r=8 alpha=32 injected into
`ffn.key / ffn.value / ffn.receptance` (:152-154, :252-264), `RwkvForSequenceEmbedding` on top
(src/model_ext.py:1690-1769: backbone -> ln_out -> pooling at the first `emb_id` -> optional dense+tanh), loss =
in-batch-negative InfoNCE over cat[query, positive, negative] (src/model_ext.py:1882-1911; negatives are per rank
only), Lightning + DeepSpeed ZeRO-2 for the gradient reduction over NCCL, and `MyBatchSampler` dealing rank-strided
slices of each length bucket (data/custom_datasets.py:19-74).

Here: the same modules restated on `callers.py`, `torch.nn.parallel.DistributedDataParallel` over the `nccl` backend
(= RCCL over xGMI on ROCm) for the one collective of the path -- the all-reduce of the trainable (LoRA / dense)
gradients, 13-28 MB per step for the 1B6 model, one or two buckets overlapped with the backward -- and
`dp.BucketBatchSampler` for the dealing.  Nothing else of the reference's trainer is rebuilt.
"""
import math
from typing import Iterable, List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import callers
from .dp import BucketBatchSampler


class LoraLinear(nn.Module):
    """src/rwkvLinear.py:40-97: y = x W^T + (alpha / r) * dropout(x) A^T B^T with W frozen, B zero-initialised
    (peft's LoraLayer has the same forward; the reference injects it through peft, :252-264)."""

    def __init__(self, in_features: int, out_features: int, r: int = 8, alpha: float = 32.0, dropout: float = 0.0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features), requires_grad=False)
        self.lora_A = nn.Parameter(torch.empty(r, in_features))
        self.lora_B = nn.Parameter(torch.zeros(out_features, r))
        self.lora_dropout = nn.Dropout(dropout)
        self.scaling = alpha / r
        self.r = r
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))

    @classmethod
    def from_linear(cls, lin: nn.Linear, r: int, alpha: float, dropout: float) -> "LoraLinear":
        assert lin.bias is None, "Biased LoraLinear not supported"          # src/rwkvLinear.py:46
        m = cls(lin.in_features, lin.out_features, r, alpha, dropout).to(lin.weight.device, lin.weight.dtype)
        with torch.no_grad():
            m.weight.copy_(lin.weight)
        return m

    def forward(self, x):
        # fused path: bf16 everywhere, no dropout, and a FROZEN base weight (the reference's trainer freezes it; a caller that trains
        # it gets the eager path, whose autograd produces the weight gradient the fused backward does not)
        if (x.is_cuda and self.lora_dropout.p == 0.0 and not self.weight.requires_grad
                and x.dtype == self.weight.dtype == self.lora_A.dtype == self.lora_B.dtype == torch.bfloat16):
            return _LoraLinearFn.apply(x, self.weight, self.lora_A, self.lora_B, self.scaling)
        return F.linear(x, self.weight) + self.scaling * F.linear(F.linear(self.lora_dropout(x), self.lora_A), self.lora_B)


class _LoraLinearFn(torch.autograd.Function):
    """y = x W^T + s (x A^T) B^T with the low-rank term added in the second GEMM's epilogue (addmm_, beta = 1: one rounding of
    the sum) instead of a separate elementwise add over [rows, out] -- at the 1B6 shape those adds were 10.7 % of the training
    step's kernel time (profiles/r03_dp_lora_kernel_shares.txt).  Backward likewise: gx = gy W, then += s (gy B) A in a GEMM
    epilogue; gA, gB from the rank-r intermediates; nothing of size [rows, out] is kept for it besides x (as F.linear keeps)."""

    @staticmethod
    def forward(ctx, x, weight, lora_A, lora_B, scaling):
        x2 = x.reshape(-1, x.shape[-1])
        xa = x2 @ lora_A.t()                                       # [rows, r]
        y = x2 @ weight.t()
        y.addmm_(xa, lora_B.t(), alpha=scaling)
        ctx.save_for_backward(x2, weight, lora_A, lora_B, xa)
        ctx.scaling, ctx.shape = scaling, x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight, lora_A, lora_B, xa = ctx.saved_tensors
        s = ctx.scaling
        gy2 = gy.reshape(-1, gy.shape[-1])
        gyb = gy2 @ lora_B                                          # [rows, r]
        gx = gA = gB = None
        if ctx.needs_input_grad[0]:
            gx = gy2 @ weight
            gx.addmm_(gyb, lora_A, alpha=s)
            gx = gx.view(ctx.shape)
        if ctx.needs_input_grad[2]:
            gA = (gyb.t() @ x2) * s
        if ctx.needs_input_grad[3]:
            gB = (gy2.t() @ xa) * s
        return gx, None, gA, gB, None


def inject_lora(model: nn.Module, targets: Sequence[str] = ("ffn.key", "ffn.value", "ffn.receptance"), r: int = 8,
                alpha: float = 32.0, dropout: float = 0.0) -> List[str]:
    """Freeze `model`, replace every nn.Linear whose qualified name ends with one of `targets` by a LoraLinear
    (peft.inject_adapter_in_model with LoraConfig(target_modules=...), peft_train_bi_encoder.py:252-264).
    Returns the names of the replaced modules."""
    for p in model.parameters():
        p.requires_grad_(False)
    replaced = []
    for name, mod in list(model.named_modules()):
        for child_name, child in list(mod.named_children()):
            full = f"{name}.{child_name}" if name else child_name
            if isinstance(child, nn.Linear) and any(full.endswith(t) for t in targets):
                setattr(mod, child_name, LoraLinear.from_linear(child, r, alpha, dropout))
                replaced.append(full)
    return replaced


class Block(nn.Module):
    """Causal RWKV-6 block (src/model.py:904-933 without dropout / tiny-att): x + att(ln1 x), x + ffn(ln2 x)."""

    def __init__(self, n_embd, dim_att, dim_ffn, layer_id, wkv=None):
        super().__init__()
        self.layer_id = layer_id
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        if layer_id == 0:
            self.ln0 = nn.LayerNorm(n_embd)
        self.att = callers.Tmix_x060(n_embd, dim_att, wkv=wkv)
        self.ffn = callers.CMix_x060(n_embd, dim_ffn)

    def forward(self, x):
        if self.layer_id == 0:
            x = self.ln0(x)
        x = x + self.att(self.ln1(x))
        return x + self.ffn(self.ln2(x))


class SequenceEmbedder(nn.Module):
    """RwkvForSequenceEmbedding.forward (src/model_ext.py:1739-1769): emb -> blocks -> ln_out -> pooling at the
    first `emb_id` token -> optional tanh(dense(.))."""

    def __init__(self, vocab_size, n_embd, n_layer, dim_att=None, dim_ffn=None, emb_id=1, pad_id=0,
                 pooling_type="weightedmean", add_mlp=False, output_dim=0, grad_cp=False, wkv=None):
        super().__init__()
        self.emb_id, self.pad_id, self.pooling_type, self.grad_cp = emb_id, pad_id, pooling_type, grad_cp
        self.emb = nn.Embedding(vocab_size, n_embd)
        self.blocks = nn.ModuleList([Block(n_embd, dim_att or n_embd, dim_ffn or int(n_embd * 3.5) // 32 * 32, i, wkv=wkv)
                                     for i in range(n_layer)])
        self.ln_out = nn.LayerNorm(n_embd)
        self.dense = nn.Linear(n_embd, output_dim or n_embd) if add_mlp else None

    def forward(self, idx):
        x = self.emb(idx)
        for block in self.blocks:
            if self.grad_cp and self.training and torch.is_grad_enabled():   # args.grad_cp, src/model_ext.py:1756-1758
                from torch.utils.checkpoint import checkpoint
                x = checkpoint(block, x, use_reentrant=False)
            else:
                x = block(x)
        x = self.ln_out(x)
        actual_len = torch.eq(idx, self.emb_id).int().argmax(-1)
        x = callers.pooling(x, actual_len, self.pooling_type)
        if self.dense is not None:
            x = torch.tanh(self.dense(x.to(self.dense.weight.dtype)))   # pooling returns bf16 (src/model_ext.py:1721)
        return x


def training_loss(model: nn.Module, query, positive, negative=None):
    """RwkvForSequenceEmbedding.training_step with is_in_batch_negative (src/model_ext.py:1882-1911)."""
    parts = [query, positive] + ([negative] if negative is not None else [])
    emb = model(torch.cat(parts, dim=0)).float()
    bs = query.size(0)
    return callers.info_nce_loss(emb[:bs], emb[bs:2 * bs], emb[2 * bs:] if negative is not None else None)


def trainable_parameters(model: nn.Module) -> List[nn.Parameter]:
    return [p for p in model.parameters() if p.requires_grad]


def grad_allreduce_bytes(model: nn.Module) -> int:
    """Bytes one gradient all-reduce moves per rank and step (the message of SURVEY.md 8e)."""
    return sum(p.numel() * p.element_size() for p in trainable_parameters(model))


def wrap_ddp(model: nn.Module, device=None, bucket_cap_mb: int = 32):
    """DDP over the default process group (nccl = RCCL on ROCm, gloo in the CPU tests).  One ring all-reduce per
    bucket; the whole LoRA gradient set of the 1B6 model (13-28 MB) fits one 32 MB bucket, so the collective is
    per-link bound on xGMI only once per step and overlaps the tail of the backward."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = None if device is None or device.type == "cpu" else [device.index]
    return DDP(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
               find_unused_parameters=False, broadcast_buffers=False)


# ---- synthetic data of the reference's shape: length-bucketed (query, positive, negative) triples --------------------
def synth_example(index: int, seq_len: int, vocab_size: int, emb_id: int = 1, pad_id: int = 0, min_len: int = 16):
    """Deterministic pseudo-random triple for dataset position `index`: tokens, then `emb_id`, then padding
    (the layout create_mask / pooling expect, src/model_ext.py:1763-1764)."""
    g = torch.Generator().manual_seed(1000003 * index + 17)
    out = []
    for _ in range(3):
        n = int(torch.randint(min_len, seq_len - 1, (1,), generator=g))
        row = torch.full((seq_len,), pad_id, dtype=torch.long)
        row[:n] = torch.randint(2, vocab_size, (n,), generator=g)
        row[n] = emb_id
        out.append(row)
    return out


def batches(sampler: BucketBatchSampler, seq_len: int, vocab_size: int) -> Iterable[dict]:
    for idxs in sampler:
        rows = [synth_example(i, seq_len, vocab_size) for i in idxs]
        yield {"query": torch.stack([r[0] for r in rows]), "positive": torch.stack([r[1] for r in rows]),
               "negative": torch.stack([r[2] for r in rows])}


def train_steps(model, opt, batch_iter, device, steps: int):
    """`steps` optimizer steps; returns the losses.  `model` is the DDP-wrapped SequenceEmbedder."""
    losses = []
    for _, batch in zip(range(steps), batch_iter):
        batch = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
        opt.zero_grad(set_to_none=True)
        loss = training_loss(model, batch["query"], batch["positive"], batch.get("negative"))
        loss.backward()              # DDP all-reduces the LoRA gradients here
        opt.step()
        losses.append(loss.detach())
    return losses
