"""BASELINE configs[3] on the CPU: the LoRA bi-encoder training step under torch DDP with world_size-2 gloo processes.

The reference's loss uses per-rank in-batch negatives (src/model_ext.py:1899-1909, no cross-rank gather), so what the
gradient all-reduce must reproduce is the gradient of mean_r(loss_r) with loss_r computed on rank r's own slice of the
global batch (data/custom_datasets.py:51-55): a single process computes exactly that on the concatenated batch, shard by
shard, and the all-reduced LoRA gradients of both ranks must equal it.  The WKV operator is replaced by the pure-PyTorch
port of the reference's CPU recurrence (oracle/, test infrastructure) -- the HIP path needs a GPU.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rwkv_lm_ext_amd import train_dp
from rwkv_lm_ext_amd.dp import BucketBatchSampler

VOCAB, N_EMBD, N_LAYER, DIM_FFN, T, BS = 64, 128, 2, 256, 24, 3


def _naive_wkv(B, T_, C, H, r, k, v, w, u):
    from oracle.wkv6_torch_naive import wkv6_naive
    return wkv6_naive(r, k, v, w, u)


def _model():
    torch.manual_seed(1234)
    m = train_dp.SequenceEmbedder(VOCAB, N_EMBD, N_LAYER, dim_ffn=DIM_FFN, add_mlp=True, output_dim=32, wkv=_naive_wkv)
    with torch.no_grad():                       # non-trivial time-mix parameters (the module zero-initialises them)
        for n, p in m.named_parameters():
            if "time_" in n or "ln_x" in n:
                p.copy_(torch.randn_like(p) * 0.1)
            if "time_decay" in n and p.dim() == 3:
                p.sub_(3.0)
    replaced = train_dp.inject_lora(m, r=4, alpha=16)
    assert len(replaced) == 3 * N_LAYER and all(".ffn." in n for n in replaced)
    for p in m.dense.parameters():              # --add_mlp: the dense head is trained too (peft_train_bi_encoder.py:105)
        p.requires_grad_(True)
    with torch.no_grad():                       # lora_B = 0 would make every lora_A gradient vanish
        for n, p in m.named_parameters():
            if n.endswith("lora_B"):
                p.copy_(torch.randn_like(p) * 0.05)
    return m


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model()
    ddp = train_dp.wrap_ddp(model)
    sampler = BucketBatchSampler([4 * BS * world], [BS], rank, world)
    batch = next(iter(train_dp.batches(sampler, T, VOCAB)))
    loss = train_dp.training_loss(ddp, batch["query"], batch["positive"], batch["negative"])
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
    q.put((rank, float(loss), {n: g.numpy() for n, g in grads.items()}, train_dp.grad_allreduce_bytes(model)))
    dist.destroy_process_group()


def test_lora_injection_and_message_size():
    m = _model()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert all(("lora_" in n) or n.startswith("dense.") for n in names)
    per_layer = 4 * (N_EMBD + DIM_FFN) * 2 + 4 * 2 * N_EMBD          # r * (in + out) for key, value + receptance
    assert train_dp.grad_allreduce_bytes(m) == 4 * (N_LAYER * per_layer + N_EMBD * 32 + 32)
    # the 1B6 configuration of SURVEY.md 8e: 24 layers, C=2048, dim_ffn=7168, r=8 -> 180 224 parameters per layer
    assert 8 * (2048 + 7168) * 2 + 8 * 4096 == 180224
    # a frozen base weight receives no gradient, LoRA factors do
    lin = train_dp.LoraLinear(8, 6, r=2, alpha=4)
    with torch.no_grad():
        lin.lora_B.normal_()
    lin(torch.randn(3, 8)).sum().backward()
    assert lin.weight.grad is None and lin.lora_A.grad is not None and lin.lora_B.grad is not None
    x = torch.randn(3, 8)
    want = x @ lin.weight.t() + 2.0 * (x @ lin.lora_A.t()) @ lin.lora_B.t()
    assert torch.allclose(lin(x), want, atol=1e-6)


def test_two_rank_gloo_lora_gradients_match_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process: the same global batch, shard by shard (per-rank negatives), mean of the two losses
    model = _model()
    losses = []
    for r in range(world):
        batch = next(iter(train_dp.batches(BucketBatchSampler([4 * BS * world], [BS], r, world), T, VOCAB)))
        losses.append(train_dp.training_loss(model, batch["query"], batch["positive"], batch["negative"]))
    (sum(losses) / world).backward()
    ref = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert len(ref) == 4 * 3 * 0 + len(res[0][2]) and len(ref) >= 2 * 3 * N_LAYER
    for rank, loss, grads, nbytes in res:
        assert abs(loss - float(losses[rank])) <= 1e-5 * max(1.0, abs(loss))
        assert nbytes == train_dp.grad_allreduce_bytes(model)
        for n, g in grads.items():
            want = ref[n]
            err = float((torch.from_numpy(g) - want).abs().max() / want.abs().max().clamp_min(1e-12))
            assert err <= 1e-4, (rank, n, err)
            assert float(want.abs().max()) > 0, n               # every trainable parameter really gets a gradient
    # both ranks hold the identical reduced gradient
    for n in res[0][2]:
        assert (res[0][2][n] == res[1][2][n]).all(), n
