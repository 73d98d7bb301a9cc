"""Child process of tests/test_train_dp_gpu.py::test_ddp_over_rccl_at_world_size_one (never imported by the product).

Started as a fresh interpreter that has not touched the GPU: initialises torch.distributed with the `nccl` backend (= RCCL on
ROCm) at world size 1, wraps the 2-layer LoRA SequenceEmbedder with train_dp.wrap_ddp (gradient_as_bucket_view, non-reentrant
activation checkpointing through the custom autograd Functions of the HIP operator) and runs ONE train_steps step; the LoRA / dense
gradients must equal the un-wrapped step's bit for bit (world size 1: the all-reduce is the identity and the mean divides by 1).
Prints one JSON line; exit code 0 only if every gradient is identical.  peft_train/peft_train_bi_encoder.py:290-311 is the
reference topology this stands in for (one process per GPU, gradient all-reduce through the trainer's DDP / ZeRO wrapper)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from rwkv_lm_ext_amd import callers, train_dp
    from rwkv_lm_ext_amd.dp import BucketBatchSampler
    from test_train_dp_cpu import BS, T, VOCAB, _model

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def fresh():
        m = _model()                                                  # seeded: identical weights every time
        m.grad_cp = True                                              # non-reentrant activation checkpointing, as the benched step
        for blk in m.blocks:
            blk.att.wkv = callers._default_wkv                        # the HIP operator
        return m.to(dev, torch.bfloat16).train()

    batch = next(iter(train_dp.batches(BucketBatchSampler([4 * BS], [BS], 0, 1), T, VOCAB)))

    plain = fresh()
    b = {k: v.to(dev) for k, v in batch.items()}
    loss0 = train_dp.training_loss(plain, b["query"], b["positive"], b["negative"])
    loss0.backward()
    g0 = {n: p.grad.detach().clone() for n, p in plain.named_parameters() if p.requires_grad}

    model = fresh()
    net = train_dp.wrap_ddp(model, dev)
    opt = torch.optim.SGD(train_dp.trainable_parameters(model), lr=0.0)   # lr 0: the step runs, the weights (hence a second look) stay
    losses = train_dp.train_steps(net, opt, iter([batch]), dev, 1)
    torch.cuda.synchronize()
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    # one all-reduce through RCCL by hand as well: a tensor comes back unchanged at world size 1
    t = torch.arange(1024, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    diffs = {n: float((g0[n].float() - g1[n].float()).abs().max()) for n in g0}
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "n_grads": len(g0), "names_equal": sorted(g0) == sorted(g1),
           "loss_plain": float(loss0), "loss_ddp": float(losses[0]), "max_abs_diff": max(diffs.values()),
           "nonzero": sum(float(g.abs().max()) > 0 for g in g0.values()),
           "allreduce_identity": bool(torch.equal(t, torch.arange(1024, device=dev, dtype=torch.float32))),
           "bitwise_equal": all(torch.equal(g0[n], g1[n]) for n in g0)}
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    ok = out["bitwise_equal"] and out["names_equal"] and out["allreduce_identity"] and out["nonzero"] == out["n_grads"] and out["loss_plain"] == out["loss_ddp"]
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
