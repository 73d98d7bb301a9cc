#!/usr/bin/env python3
"""WKV6 fwd+bwd micro-benchmark on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload wkv6|infctx|bi|dp_lora|prefill] [--no-cpu]

One "step" = one forward + one backward of the WKV6 operator through the C ABI of librwkv6_amd.so on one
batch of synthetic bf16 inputs that already live in HBM (BASELINE.json configs[1]: B=8, T=4096, C=2048,
H=32).  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs the same
per-GPU workload -- the path shards over the batch with no data-path collective -- and the
reported value is the whole-job rate: N * tokens / max-over-ranks time.  Rank 0 prints ONE JSON line.

Algorithmic bytes: forward reads r,k,v,w and writes y = 10 B per token-channel, backward
reads r,k,v,w,gy and writes gr,gk,gv,gw = 18 B; 28 B per token-channel for the step.  The forward also writes, and
the backward reads, one fp32 64x64 state checkpoint per 64 tokens (4 B per token-channel each way) -- real traffic
(reported in roofline.traffic) that is NOT counted in the algorithmic figure.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FWD_BYTES, BWD_BYTES = 10, 18   # per token-channel
GROUP_TOKENS, STAGE_TOKENS = 64, 32     # the forward kernel's loop unit (one workgroup barrier per 64-token group), the backward's (32-token stage)


PMC_FILE = os.path.join(ROOT, "profiles", "r06_final_pmc.json")


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a five-line version banner when its
    first communicator is made): file descriptor 1 is pointed at stderr for the whole run and the line goes out through a saved
    duplicate of the original descriptor (emit)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def pmc_from_file():
    """Fallback for roofline.traffic / valu_issue_share: the committed rocprofv3 PMC passes of this round's kernels."""
    try:
        with open(PMC_FILE) as f:
            return json.load(f)["kernels"]
    except Exception:
        return {}


def add_counter_rows(rows, acc):
    """Fold the rows of one rocprofv3 counter_collection.csv into acc[kernel][counter] = [value per dispatch]: a counter of one
    dispatch may come as several rows (one per dimension instance), which are summed."""
    rows = list(rows)                   # (a csv.DictReader is consumed by the first pass)
    per_dispatch = {}
    for row in rows:
        kn = row["Kernel_Name"]
        key = "chunk_bwd12k_kernel" if "chunk_bwd12k" in kn else "chunk_fwd_kernel" if "chunk_fwd_kernel" in kn else None
        if key:
            id_ = (key, row["Counter_Name"], row["Dispatch_Id"])
            per_dispatch[id_] = per_dispatch.get(id_, 0.0) + float(row["Counter_Value"])
    for (key, cname, _), val in per_dispatch.items():
        acc.setdefault(key, {}).setdefault(cname, []).append(val)
    # whole passes (workloads of several launches per pass: wkv6_bi, infctx, two-level scans): every dispatch of this library's kernels
    # between the phase markers the child launches (pmc_child: marker, forwards, marker, backwards, marker), summed per phase.  A
    # backward's own state passes and scan helpers carry forward-style names: the phase, not the name, says whose traffic they are.
    marks = sorted(int(row["Dispatch_Id"]) for row in rows if "pass_marker_kernel" in row["Kernel_Name"] and row["Counter_Name"] == rows[0]["Counter_Name"])
    if len(marks) >= 3:
        for row in rows:
            kn, d = row["Kernel_Name"], int(row["Dispatch_Id"])
            if "wkv6::" not in kn or "pass_marker_kernel" in kn or d < marks[0] or d > marks[2]:
                continue                        # (anything in front of the first marker -- module load, self-tests -- belongs to no pass)
            side = "_pass_fwd" if d < marks[1] else "_pass_bwd"
            tot = acc.setdefault(side, {}).setdefault(row["Counter_Name"], [0.0])
            tot[0] += float(row["Counter_Value"])


def reduce_counters(acc):
    """Per-launch averages and the HBM bytes they imply: FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE counts half of a coalesced
    stream's bytes on gfx950 (MI355X_MICROARCH.md)."""
    out = {}
    for key, d in acc.items():
        if key.startswith("_pass_"):        # totals over the child's PMC_CHILD_STEPS passes -> per pass
            avg = {c: v[0] / PMC_CHILD_STEPS for c, v in d.items()}
        else:
            avg = {c: sum(v) / len(v) for c, v in d.items()}
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            out[key] = {"counters": avg, "hbm_bytes": int(2 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024)}
    return out


def pmc_live(workload="wkv6", timeout_s=75):
    """HBM bytes and VALU utilisation of the two kernels, measured NOW: rank 0 at N = 1 runs this same script as a child under
    `rocprofv3 --kernel-trace --pmc <group>` (one run per counter group -- FETCH_SIZE and WRITE_SIZE cannot share a pass -- with a
    handful of fwd+bwd launches each) before it touches the GPU itself, and averages the counters per launch.  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950 (KiB units).  Returns {} when rocprofv3 is not usable; the caller then
    falls back to the committed file and says so in `traffic_source`."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof) or under_profiler():
        return {}
    acc = {}
    tmp = tempfile.mkdtemp(prefix="wkv6_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RWKV_AMD_NO_SELFTEST="1")
    try:
        for i, grp in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])):
            out = os.path.join(tmp, f"g{i}")
            cmd = [prof, "--kernel-trace", "--pmc", *grp, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload]
            # own session: on a timeout the whole group goes (rocprofv3 AND the profiled python under it), so that no GPU holder
            # is left running beside the timed section
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = child.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                child.wait()
                return {}
            if rc != 0:
                return {}
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    add_counter_rows(csv.DictReader(fh), acc)
    except Exception:
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return reduce_counters(acc)


def under_profiler():
    """Is this process itself running under rocprofv3 / another preloaded tool?  Then no second profiler is started beneath it
    (the children would inherit the preload environment)."""
    # (rocprofv3 exports ROCP_TOOL_LIBRARIES and preloads librocprofiler-sdk-tool.so for its child; a ROCPROFILER_* / ROCPROF_* path
    # variable of the image or the user is not a profiler)
    return "ROCP_TOOL_LIBRARIES" in os.environ or "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", "")


PMC_CHILD_STEPS = 6


def pmc_child(workload):
    """The profiled child of pmc_live(): PMC_CHILD_STEPS forward + backward passes of the workload, nothing else."""
    from rwkv_lm_ext_amd import wkv6_op
    fwd, bwd = build_workload(workload, torch.device("cuda", 0))[:2]
    fwd()                                   # (checkpoints for the first backward; in front of the first marker: counted in no pass)
    wkv6_op.pass_marker()
    for _ in range(PMC_CHILD_STEPS):
        fwd()
    wkv6_op.pass_marker()
    for _ in range(PMC_CHILD_STEPS):
        bwd()
    wkv6_op.pass_marker()
    torch.cuda.synchronize()


def valu_issue_share_of(c):
    """Sum over a SIMD's resident waves of their VALU issue cycles / kernel cycles = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles),
    GRBM_GUI_ACTIVE being summed over the 8 XCDs.  NOT a pipe utilisation: the counter charges every VALU instruction 4 cycles
    (8 for transcendental / permlane) whatever the instruction's real issue cost and however many waves share the SIMD -- it reads
    1.75-1.9 for a full-rate stream at 2-4 waves per SIMD (profiles/r05_issue_floor.md).  Kept as an instruction-density figure."""
    try:
        return round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
    except Exception:
        return None


def profiled_clock_ghz(c, ms):
    """Effective shader clock of a PROFILED launch (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE / 8 XCDs / duration."""
    try:
        return round(c["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9, 3)
    except Exception:
        return None


def synth(B, T, H, device, seed=0):
    """SURVEY.md 8d synthetic inputs: r,k,v ~ N(0,1)*0.5, w = model decay-init ramp + N(0,0.1^2), u ~ N(0,0.3^2)."""
    C = H * 64
    g = torch.Generator(device=device).manual_seed(seed)
    bf = torch.bfloat16
    r, k, v = (torch.randn(B, T, C, device=device, generator=g).mul_(0.5).to(bf) for _ in range(3))
    ramp = torch.tensor([-6 + 5 * (n / (C - 1)) ** (0.7 + 1.3 * 0.5) for n in range(C)], device=device)
    w = (ramp.view(1, 1, C) + 0.1 * torch.randn(B, T, C, device=device, generator=g)).to(bf)
    u = (torch.randn(H, 64, device=device, generator=g) * 0.3).to(bf)
    gy = torch.randn(B, T, C, device=device, generator=g).to(bf)
    return r, k, v, w, u, gy


def cpu_baseline(budget_s=20.0):
    """The reference's pure-PyTorch CPU algorithm (fla naive recurrence + autograd), restated in
    oracle/wkv6_torch_naive.py, timed on the host cores at BASELINE config 1's shape."""
    from oracle.wkv6_torch_naive import wkv6_naive_fwd_bwd
    B, T, H = 2, 128, 32
    C = H * 64
    g = torch.Generator().manual_seed(0)
    r, k, v = (torch.randn(B, T, C, generator=g) * 0.5 for _ in range(3))
    w = -1.0 + 0.5 * torch.randn(B, T, C, generator=g)
    u = torch.randn(H, 64, generator=g) * 0.3
    gy = torch.randn(B, T, C, generator=g)
    wkv6_naive_fwd_bwd(r[:, :8], k[:, :8], v[:, :8], w[:, :8], u, gy[:, :8])      # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        wkv6_naive_fwd_bwd(r, k, v, w, u, gy)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 8:
            break
    return {"value": round(B * T * reps / el, 1), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{reps} x (fwd + autograd bwd), fp32, B={B} T={T} C={C} H={H} (BASELINE config 1 shape), "
                      f"{el:.1f} s on {os.cpu_count()} host cpus"}


def bench_dp_lora(args, rank, world, dev, dist):
    """BASELINE configs[3]: LoRA bi-encoder step, DP over the ranks, gradient all-reduce through DDP (RCCL)."""
    from rwkv_lm_ext_amd import train_dp
    from rwkv_lm_ext_amd.dp import BucketBatchSampler, timed_steps
    vocab, n_embd, dim_ffn, T, bs = 65536, 2048, 7168, 512, args.per_gpu_batch     # RWKV-x060-1B6
    torch.manual_seed(0)
    with torch.device(dev):
        model = train_dp.SequenceEmbedder(vocab, n_embd, args.layers, dim_ffn=dim_ffn, add_mlp=True, output_dim=1024,
                                          grad_cp=True).to(torch.bfloat16)
    with torch.no_grad():                           # random-init weights of the architecture (no checkpoints offline)
        for n, p in model.named_parameters():
            if p.dim() >= 2 and "emb" not in n:
                p.normal_(0.0, 0.02)
            if "time_decay" in n and p.dim() == 3:
                p.copy_(torch.linspace(-6, -1, p.numel(), device=dev).view_as(p))
            if "ln_x.weight" in n or (n.endswith(".weight") and ".ln" in n):
                p.fill_(1.0)
    train_dp.inject_lora(model, r=8, alpha=32)
    for p in model.dense.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("lora_B"):
                p.normal_(0.0, 0.01)
    msg_bytes = train_dp.grad_allreduce_bytes(model)
    if dist is None:
        # one GPU: the same DDP wrapper over a one-rank RCCL group (hooks, bucket views and the all-reduce all execute)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    net = train_dp.wrap_ddp(model, dev)
    opt = torch.optim.AdamW(train_dp.trainable_parameters(model), lr=1e-5)
    total = args.steps + args.warmup + 1
    sampler = BucketBatchSampler([total * bs * world], [bs], rank, world)
    # the rank's batches, dealt by the sampler, staged in pinned host memory ahead of the timed region
    it = iter([{k: v.pin_memory() for k, v in b.items()} for b in train_dp.batches(sampler, T, vocab)])
    model.train()

    def step():
        batch = {k: v.to(dev, non_blocking=True) for k, v in next(it).items()}
        opt.zero_grad(set_to_none=True)
        loss = train_dp.training_loss(net, batch["query"], batch["positive"], batch["negative"])
        loss.backward()                             # DDP all-reduces the LoRA / dense gradients here
        opt.step()

    step()                                          # allocator / module-load warm-up, outside the W warm-up steps
    torch.cuda.synchronize()
    elapsed = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dist, dev)
    if rank == 0:
        tokens = 3 * bs * T                         # per GPU and step
        emit(({
            "metric": "LoRA bi-encoder training tokens/sec (RWKV-x060-1B6 shape, T=512, DP, RCCL grad all-reduce)",
            "value": round(world * tokens * args.steps / elapsed, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": 1,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "peft_train_bi_encoder LoRA r=8 alpha=32 on ffn.key/value/receptance + dense head "
                                   "(BASELINE configs[3])", "layers": args.layers, "n_embd": n_embd, "dim_ffn": dim_ffn,
                       "seq_len": T, "per_gpu_batch": bs, "global_batch": bs * world,
                       "sequences_per_gpu_step": 3 * bs, "grad_checkpointing": True,
                       "parallelism": f"dp{world} (DDP, one bucketed all-reduce of the trainable gradients per step)",
                       "allreduce_bytes_per_step": msg_bytes}}))
    if dist is not None:
        dist.destroy_process_group()


PREWARM_MIN_S, PREWARM_MAX_S = 0.3, 2.0      # MI355X_MICROARCH.md, DVFS give-back item 6: sustained clocks need back-to-back launches
PREWARM_BATCH, PREWARM_WINDOW = 8, 4         # convergence: the last 4 batches of 8 iterations (32 iterations) within 1 % of each other
PREWARM_TOL = 0.01


def prewarm_until_steady(fwd, bwd):
    """Device pre-warm, outside both the W warm-up steps and the timed region: back-to-back fwd+bwd iterations until the mean iteration
    time of the last PREWARM_WINDOW batches agrees within PREWARM_TOL (max - min over mean), at least PREWARM_MIN_S of device time and
    at most PREWARM_MAX_S.  The queue never drains: the host waits for the end of batch n - 1 with batch n already enqueued, and on
    return the last batch is still in flight, so the caller's next launch follows it without an idle gap.  Nothing is cached by it:
    every iteration recomputes forward and backward from the same inputs."""
    marks = [torch.cuda.Event(enable_timing=True)]
    marks[0].record()
    n, per_batch, total_ms, converged = 0, [], 0.0, False
    while n < 100000:
        for _ in range(PREWARM_BATCH):
            fwd()
            bwd()
        n += 1
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[n].record()
        if n < 2:
            continue
        marks[n - 1].synchronize()                      # batch n - 1 is done, batch n keeps the device busy
        ms = marks[n - 2].elapsed_time(marks[n - 1])
        per_batch.append(ms / PREWARM_BATCH)
        total_ms += ms
        last = per_batch[-PREWARM_WINDOW:]
        spread = (max(last) - min(last)) / (sum(last) / len(last))
        converged = len(last) == PREWARM_WINDOW and spread < PREWARM_TOL
        if (converged and total_ms >= PREWARM_MIN_S * 1e3) or total_ms >= PREWARM_MAX_S * 1e3:
            break
    return {"prewarm_iters": n * PREWARM_BATCH, "prewarm_ms": round(total_ms, 1), "prewarm_converged": bool(converged),
            "prewarm_first_ms_per_iter": round(per_batch[0], 4) if per_batch else None,
            "prewarm_last_ms_per_iter": round(per_batch[-1], 4) if per_batch else None}


def timed_launch_clocks(clocks, side, steps, launches_per_pass, what="ghz"):
    """In-kernel GHz (what = "ghz": mean over the step's launches) or kernel-only duration in ms (what = "us": sum over them) of the timed
    steps' launches of one kernel: the last steps x launches_per_pass entries of the clock ring, one figure per step."""
    xs = [g for g in clocks.get(f"{side}_{what}_launches", [])][-steps * launches_per_pass:]
    if len(xs) < steps * launches_per_pass or any(g is None for g in xs):
        return []
    per = [sum(xs[i * launches_per_pass:(i + 1) * launches_per_pass]) for i in range(steps)]
    return [round(x / launches_per_pass, 3) for x in per] if what == "ghz" else [round(x * 1e-3, 4) for x in per]


def build_workload(workload, dev, seed=0):
    """(fwd, bwd, tokens, B, T, H, name) of one of the operator-level workloads: closures over synthetic inputs resident in HBM."""
    from rwkv_lm_ext_amd import wkv6_op
    if workload == "wkv6":
        B, T, H = 8, 4096, 32
        name = "wkv6_fwd_bwd B=8 T=4096 C=2048 H=32 (BASELINE configs[1])"
    elif workload == "prefill":
        B, T, H = 1, 16384, 32
        name = "WKV6 forward only (inference prefill) B=1 T=16384 C=2048, two-level scan over T; not a BASELINE config"
    elif workload == "infctx":
        B, T, H = 4, 16384, 32
        name = "wkv6infctx fwd+bwd B=4 T=16384 in 8 chunks of 2048, bf16 state carry (BASELINE configs[4])"
    else:
        B, T, H = 48, 512, 32
        name = "wkv6_bi fwd+bwd B=48 (16x3) T=512, mask lengths U[64,512] (BASELINE configs[2])"
    C = H * 64
    r, k, v, w, u, gy = synth(B, T, H, dev, seed=seed)
    tokens = B * T

    if workload == "wkv6":
        y = torch.empty_like(r)
        ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)     # forward-state checkpoints, as WKV_6.apply keeps them

        def fwd():
            wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)

        def bwd():
            wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
    elif workload == "prefill":
        y = torch.empty_like(r)

        def fwd():
            wkv6_op.forward_ex(r, k, v, w, u, H, y=y)

        def bwd():
            pass
    elif workload == "infctx":
        chunks = [slice(2048 * c, 2048 * (c + 1)) for c in range(8)]
        parts = [[x[:, sl].contiguous() for x in (r, k, v, w, gy)] for sl in chunks]
        states = [torch.zeros(B, H, 64, 64, device=dev, dtype=torch.bfloat16) for _ in range(9)]
        ckpts = [wkv6_op.new_checkpoint(B, 2048, C, H, dev) for _ in range(8)]   # as WKV_6STATE_INFCTX keeps them

        def fwd():
            for c, (rc, kc, vc, wc, _) in enumerate(parts):
                wkv6_op.forward_ex(rc, kc, vc, wc, u, H, s0=states[c], s_out=states[c + 1], ckpt=ckpts[c])

        def bwd():      # truncated BPTT as the reference trains it: each chunk's backward from its entry state
            for c, (rc, kc, vc, wc, gc) in enumerate(parts):
                wkv6_op.backward_ex(rc, kc, vc, wc, u, gc, H, s0=states[c], want_gs=True, ckpt=ckpts[c])
    else:
        g = torch.Generator(device=dev).manual_seed(1)
        lens = torch.randint(64, 513, (B,), device=dev, generator=g)
        mask = (torch.arange(T, device=dev).view(1, T) < (lens.view(B, 1) - 1)).to(torch.int32).contiguous()
        tokens = int(lens.clamp(max=T).sum().item())

        bi_ws = wkv6_op.bi_new_workspace(B, T, C, H, dev)      # as WKV_6_BI keeps it from forward to backward

        def fwd():
            wkv6_op.bi_forward_ex(mask, r, k, v, w, u, H, ws=bi_ws)

        def bwd():
            wkv6_op.bi_backward_ex(mask, r, k, v, w, u, gy, H, ws=bi_ws)

    return fwd, bwd, tokens, B, T, H, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="wkv6", choices=["wkv6", "infctx", "bi", "dp_lora", "prefill"])
    ap.add_argument("--traffic", default="live", choices=["live", "file", "none"],
                    help="roofline.traffic of the headline workload: measured now under rocprofv3 (N = 1), from the committed PMC file, or omitted")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run also at --gpus 1 (the self-launch of `--gpus N`, on the box that exists)")
    ap.add_argument("--layers", type=int, default=24, help="dp_lora: number of RWKV blocks (1B6: 24)")
    ap.add_argument("--per-gpu-batch", type=int, default=32, help="dp_lora: triples per GPU and step")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args.workload)

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: spawn the ranks (nothing has touched the GPU yet) and relay
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # the profiled child runs come first, before this process makes ANY torch.cuda call (no fork + exec from a process that holds a
    # HIP context; the presence of a GPU is read from the device node, not from the runtime)
    pmc, pmc_source = {}, None
    if args.workload != "dp_lora" and rank == 0 and args.traffic != "none" and os.path.exists("/dev/kfd"):
        if args.traffic == "live" and world == 1:
            pmc, pmc_source = pmc_live(args.workload), "rocprofv3 --pmc child runs of this invocation"
        if not pmc and args.workload == "wkv6":
            pmc, pmc_source = pmc_from_file(), "profiles/" + os.path.basename(PMC_FILE)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the WKV6 operator has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # the rank path -- RCCL process group, barrier in the fence, MAX all-reduce of the elapsed time -- runs whenever a rendezvous
    # environment is present (torch.distributed.run sets it), also at world size 1: what a one-GPU box can execute of `--gpus N`
    if world > 1 or all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()                          # (the communicator exists before the pre-warm: its first collective is not a cold start)

    from rwkv_lm_ext_amd import wkv6_op

    if args.workload == "dp_lora":
        return bench_dp_lora(args, rank, world, dev, dist)
    fwd, bwd, tokens, B, T, H, name = build_workload(args.workload, dev, seed=rank)
    C = H * 64

    from rwkv_lm_ext_amd.dp import hold_until_all_ranks_ready, timed_steps
    # Everything the timed region needs exists BEFORE the pre-warm: events, the clock ring, the first use of every kernel (module load,
    # LDS attributes, the library's device self-test).  Between the pre-warm, the W warm-up steps and the first timed step the host does
    # nothing but launch: an idle gap of a few milliseconds re-arms the boost -> clamp -> recover transient of the power manager
    # (profiles/r06_dvfs_transient.txt), which is worth +-20 % on a 14 ms window.
    # two events per timed step, not three: a step's closing event IS the next step's opening one (an event record costs the stream ~2 us --
    # tools/event_overhead.py: 0 / 1 / 2 / 3 records per step = 0.5557 / 0.5569 / 0.5592 / 0.5617 ms per step -- and is instrumentation, not work)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps + 1)]
    counter = {"i": -args.warmup}
    launches_per_pass = 8 if args.workload == "infctx" else 1
    ring = launches_per_pass * (args.steps + args.warmup + 8)

    def step():                       # HIP events on the launch stream bracket fwd and bwd of every timed step
        i = counter["i"]
        counter["i"] += 1
        if i == 0:
            ev[0][0].record()
        fwd()
        if i >= 0:
            ev[i][1].record()
        bwd()
        if i >= 0:
            ev[i + 1][0].record()         # closes step i, opens step i + 1

    fwd()
    bwd()
    torch.cuda.synchronize()
    # in-run shader clock and duration of every launch of the warm-up and timed steps: wave 0 of the first 64 workgroups of the plain
    # chunked kernels stamps {s_memtime, s_memrealtime} at its start and end (four scalar instructions and two 8-byte stores per
    # workgroup and launch, inside the timed region like everything else) into a ring of the last `ring` launches per kernel
    # no garbage-collector pause between here and the end of the timed region: a generation-2 collection over the interpreter's ~10^6 objects
    # takes tens of milliseconds, and a host that stops launching for >= 1 ms re-arms the clock transient the pre-warm has just outlasted
    import gc
    gc.collect()
    gc.disable()
    with wkv6_op.ClockProbe(dev, n_slots=64, n_launches=ring) as probe:
        prewarm = prewarm_until_steady(fwd, bwd)
        # N > 1: the ranks converge at different moments; a ready rank keeps launching until all are (no idle GPU at the fence's barrier)
        prewarm["prewarm_hold_iters"] = hold_until_all_ranks_ready(lambda: (fwd(), bwd()), dist, dev)
        # under a profiler (tools/collect_profiles.sh) two empty marker kernels bracket the W warm-up + K timed steps in the dispatch list, so
        # that tools/aggregate_profiles.py can average the TIMED launches only (the last K of each kernel between the markers); both are
        # launches, not waits, and both lie outside the timed region
        marked = under_profiler()
        if marked:
            wkv6_op.pass_marker()
        elapsed = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dist, dev)
        if marked:
            wkv6_op.pass_marker()
        clocks = probe.read()
    gc.enable()
    fwd_steps = [ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)]
    bwd_steps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(args.steps)]
    fwd_ms = sum(fwd_steps) / args.steps
    bwd_ms = sum(bwd_steps) / args.steps

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        fwd_ghz = timed_launch_clocks(clocks, "fwd", args.steps, launches_per_pass)
        bwd_ghz = timed_launch_clocks(clocks, "bwd", args.steps, launches_per_pass)
        fwd_kus = timed_launch_clocks(clocks, "fwd", args.steps, launches_per_pass, "us")
        bwd_kus = timed_launch_clocks(clocks, "bwd", args.steps, launches_per_pass, "us")
        seq_tokens = T // launches_per_pass                   # tokens one workgroup walks per launch

        def mean_or_none(xs):
            return round(sum(xs) / len(xs), 3) if xs else None

        def cycles_per(ms, ghz, unit_tokens):
            if not ghz or args.workload == "bi":              # (ragged rows, persistent slots: no fixed unit count per workgroup)
                return None
            return round(ms / launches_per_pass * 1e-3 * mean_or_none(ghz) * 1e9 / (seq_tokens / unit_tokens), 1)
        units = tokens * C                                    # token-channels per step per GPU
        dom_name, dom_ms, dom_b = ("backward", bwd_ms, BWD_BYTES) if bwd_ms >= fwd_ms else ("forward", fwd_ms, FWD_BYTES)
        dom_kernel = "chunk_fwd_kernel" if dom_name == "forward" else "chunk_bwd12k_kernel"
        ach = units * dom_b / (dom_ms * 1e-3) / 1e9

        def traffic_of(kernel, side):
            """HBM bytes of the dominant pass: one launch at the headline workload (per-launch average of its kernel), every
            launch of the pass where a pass is several launches (wkv6_bi: two scans + helpers; infctx: eight chunks)."""
            if args.workload == "wkv6" and pmc.get(kernel):
                return pmc[kernel]["hbm_bytes"]
            return pmc.get(side, {}).get("hbm_bytes")
        step_bytes = FWD_BYTES if args.workload == "prefill" else FWD_BYTES + BWD_BYTES
        step_ach = units * step_bytes / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
        out = {
            "metric": "WKV6 forward tokens/sec/GPU (inference prefill)" if args.workload == "prefill"
            else "WKV6 fwd+bwd tokens/sec/GPU (B=8,T=4096,C=2048) + %HBM roofline",
            "value": round(world * tokens * args.steps / elapsed, 1),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, **prewarm,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": name, "io": "bf16",
                       "math": "split-bf16 (hi+lo) MFMA operands, fp32 accumulate and state",
                       "tokens_per_gpu": tokens, "channels": C,
                       "parallelism": f"dp{world} (independent batches per GPU, no data-path collective)",
                       "rank_path": "nccl" if dist is not None else None,     # RCCL group + barrier + MAX all-reduce of the time executed
                       "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
                       # every timed step by itself (HIP events on the launch stream): a steady state shows as flat lists
                       "fwd_ms_steps": [round(x, 4) for x in fwd_steps], "bwd_ms_steps": [round(x, 4) for x in bwd_steps],
                       # ... and the kernels alone, from the same launches' in-kernel s_memrealtime stamps (last wave's end - first wave's start): the
                       # first timed step's EVENT interval also holds the launch latency behind the fence's synchronize (~25 us once per run)
                       "fwd_kernel_ms_steps": fwd_kus or None, "bwd_kernel_ms_steps": bwd_kus or None,
                       "first_last_step_ratio": round((fwd_kus[-1] + bwd_kus[-1]) / (fwd_kus[0] + bwd_kus[0]), 4) if fwd_kus and bwd_kus
                       else round((fwd_steps[-1] + bwd_steps[-1]) / (fwd_steps[0] + bwd_steps[0]), 4),
                       # drift indicator that single-launch jitter (+-3 % in the steadiest state: profiles/r06_dvfs_transient.txt, table A) does
                       # not dominate: mean of the last quarter of the timed steps / mean of the first quarter (kernels alone)
                       "tail_head_ratio": (lambda ks: round((sum(ks[-max(1, len(ks) // 4):]) / max(1, len(ks) // 4)) /
                                                            (sum(ks[:max(1, len(ks) // 4)]) / max(1, len(ks) // 4)), 4))(
                           [a + b for a, b in zip(fwd_kus, bwd_kus)] if fwd_kus and bwd_kus else [a + b for a, b in zip(fwd_steps, bwd_steps)]),
                       "step_spread": round((max(a + b for a, b in zip(fwd_steps, bwd_steps)) - min(a + b for a, b in zip(fwd_steps, bwd_steps)))
                                            / (fwd_ms + bwd_ms), 4),
                       # in-kernel shader clock of the timed launches of each kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz, median
                       # over workgroups (MI355X_MICROARCH.md, DVFS give-back item 6): mean over the timed steps, first and last step
                       "fwd_clock_ghz": mean_or_none(fwd_ghz), "bwd_clock_ghz": mean_or_none(bwd_ghz),
                       "fwd_clock_ghz_first_last": [fwd_ghz[0], fwd_ghz[-1]] if fwd_ghz else None,
                       "bwd_clock_ghz_first_last": [bwd_ghz[0], bwd_ghz[-1]] if bwd_ghz else None,
                       # kernel time in shader cycles per workgroup and unit of its loop (ms x in-run GHz / units): what a code change
                       # changes; the clock is what the box and the power manager change
                       "cycles_per_group": cycles_per(fwd_ms, fwd_ghz, GROUP_TOKENS),
                       "cycles_per_stage": cycles_per(bwd_ms, bwd_ghz, STAGE_TOKENS)},
            "roofline": {"bound": "hbm", "kernel": dom_name,
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4),
                         "traffic": traffic_of(dom_kernel, "_pass_bwd" if dom_name == "backward" else "_pass_fwd"),
                         "traffic_source": pmc_source if (pmc.get(dom_kernel) or pmc.get("_pass_fwd")) else None,
                         "algorithmic_bytes_per_launch": units * dom_b, "avg_ms": round(dom_ms, 4),
                         "valu_issue_share": valu_issue_share_of(pmc.get(dom_kernel, {}).get("counters", {})),
                         "valu_issue_share_note": "sum of the resident waves' VALU issue cycles / kernel cycles (4 cycles charged per "
                                                  "instruction): instruction density, not pipe occupancy (profiles/r05_issue_floor.md)"},
            "roofline_step": {"bound": "hbm", "achieved": round(step_ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(step_ach / HBM_PEAK_GBPS, 4),
                              "algorithmic_bytes": units * step_bytes,
                              "traffic": (pmc["_pass_fwd"]["hbm_bytes"] + pmc.get("_pass_bwd", {}).get("hbm_bytes", 0))
                              if pmc.get("_pass_fwd") else None},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
