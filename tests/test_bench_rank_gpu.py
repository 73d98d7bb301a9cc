"""The rank path of the operator bench on the one GPU a box has: `torch.distributed.run --nproc-per-node=1 bench.py
--gpus 1` in a fresh child process -- the `nccl` (= RCCL) process group, the barrier of dp.timed_steps' fence and the MAX all-reduce of
the elapsed time all execute, and the ONE-JSON-line contract survives RCCL's version banner.  Reference topology: one process per GPU,
/root/reference/peft_train/peft_train_bi_encoder.py:290-311."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # a child of the pytest process (which is never replaced); the child has not touched the GPU when torch.distributed.run starts its rank
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
def test_operator_bench_through_torchrun_at_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--traffic", "none"]
    res = _run(cmd)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # exactly one line on stdout: the banner went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["rank_path"] == "nccl"
    assert out["value"] > 1e6 and out["scaling"] == "weak"
    assert len(out["config"]["fwd_ms_steps"]) == 2


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` started without a launcher re-launches itself under torch.distributed.run (bench.py main); `--spawn` takes
    that path at N = 1: the spawn, the relay of the rank's one JSON line and the exit code."""
    res = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--no-cpu",
                "--traffic", "none"])
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["rank_path"] == "nccl" and out["value"] > 1e6
