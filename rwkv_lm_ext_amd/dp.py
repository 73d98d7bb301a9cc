"""Data-parallel host logic for the WKV6 path: how batches are dealt to ranks and how a multi-rank run is
timed.  No data-path collective exists on this path -- every (batch, head) recurrence is independent
(cuda/wkv6_cuda.cu:11-12, 233) -- so ranks only meet at the timing barrier (and, in training, at the
gradient all-reduce that torch's DDP performs over RCCL).

`BucketBatchSampler` restates the reference's rank-strided dealing (data/custom_datasets.py:19-74,
`MyBatchSampler`): the dataset is a concatenation of length buckets with cumulative sizes, bucket i uses
batch size bs_i, buckets are visited round-robin, and in each visit rank r of W takes the contiguous slice
[end_i - remaining_i*bs_i*W + r*bs_i, ... + bs_i).  len() = sum_i size_i // (bs_i * W) (minus skipped).
"""
import time
from typing import Callable, Iterator, List, Optional, Sequence

import torch


class BucketBatchSampler:
    def __init__(self, cumulative_sizes: Sequence[int], batch_sizes: Sequence[int], rank: int = 0,
                 world_size: int = 1, skipped_batches: int = 0):
        assert len(cumulative_sizes) == len(batch_sizes) and world_size >= 1 and 0 <= rank < world_size
        self.cumulative_sizes = list(cumulative_sizes)
        self.batch_sizes = list(batch_sizes)
        self.rank, self.world_size, self.skipped_batches = rank, world_size, skipped_batches

    def _steps_per_bucket(self) -> List[int]:
        out, prev = [], 0
        for end, bs in zip(self.cumulative_sizes, self.batch_sizes):
            out.append((end - prev) // (bs * self.world_size))
            prev = end
        return out

    def __len__(self) -> int:
        return sum(self._steps_per_bucket()) - self.skipped_batches

    def __iter__(self) -> Iterator[List[int]]:
        remaining = self._steps_per_bucket()
        nb, cur, skipped = len(remaining), 0, 0
        while sum(remaining) > 0:
            while remaining[cur] == 0:
                cur = (cur + 1) % nb
            bs = self.batch_sizes[cur]
            if skipped < self.skipped_batches:          # resume support: burn batches without yielding
                skipped += 1
                remaining[cur] -= 1
                continue
            first = self.cumulative_sizes[cur] - remaining[cur] * bs * self.world_size + self.rank * bs
            remaining[cur] -= 1
            cur = (cur + 1) % nb
            yield list(range(first, first + bs))


def shard_rows(global_batch: int, rank: int, world_size: int) -> range:
    """Rows of a [global_batch, ...] tensor that rank `rank` owns (contiguous, equal shares)."""
    assert global_batch % world_size == 0, "global batch must divide evenly over the ranks"
    per = global_batch // world_size
    return range(rank * per, (rank + 1) * per)


def timed_steps(step: Callable[[], None], steps: int, warmup: int, device_sync: Callable[[], None],
                dist=None, device: Optional[torch.device] = None) -> float:
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over the ranks."""
    def fence():
        if dist is not None and dist.is_initialized():
            dist.barrier()
        device_sync()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    elapsed = time.perf_counter() - t0
    fence()
    if dist is not None and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def hold_until_all_ranks_ready(step: Callable[[], None], dist=None, device: Optional[torch.device] = None, max_s: float = 5.0) -> int:
    """Multi-rank pre-warm hand-over: each rank's own pre-warm ends when ITS step times have converged, which is not the same moment on
    every rank -- and a rank that then sat idle at the timing fence's barrier would enter its timed steps on a cold GPU (an idle gap of
    >= 1 ms re-arms the power manager's boost -> clamp -> recover transient, worth up to -18 % on the next 20 steps: bench.py,
    profiles/r06_dvfs_transient.txt).  So a rank that is ready posts an asynchronous all-reduce and KEEPS LAUNCHING `step`, one per poll,
    until the collective completes, i.e. until every rank is ready: all ranks leave within one step of each other and reach the fence's
    barrier together.  Returns the number of extra steps launched (0 without a process group)."""
    if dist is None or not dist.is_initialized():
        return 0
    flag = torch.ones(1, dtype=torch.float32, device=device or "cpu")
    work = dist.all_reduce(flag, async_op=True)
    extra, t0 = 0, time.perf_counter()
    while not work.is_completed():
        step()
        extra += 1
        if time.perf_counter() - t0 > max_s:       # a peer is far behind (or gone): stop feeding, wait for it
            break
    work.wait()
    return extra
