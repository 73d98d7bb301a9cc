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
