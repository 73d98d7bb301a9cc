"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the batch dealing and the
max-over-ranks timing that bench.py uses (no GPU involved)."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rwkv_lm_ext_amd.dp import BucketBatchSampler, hold_until_all_ranks_ready, shard_rows, timed_steps


def test_sampler_deals_disjoint_contiguous_slices():
    cum, bss = [64, 64 + 48, 64 + 48 + 40], [8, 4, 2]
    for world in (1, 2, 4):
        per_rank = [list(BucketBatchSampler(cum, bss, r, world)) for r in range(world)]
        n = len(BucketBatchSampler(cum, bss, 0, world))
        assert all(len(p) == n for p in per_rank)
        assert n == 64 // (8 * world) + 48 // (4 * world) + 40 // (2 * world)
        seen = set()
        for step in range(n):
            sizes = {len(per_rank[r][step]) for r in range(world)}
            assert len(sizes) == 1                                  # every rank gets the same batch size
            idx = sorted(i for r in range(world) for i in per_rank[r][step])
            assert idx == list(range(idx[0], idx[0] + len(idx)))    # one contiguous block per step, rank-strided
            assert not (seen & set(idx))
            seen |= set(idx)
    # reference arithmetic, first step of bucket 0 at world 2: rank r starts at 64 - 4*8*2 + r*8
    assert next(iter(BucketBatchSampler(cum, bss, 1, 2))) == list(range(8, 16))
    # resume (reference quirk, data/custom_datasets.py:47-50): skipped batches are burned from the CURRENT
    # bucket without rotating to the next one, so 3 skips eat 3 of bucket 0's 4 steps
    full = list(BucketBatchSampler(cum, bss, 0, 2))
    resumed = list(BucketBatchSampler(cum, bss, 0, 2, skipped_batches=3))
    assert len(resumed) == len(full) - 3 == len(BucketBatchSampler(cum, bss, 0, 2, skipped_batches=3))
    assert resumed[0] == list(range(48, 56))
    flat = [i for b in resumed for i in b]
    assert len(flat) == len(set(flat))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = shard_rows(8, rank, world)
    x = torch.arange(8.0)[rows.start:rows.stop]
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))          # rank 1 is the straggler

    el = timed_steps(step, steps=3, warmup=1, device_sync=lambda: None, dist=dist)
    tot = x.sum()
    dist.all_reduce(tot)
    q.put((rank, list(rows), el, len(calls), float(tot)))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, rows0, el0, n0, tot0), (r1, rows1, el1, n1, tot1) = res
    assert rows0 == [0, 1, 2, 3] and rows1 == [4, 5, 6, 7]         # disjoint cover of the global batch
    assert n0 == n1 == 4                                           # 1 warm-up + 3 timed steps each
    assert el0 == el1 and el0 >= 3 * 0.04 * 0.9                    # both report the slowest rank's time
    assert tot0 == tot1 == 28.0


def _hold_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    time.sleep(0.4 * rank)                     # rank 1's "pre-warm" converges 0.4 s later than rank 0's
    t_ready = time.perf_counter()
    extra = hold_until_all_ranks_ready(lambda: time.sleep(0.005), dist)
    held = time.perf_counter() - t_ready
    t_left = time.time()
    dist.barrier()
    q.put((rank, extra, held, t_left))
    dist.destroy_process_group()


def test_ready_ranks_keep_launching_until_every_rank_is_ready():
    """bench.py's multi-rank pre-warm hand-over (dp.hold_until_all_ranks_ready): the early rank does not idle -- it keeps stepping for as
    long as the late rank needs -- and both leave within a few steps of each other."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hold_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, extra0, held0, left0), (_, extra1, held1, left1) = res
    assert extra0 >= 30 and held0 >= 0.3           # rank 0 stepped through rank 1's remaining 0.4 s (5 ms steps)
    assert extra1 <= 10 and held1 < 0.2            # rank 1 was the last to arrive: released almost at once
    assert abs(left0 - left1) < 0.1                # and they left together
    assert hold_until_all_ranks_ready(lambda: None, None) == 0
