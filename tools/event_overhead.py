import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from rwkv_lm_ext_amd import wkv6_op
dev = torch.device('cuda', 0)
fwd, bwd = bench.build_workload('wkv6', dev)[:2]
fwd(); bwd(); torch.cuda.synchronize()
def timed(n, mode):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if mode >= 1: evs[i][0].record()
        fwd()
        if mode >= 2: evs[i][1].record()
        bwd()
        if mode >= 3: evs[i][2].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
bench.prewarm_until_steady(fwd, bwd)
for rep in range(3):
    for mode in (0, 1, 2, 3):
        for _ in range(50): fwd(); bwd()
        print(rep, 'events per step', mode, 'ms per step %.4f' % timed(200, mode), flush=True)
