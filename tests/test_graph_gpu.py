"""The operator under HIP-graph capture (torch.cuda.CUDAGraph): every entry point launches on the caller's stream with caller-owned
buffers, and the workspace-less reference-signature symbols take their scratch from the library's stream-ordered pool
(hipMallocFromPoolAsync, capturable), so a training step's operator calls -- or a decode step's -- replay from a graph with the results
of the eager calls.  (launch-bound inner loops belong in hipGraphs.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def rnd(*shape, scale=1.0, seed=0, dtype=bf):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()                                       # warm-up on the capture stream: library load, self-test, attribute calls, pool creation
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    return graph


def test_training_calls_replay_from_a_graph():
    from rwkv_lm_ext_amd import wkv6_op as op
    B, T, H = 2, 200, 2
    C = 64 * H
    r, k, v, gy = (rnd(B, T, C, scale=0.5, seed=s) for s in (1, 2, 3, 4))
    w = (rnd(B, T, C, scale=0.7, seed=5).float() - 2.0).to(bf)
    u = rnd(H, 64, scale=0.3, seed=6)
    ckpt = op.new_checkpoint(B, T, C, H, r.device)
    y_ref = op.forward_ex(r, k, v, w, u, H, ckpt=ckpt)
    g_ref = op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
    # static buffers of the captured step
    y = torch.empty_like(r)
    outs = {}

    def step():
        op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
        outs["g"] = op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)

    graph = capture(step)
    captured = outs["g"]                            # the tensors the captured backward writes
    y.zero_()
    for t in captured[:5]:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    for name, a_, b_ in zip("gr gk gv gw gu".split(), captured, g_ref):
        assert torch.equal(a_, b_), name
    # new inputs in the same buffers: the graph computes on what the buffers hold at replay time
    r.copy_(rnd(B, T, C, scale=0.5, seed=11))
    y2 = op.forward_ex(r, k, v, w, u, H)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


def test_reference_signature_symbols_without_workspace_replay_from_a_graph():
    """wkv6_cuda.backward has no workspace argument (cuda/wkv6_op.cpp:10-13): its scratch is a stream-ordered pool allocation made
    inside the call, which must be legal while the stream is capturing."""
    from rwkv_lm_ext_amd import wkv6_op as op
    B, T, H = 2, 96, 2
    C = 64 * H
    r, k, v, gy = (rnd(B, T, C, scale=0.5, seed=s) for s in (21, 22, 23, 24))
    w = (rnd(B, T, C, scale=0.7, seed=25).float() - 2.0)
    ew = (-torch.exp(w.to(bf).float())).contiguous()
    u = rnd(H, 64, scale=0.3, seed=26)
    outs = [torch.empty(B, T, C, device="cuda", dtype=bf) for _ in range(5)]
    gu = torch.empty(B, C, device="cuda", dtype=bf)

    def step():
        op.wkv6_cuda.forward(B, T, C, H, r, k, v, ew, u, outs[0])
        op.wkv6_cuda.backward(B, T, C, H, r, k, v, ew, u, gy, *outs[1:], gu)

    step()
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs] + [gu.clone()]
    graph = capture(step)
    for t in outs + [gu]:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a_, b_ in zip(outs + [gu], ref):
        assert torch.equal(a_, b_)


def test_decode_step_replays_from_a_graph():
    from rwkv_lm_ext_amd.wkv6_op import rwkv6
    B, T, H, L = 4, 1, 2, 6
    C = 64 * H
    r, k, v = (rnd(B, T, C, scale=0.5, seed=s) for s in (31, 32, 33))
    w = torch.exp(-torch.exp(rnd(B, T, C, seed=34, dtype=torch.float32) - 2.0)).contiguous()
    u = rnd(H, 64, scale=0.3, seed=35)
    states = [torch.zeros(B, H, 64, 64, device="cuda") for _ in range(L)]
    ys = [torch.empty(B, T, C, device="cuda", dtype=bf) for _ in range(L)]

    def step():
        for st, y in zip(states, ys):
            rwkv6.forward_bf16(B, T, C, H, st, r, k, v, w, u, y)

    for _ in range(3):                              # three eager steps
        step()
    torch.cuda.synchronize()
    want_s, want_y = [s.clone() for s in states], [y.clone() for y in ys]
    for s in states:
        s.zero_()
    graph = capture(step)                           # (the capture helper runs one warm-up step and captures a second without executing it)
    for s in states:
        s.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a_, b_ in zip(states + ys, want_s + want_y):
        assert torch.equal(a_, b_)
