"""Carried-state ("infctx") containers and block forward for long sequences processed in chunks.

Mirrors src/infctx_module.py:1-50 (TimeMixState, ChannelMixState, BlockState, BlockStateList; wkv states
[L, B, H, N, N] in bf16, value-major like the kernels, shift states [L, 2, B, C]) and the *_infctx module forwards of
src/model.py:738-812: token shift starts from the previous chunk's last token, the WKV starts from the carried
state and hands the final state on (RUN_CUDA_RWKV6_STATE of the 'infctx' flavour, src/model.py:130-132).
"""
import torch


from dataclasses import dataclass


@dataclass
class TimeMixState:
    """What a time-mix layer carries to the next chunk: the last token (for the token shift) and the WKV state [B, H, N, N]."""
    shift_state: torch.Tensor
    wkv_state: torch.Tensor


@dataclass
class ChannelMixState:
    """What a channel-mix layer carries: the last token of the chunk."""
    shift_state: torch.Tensor


@dataclass
class BlockState:
    time_mix_state: TimeMixState
    channel_mix_state: ChannelMixState


_TMIX, _CMIX = 0, 1          # slot of each sub-layer in shift_states[layer]


class BlockStateList:
    """All layers' carried state in two tensors: `wkv_states` [L, B, H, N, N] (always bf16: the kernels' state dtype) and
    `shift_states` [L, 2, B, C] (activation dtype).  Indexing with a layer number gives views, assignment copies into them.
    Same constructor, factories, attribute names and indexing protocol as the reference's container (src/infctx_module.py:20-50),
    because training scripts build and index it directly."""

    def __init__(self, shift_states, wkv_states):
        self.shift_states, self.wkv_states = shift_states, wkv_states

    @classmethod
    def _allocate(cls, alloc, n_layer, B, C, H, device, dtype):
        head = C // H
        return cls(alloc((n_layer, 2, B, C), device=device, dtype=dtype),
                   alloc((n_layer, B, H, head, head), device=device, dtype=torch.bfloat16))

    @staticmethod
    def empty(N, B, C, H, device, dtype):
        return BlockStateList._allocate(torch.empty, N, B, C, H, device, dtype)

    @staticmethod
    def create(N, B, C, H, device, dtype):
        return BlockStateList._allocate(torch.zeros, N, B, C, H, device, dtype)

    def __len__(self):
        return self.wkv_states.shape[0]

    def __getitem__(self, layer):
        shift = self.shift_states[layer]
        return BlockState(TimeMixState(shift[_TMIX], self.wkv_states[layer]), ChannelMixState(shift[_CMIX]))

    def __setitem__(self, layer, state):
        tm, cm = state.time_mix_state, state.channel_mix_state
        self.wkv_states[layer].copy_(tm.wkv_state)
        self.shift_states[layer, _TMIX].copy_(tm.shift_state)
        self.shift_states[layer, _CMIX].copy_(cm.shift_state)


def _default_wkv_state(B, T, C, H, r, k, v, w, u, s):
    from .wkv import RUN_CUDA_RWKV6_INFCTX
    bf = torch.bfloat16
    y, s = RUN_CUDA_RWKV6_INFCTX(B, T, C, H, *(t.to(bf).contiguous() for t in (r, k, v, w, u)), s)
    return y.to(r.dtype), s


def tmix_forward_infctx(tm, x, last_state, wkv_state=None):
    """RWKV_Tmix_x060_infctx.forward (src/model.py:773-782) for a callers.Tmix_x060 `tm`."""
    B, T, C = x.size()
    shifted = torch.cat((last_state.shift_state.unsqueeze(1), x[:, :-1]), dim=1)
    r, k, v, g, w = tm.jit_func(x, shifted=shifted)
    s = last_state.wkv_state.clone().contiguous()
    y, s = (wkv_state or _default_wkv_state)(B, T, C, tm.n_head, r, k, v, w, tm.time_faaaa, s)
    return tm.jit_func_2(y, g), TimeMixState(x[:, -1], s)


def cmix_forward_infctx(cm, x, last_state):
    """RWKV_CMix_x060_infctx.forward (src/model.py:803-812) for a callers.CMix_x060 `cm`."""
    xx = torch.cat((last_state.shift_state.unsqueeze(1), x[:, :-1]), dim=1) - x
    k = torch.relu(cm.key(x + xx * cm.time_maa_k)) ** 2
    return torch.sigmoid(cm.receptance(x + xx * cm.time_maa_r)) * cm.value(k), ChannelMixState(x[:, -1])
