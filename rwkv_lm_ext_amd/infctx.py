"""Carried-state ("infctx") containers and block forward for long sequences processed in chunks.

Mirrors src/infctx_module.py:1-50 (TimeMixState, ChannelMixState, BlockState, BlockStateList; wkv states
[L, B, H, N, N] in bf16, value-major like the kernels, shift states [L, 2, B, C]) and the *_infctx module forwards of
src/model.py:738-812: token shift starts from the previous chunk's last token, the WKV starts from the carried
state and hands the final state on (RUN_CUDA_RWKV6_STATE of the 'infctx' flavour, src/model.py:130-132).
"""
import torch


class TimeMixState:
    def __init__(self, shift_state, wkv_state):
        self.shift_state = shift_state
        self.wkv_state = wkv_state


class ChannelMixState:
    def __init__(self, shift_state):
        self.shift_state = shift_state


class BlockState:
    def __init__(self, time_mix_state, channel_mix_state):
        self.time_mix_state = time_mix_state
        self.channel_mix_state = channel_mix_state


class BlockStateList:
    def __init__(self, shift_states, wkv_states):
        self.wkv_states = wkv_states
        self.shift_states = shift_states

    @staticmethod
    def empty(N, B, C, H, device, dtype):
        wkv_states = torch.empty((N, B, H, C // H, C // H), device=device, dtype=torch.bfloat16)
        shift_states = torch.empty((N, 2, B, C), device=device, dtype=dtype)
        return BlockStateList(shift_states, wkv_states)

    @staticmethod
    def create(N, B, C, H, device, dtype):
        result = BlockStateList.empty(N, B, C, H, device, dtype)
        result.wkv_states[:] = 0
        result.shift_states[:] = 0
        return result

    def __getitem__(self, layer):
        return BlockState(TimeMixState(self.shift_states[layer, 0], self.wkv_states[layer]),
                          ChannelMixState(self.shift_states[layer, 1]))

    def __setitem__(self, layer, state):
        self.shift_states[layer, 0] = state.time_mix_state.shift_state
        self.wkv_states[layer] = state.time_mix_state.wkv_state
        self.shift_states[layer, 1] = state.channel_mix_state.shift_state


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
