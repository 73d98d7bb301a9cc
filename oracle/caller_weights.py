"""Deterministic tiny-model weights for the caller fixtures (TEST INFRASTRUCTURE ONLY).

The same seeded generator is used by oracle/gen_golden_callers.py (to load the REFERENCE modules) and by the tests
(to load rwkv_lm_ext_amd.callers), so the fixtures only need to store inputs and reference outputs.  Keys are the
reference's own state_dict names (src/model_encoder_run.py:96-290).
"""
import torch

N_EMBD, DIM_ATT, DIM_FFN, N_LAYER, VOCAB, HEAD = 128, 128, 256, 2, 100, 64


def tmix_weights(g, prefix="", n_embd=N_EMBD, dim_att=DIM_ATT, layer_id=0, n_layer=N_LAYER):
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    w = {}
    for n in ("x", "w", "k", "v", "r", "g"):
        w[f"{prefix}time_maa_{n}"] = torch.rand(1, 1, n_embd, generator=g)
    w[f"{prefix}time_maa_w1"] = r(n_embd, 160, scale=0.1)
    w[f"{prefix}time_maa_w2"] = r(5, 32, n_embd, scale=0.1)
    ramp = torch.tensor([-6 + 5 * (n / (dim_att - 1)) ** (0.7 + 1.3 * layer_id / max(n_layer - 1, 1)) for n in range(dim_att)])
    w[f"{prefix}time_decay"] = (ramp + 0.3 * torch.randn(dim_att, generator=g)).reshape(1, 1, dim_att)
    w[f"{prefix}time_decay_w1"] = r(n_embd, 64, scale=0.1)
    w[f"{prefix}time_decay_w2"] = r(64, dim_att, scale=0.1)
    w[f"{prefix}time_faaaa"] = r(dim_att // HEAD, HEAD, scale=0.3)
    for n in ("receptance", "key", "value", "gate"):
        w[f"{prefix}{n}.weight"] = r(dim_att, n_embd, scale=n_embd ** -0.5)
    w[f"{prefix}output.weight"] = r(n_embd, dim_att, scale=dim_att ** -0.5)
    w[f"{prefix}ln_x.weight"] = 1 + 0.1 * torch.randn(dim_att, generator=g)
    w[f"{prefix}ln_x.bias"] = 0.1 * torch.randn(dim_att, generator=g)
    return w


def cmix_weights(g, prefix="", n_embd=N_EMBD, dim_ffn=DIM_FFN):
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    return {
        f"{prefix}time_maa_k": torch.rand(1, 1, n_embd, generator=g),
        f"{prefix}time_maa_r": torch.rand(1, 1, n_embd, generator=g),
        f"{prefix}key.weight": r(dim_ffn, n_embd, scale=n_embd ** -0.5),
        f"{prefix}receptance.weight": r(n_embd, n_embd, scale=n_embd ** -0.5),
        f"{prefix}value.weight": r(n_embd, dim_ffn, scale=dim_ffn ** -0.5),
    }


def encoder_weights(seed=1234):
    g = torch.Generator().manual_seed(seed)
    w = {"emb.weight": torch.randn(VOCAB, N_EMBD, generator=g) * 0.5}
    ln = lambda p: {p + "weight": 1 + 0.1 * torch.randn(N_EMBD, generator=g), p + "bias": 0.1 * torch.randn(N_EMBD, generator=g)}
    for i in range(N_LAYER):
        p = f"blocks.{i}."
        w.update(ln(p + "ln1."))
        w.update(ln(p + "ln2."))
        if i == 0:
            w.update(ln(p + "ln0."))
        w.update(tmix_weights(g, p + "att.", layer_id=i))
        w.update(cmix_weights(g, p + "ffn."))
    w.update(ln("ln_out."))
    return w
