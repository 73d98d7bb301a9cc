#!/usr/bin/env python3
"""Golden vectors for the callers of the WKV6 operator, from the REFERENCE modules.

Build container only (needs /root/reference):   python oracle/gen_golden_callers.py

Imports src/model_encoder_run.py with NO_CUDA=1 (its WKV is the pure-python run_rwkv6_forward, :30-62), builds
BiRWKV_Tmix_x060 / BiRWKV_CMix_x060 / RwkvEncoder (:96-348) at a tiny shape, loads the deterministic weights of
oracle/caller_weights.py and stores inputs + outputs (weights are regenerated from the seed by the tests).
"""
import os
import sys
from types import SimpleNamespace

REF = "/root/reference"
os.environ.setdefault("NO_CUDA", "1")
os.environ.setdefault("RWKV_HEAD_SIZE_A", "64")
os.environ.setdefault("RWKV_FLOAT_MODE", "fp32")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch

import src.model_encoder_run as ref                                    # reference
from oracle import caller_weights as cw

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    torch.manual_seed(0)
    args = SimpleNamespace(n_embd=cw.N_EMBD, dim_att=cw.DIM_ATT, dim_ffn=cw.DIM_FFN, n_layer=cw.N_LAYER,
                           head_size_a=cw.HEAD, head_size_divisor=8, vocab_size=cw.VOCAB, ctx_len=64, my_pos_emb=0,
                           pre_ffn=0, tiny_att_dim=0, tiny_att_layer=-1, head_qk=0, dropout=0.0)
    g = torch.Generator().manual_seed(7)
    B, T = 2, 24
    x = torch.randn(B, T, cw.N_EMBD, generator=g)

    # ---- a13: time-mix around the op; a15: composition C on one layer
    tm = ref.BiRWKV_Tmix_x060(args, 1)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    mask = torch.ones(B, T, dtype=torch.int)
    mask[0, 20:] = 0
    mask[1, 13:] = 0
    rev_idx = ref.reverse_x_idx(mask, T)
    with torch.no_grad():
        r, k, v, gg, w = tm.jit_func(x)
        y = ref.run_rwkv6_forward(r, k, v, w, tm.time_faaaa)
        out = tm.jit_func_2(y, gg)
        out_bi = tm(x, rev_idx, mask)
    # ---- a14: channel mix
    cm = ref.BiRWKV_CMix_x060(args, 1)
    cm.load_state_dict(cw.cmix_weights(torch.Generator().manual_seed(12)), strict=True)
    with torch.no_grad():
        cm_out = cm(x)
    # ---- a15: whole encoder, padded batch, embedding marker (id 1), pad (id 0)
    enc = ref.RwkvEncoder(args)
    enc.load_state_dict(cw.encoder_weights(), strict=True)
    idx = torch.randint(4, cw.VOCAB, (3, 20), generator=g)
    idx[0, 15] = 1; idx[0, 16:] = 0
    idx[1, 19] = 1
    idx[2, 7] = 1; idx[2, 8:] = 0
    with torch.no_grad():
        logits, hidden = enc(idx, True)
        sent = enc.encode_sentence(idx)
    np.savez_compressed(os.path.join(OUT, "callers.npz"), x=x.numpy(), mask=mask.numpy(), rev_idx=rev_idx.numpy(),
                        r=r.numpy(), k=k.numpy(), v=v.numpy(), g=gg.numpy(), w=w.numpy(), y=y.numpy(), out=out.numpy(),
                        out_bi=out_bi.numpy(), cm_out=cm_out.numpy(), idx=idx.numpy(), logits=logits.numpy(),
                        hidden=hidden.numpy(), sent=sent.numpy())
    print("wrote tests/golden/callers.npz", os.path.getsize(os.path.join(OUT, "callers.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
