#!/usr/bin/env python3
"""Channel-mix FFN: error of the fused (HIP glue kernels) and the eager bf16 forms against the fp32 module on the same inputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkv_lm_ext_amd import callers

torch.manual_seed(0)
bf = torch.bfloat16
for C, F_ in ((128, 448), (2048, 7168)):
    cm32 = callers.CMix_x060(C, F_)
    with torch.no_grad():
        cm32.time_maa_k.uniform_(0.1, 0.9); cm32.time_maa_r.uniform_(0.1, 0.9)
        for lin in (cm32.key, cm32.receptance, cm32.value):
            lin.weight.normal_(0, 0.05)
    x = torch.randn(4, 64, C).to(bf)
    cm16 = callers.CMix_x060(C, F_).cuda().to(bf)
    cm16.load_state_dict({k: v.to(bf) for k, v in cm32.state_dict().items()})
    cm32w = callers.CMix_x060(C, F_)
    cm32w.load_state_dict({k: v.float() for k, v in cm16.state_dict().items()})    # fp32 math on the bf16-rounded parameters
    ref = cm32w(x.float())
    for fused in (True, False):
        cm16.fused = fused
        y = cm16(x.cuda()).float().cpu()
        err = (y - ref).abs()
        print(f"C={C} fused={fused}: max-normalised {float(err.max() / ref.abs().max()):.3e}  rel-rms {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
