"""MI355X (gfx950) implementation of the RWKV-6 WKV operator family of yynil/RWKV_LM_EXT.

Public surface (mirrors the reference):
  rwkv_lm_ext_amd.wkv6_op   -- `wkv6_cuda`, `wkv6_bi_cuda`, `wkv6state_cuda`, `wkv6infctx_cuda` objects with
                               the forward/backward signatures of cuda/wkv6*_op.cpp
  rwkv_lm_ext_amd.wkv       -- WKV_6, WKV_6STATE, WKV_6_BI autograd.Functions, RUN_CUDA_RWKV6[_STATE]
"""
__version__ = "0.1"
