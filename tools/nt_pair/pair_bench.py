import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from asr_chinese_e2e_amd import kernels as K
from asr_chinese_e2e_amd._lib import ACT_RELU, ACT_RELU_MASK
DEV = "cuda"
M, d, ff = 16000, 512, 1024
x = torch.randn(M, d, device=DEV).bfloat16()
w1, w2 = (torch.randn(ff, d, device=DEV) * 0.05).bfloat16(), (torch.randn(d, ff, device=DEV) * 0.05).bfloat16()
b1, b2 = torch.randn(ff, device=DEV) * 0.1, torch.randn(d, device=DEV) * 0.1
h, o = torch.empty(M, ff, dtype=torch.bfloat16, device=DEV), torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
dy = torch.randn(M, d, device=DEV).bfloat16()
w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
dh, dx = torch.empty_like(h), torch.empty_like(o)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def two_f(): K.gemm_nt(x, w1, b1, h, act=ACT_RELU); K.gemm_nt(h, w2, b2, o)
def pair_f(): K.gemm_nt_pair(x, w1, b1, h, w2, b2, o)
def two_b(): K.gemm_nt(dy, w2t, None, dh, act=ACT_RELU_MASK, res=h); K.gemm_nt(dh, w1t, None, dx)
def pair_b(): K.gemm_nt_pair(dy, w2t, None, dh, w1t, None, dx, act1=ACT_RELU_MASK, mask1=h)
two_f()
for r in range(3):
    print(f"forward pair : two launches {t(two_f):6.1f} us, one launch {t(pair_f):6.1f} us | backward pair: two launches {t(two_b):6.1f} us, one launch {t(pair_b):6.1f} us")
