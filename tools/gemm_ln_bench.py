"""Fused projection + residual + LayerNorm against the two-kernel path (config-2 shapes: M = 16000, N = 512, K = 512 / 1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K

B, T, N = 32, 500, 512
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for Kd in (64, 128, 512, 1024):
    a = torch.randn(B * T, Kd, device="cuda").bfloat16()
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(B * T, N, device="cuda").bfloat16()
    g, bt = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    c = torch.empty(B * T, N, device="cuda", dtype=torch.bfloat16)
    y, xh, rs = torch.empty_like(res), torch.empty_like(res), torch.empty(B * T, device="cuda")
    t_f = timeit(lambda: K.gemm_nt_add_ln(a, w, bias, res, g, bt, lens, B, T, y=y, xhat=xh, rstd=rs))
    t_g = timeit(lambda: K.gemm_nt(a, w, bias, c))
    t_l = timeit(lambda: K.add_ln_fwd(c, res, g, bt, None, lens, B, T, y=y, xhat=c, rstd=rs))
    fl = 2.0 * B * T * N * Kd
    print(f"K={Kd}: fused {t_f:6.1f} us ({fl / t_f / 1e6:5.0f} TF/s)   gemm {t_g:6.1f} + add_ln {t_l:6.1f} = {t_g + t_l:6.1f} us")
