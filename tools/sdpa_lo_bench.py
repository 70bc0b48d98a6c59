"""Attention forward / backward kernel times at the decoder's shapes (cross: 17 x 500, self: 17 x 17 causal) and the encoder's (500 x 500), with and
without a low-order output piece (asr_sdpa_fwd's o_lo; only heads of more than 512 keys use it).  ASR_HIP_LIB=<other build> for an A/B in one call."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from asr_chinese_e2e_amd import kernels as K
def timeit(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, H, dk = 32, 8, 64; d = H * dk
for name, Tq, Tk, causal in (("cross", 17, 500, False), ("self", 17, 17, True), ("encoder", 500, 500, False)):
    q = torch.randn(B * Tq, d, device="cuda").bfloat16(); k = torch.randn(B * Tk, d, device="cuda").bfloat16(); v = torch.randn(B * Tk, d, device="cuda").bfloat16()
    klen = torch.full((B,), Tk, dtype=torch.int32, device="cuda"); do = torch.randn_like(q)
    for lo in (False, True):
        o_lo = torch.empty_like(q) if lo else None
        o, lse = K.sdpa_fwd(q, k, v, klen, B, H, Tq, Tk, dk, causal, -1, o_lo=o_lo)
        dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        tf = timeit(lambda: K.sdpa_fwd(q, k, v, klen, B, H, Tq, Tk, dk, causal, -1, o=o, lse=lse, o_lo=o_lo))
        tb = timeit(lambda: K.sdpa_bwd(q, k, v, o, do, lse, klen, B, H, Tq, Tk, dk, dq, dk_, dv, causal, -1, o_lo=o_lo))
        print(f"{name:8s} lo={int(lo)}: fwd {tf:6.1f} us  bwd {tb:6.1f} us")
