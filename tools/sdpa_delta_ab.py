"""The decoder's self-attention backward (causal, T <= 64) against an fp64 reference of the UNROUNDED inputs, in the regime where
dP - delta cancels (rows of V = a common vector + VN x noise): 1 - cos of dQ / dK / dV and the kernel's time.
ASR_HIP_LIB=<previous library> python tools/sdpa_delta_ab.py   for the other side of the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
B, H, dk = 32, 8, 64
d = H * dk
def cosd(a, b): return 1 - float(torch.nn.functional.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0))
for T in (17, 64):
    for VN in (1.0, 0.1, 0.03):
        g = torch.Generator(device="cuda").manual_seed(T)
        q = torch.randn(B * T, d, device="cuda", generator=g, dtype=torch.float64)
        k = torch.randn(B * T, d, device="cuda", generator=g, dtype=torch.float64)
        v = torch.randn(1, d, device="cuda", generator=g, dtype=torch.float64) + VN * torch.randn(B * T, d, device="cuda", generator=g, dtype=torch.float64)
        do = torch.randn(B * T, d, device="cuda", generator=g, dtype=torch.float64)
        qr, kr, vr = (x.clone().requires_grad_() for x in (q, k, v))
        def heads(x): return x.view(B, T, H, dk).transpose(1, 2)
        s = heads(qr) @ heads(kr).transpose(-1, -2) / 8.0
        s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device="cuda")), float("-inf"))
        o_ref = (torch.softmax(s, -1) @ heads(vr)).transpose(1, 2).reshape(B * T, d)
        o_ref.backward(do)
        qb, kb, vb, dob = (x.bfloat16().contiguous() for x in (q, k, v, do))
        klen = torch.full((B,), T, dtype=torch.int32, device="cuda")
        o, lse = K.sdpa_fwd(qb, kb, vb, klen, B, H, T, T, dk, True, -1)
        dq, dkk, dv = (torch.empty_like(qb) for _ in range(3))
        run = lambda: K.sdpa_bwd(qb, kb, vb, o, dob, lse, klen, B, H, T, T, dk, dq, dkk, dv, True, -1)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        print(f"T={T:2d} V noise {VN:4.2f}: 1-cos dQ {cosd(dq, qr.grad):.2e}  dK {cosd(dkk, kr.grad):.2e}  dV {cosd(dv, vr.grad):.2e}   {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
