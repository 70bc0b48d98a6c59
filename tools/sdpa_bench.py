"""Micro-benchmark of the attention kernels at the config-2 shape (B=32, H=8, T=500, dk=64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K

B, H, T, dk = int(os.environ.get("B", "32")), 8, int(os.environ.get("T", "500")), 64      # long-form band: B=8 T=2000 WINDOW=50
window = int(os.environ.get("WINDOW", "-1"))
d = H * dk
def timeit(fn, reps=int(os.environ.get("REPS", "20"))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
klen = torch.full((B,), T, dtype=torch.int32, device="cuda")
DROP = float(os.environ.get("DROP", "0"))      # attention dropout
LO = os.environ.get("LO", "1") == "1"      # with the low-order piece of O (the training step's default)
o_lo = torch.empty(B * T, d, device="cuda", dtype=torch.bfloat16) if LO else None
o, lse = K.sdpa_fwd(q, k, v, klen, B, H, T, T, dk, False, window, drop_p=DROP, drop_seed=7, o_lo=o_lo)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
fl = 4.0 * B * H * T * T * dk
t = timeit(lambda: K.sdpa_fwd(q, k, v, klen, B, H, T, T, dk, False, window, o=o, lse=lse, drop_p=DROP, drop_seed=7, o_lo=o_lo))
print(f"fwd  {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s  {4 * B * T * d * 2 / t / 1e6:6.2f} TB/s algorithmic")
t = timeit(lambda: K.sdpa_bwd(q, k, v, o, do, lse, klen, B, H, T, T, dk, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], False, window, drop_p=DROP, drop_seed=7, o_lo=o_lo))
print(f"bwd  {t:7.1f} us  {2.5 * fl / t / 1e6:6.0f} TF/s (5 products: algorithmic)  {8 * B * T * d * 2 / t / 1e6:6.2f} TB/s algorithmic")
