import os, sys
sys.path.insert(0, "/root/repo")
import torch
from asr_chinese_e2e_amd import kernels as K
B, H, dk = 32, 8, 64
d = H * dk
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for Tq, Tk in ((32, 500), (64, 500), (128, 500), (256, 500), (500, 500), (500, 64), (500, 128), (500,256)):
    q = torch.randn(B * Tq, d, device="cuda").bfloat16()
    kv = torch.randn(B * Tk, 2 * d, device="cuda").bfloat16()
    klen = torch.full((B,), Tk, dtype=torch.int32, device="cuda")
    o, lse = K.sdpa_fwd(q, kv[:, :d], kv[:, d:], klen, B, H, Tq, Tk, dk, False, -1)
    t = timeit(lambda: K.sdpa_fwd(q, kv[:, :d], kv[:, d:], klen, B, H, Tq, Tk, dk, False, -1, o=o, lse=lse))
    print(f"Tq {Tq:4d} Tk {Tk:4d}  {t:6.1f} us")
