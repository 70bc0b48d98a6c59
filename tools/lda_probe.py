"""Does the row pitch of the A operand matter to the persistent NT GEMM?  (L2 channel mapping of the 128-byte row pieces a k-step
fetches: rows of 1024 bytes put every 4th row piece on the same channel if channels interleave at 256 bytes.)
python tools/lda_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
M = 16000
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N, Kd, name in [(512, 512, "fc"), (1536, 512, "qkv"), (512, 1024, "w2"), (1024, 512, "w1")]:
    w = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    line = f"{name:4s} N={N:5d} K={Kd:5d}:"
    for pad in (0, 8, 32, 64, 128, 192, 320):
        buf = torch.randn(M, Kd + pad, device="cuda").bfloat16()
        a = buf[:, :Kd]
        for cpad in (0, 64):
            obuf = torch.empty(M, N + cpad, device="cuda", dtype=torch.bfloat16)
            out = obuf[:, :N]
            t = timeit(lambda: K.gemm_nt(a, w, None, out))
            line += f"  lda+{pad}/ldc+{cpad} {t:5.1f}"
    print(line, flush=True)
