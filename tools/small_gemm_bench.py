"""Micro-benchmark of the small-M projection kernel (decoder shapes: M = B*To ~ 544 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
M = int(os.environ.get("M", "544"))
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N, Kd, tb in ((1536, 512, False), (512, 512, False), (1024, 512, False), (512, 1024, False), (512, 1536, True), (512, 512, True), (512, 1024, True), (1024, 512, True)):
    a = torch.randn(M, Kd, device="cuda").bfloat16()
    b = (torch.randn(Kd, N, device="cuda") * 0.05).bfloat16() if tb else (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    bias = None if tb else torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: K.gemm_small(a, b, bias, out, trans_b=tb))
    tl = timeit(lambda: torch.mm(a, b if tb else b.t(), out=out))
    print(f"N={N:5d} K={Kd:5d} trans_b={int(tb)}  mine {t:6.1f} us   lib {tl:6.1f} us")
