"""Weight gradient of linear_in (M = 16000 frames, N = 512, K = 80 mel bins) under different M-split counts (tuning option tn_split, percent of the
planned splits): python tools/tn_small_k.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M, N, Kd = 16000, 512, 80
dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, Kd, device="cuda").bfloat16()
dw = torch.zeros(N, Kd, device="cuda"); db = torch.zeros(N, device="cuda")
for pct in (100, 50, 25, 12, 6):
    K.set_option("tn_split", pct)
    t = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, dbias=db))
    print(f"tn_split {pct:3d} %: {t:6.1f} us", flush=True)
K.set_option("tn_split", 0)
