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
N, Kd = 512, 512
for M in (256, 2048, 4096, 8192, 12288, 16000, 16384, 32768, 65536):
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: K.gemm_nt(x, w, b, out))
    blocks = ((M + 255) // 256) * ((N + 127) // 128)
    print(f"M={M:6d} blocks={blocks:5d} {t:7.1f} us {2.0*M*N*Kd/t/1e6:6.0f} TF/s", flush=True)
