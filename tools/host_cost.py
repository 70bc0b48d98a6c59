"""HOST time per launch (the GPU drains behind; bursts of 300 launches, timed before the sync): wrappers vs bare calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K, _lib
fast, ct = _lib.fast, _lib.lib
x = torch.randn(544, 512, device="cuda").bfloat16(); w = torch.randn(512, 512, device="cuda").bfloat16(); b = torch.randn(512, device="cuda")
out = torch.empty(544, 512, device="cuda", dtype=torch.bfloat16)
g, bt = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
def t(fn, n=300, rounds=5):
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
        torch.cuda.synchronize()
    return best
args = (x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), 544, 512, 512, 512, 512, 512, 0, 0)
st = K._stream()
print("fastcall bare launch (gemm_small)   %.2f us" % t(lambda: fast.asr_gemm_small_bf16(*args, st)))
print("ctypes bare launch                  %.2f us" % t(lambda: ct.asr_gemm_small_bf16(*args, st)))
print("K.gemm_small wrapper                %.2f us" % t(lambda: K.gemm_small(x, w, b, out)))
print("K.add_ln_fwd wrapper                %.2f us" % t(lambda: K.add_ln_fwd(x, None, g, bt, None, None, 1, 544, y=out, xhat=x)))
print("torch.empty                         %.2f us" % t(lambda: torch.empty(544, 512, device="cuda", dtype=torch.bfloat16)))
print("K.stream_fork                       %.2f us" % t(lambda: K.stream_fork(st)))
