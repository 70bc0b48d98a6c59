"""Host cost of one kernel launch through the Python wrappers vs the bare ctypes call (us per call, GPU queue kept short)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K, _lib
lib = _lib.lib
x = torch.randn(64, 512, device="cuda").bfloat16(); w = torch.randn(512, 512, device="cuda").bfloat16(); b = torch.randn(512, device="cuda")
out = torch.empty(64, 512, device="cuda", dtype=torch.bfloat16)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % 200 == 199: torch.cuda.synchronize()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("gemm_small wrapper      %.2f us" % t(lambda: K.gemm_small(x, w, b, out)))
args = (K._p(x), K._p(w), K._p(b), None, K._p(out), 64, 512, 512, 512, 512, 512, 0, 0)
print("bare ctypes call         %.2f us" % t(lambda: lib.asr_gemm_small_bf16(*args, K._stream())))
fast = _lib.fast
print("fastcall, fixed stream   %.2f us" % t(lambda: fast.asr_gemm_small_bf16(*args, K._stream())))
print("fastcall asr_abi_version %.2f us (no launch: the trampoline alone)" % t(lambda: fast.asr_abi_version()))
st = K._stream()
print("bare ctypes, fixed strm  %.2f us" % t(lambda: lib.asr_gemm_small_bf16(*args, st)))
print("_stream()                %.2f us" % t(lambda: K._stream()))
print("_p(x) x5                 %.2f us" % t(lambda: (K._p(x), K._p(w), K._p(b), K._p(out), K._p(None))))
print("torch.empty              %.2f us" % t(lambda: torch.empty(64, 512, device="cuda", dtype=torch.bfloat16)))
print("torch.mm                 %.2f us" % t(lambda: torch.mm(x, w, out=out)))
g, bt = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
print("add_ln_fwd wrapper       %.2f us" % t(lambda: K.add_ln_fwd(x, None, g, bt, None, None, 1, 64, y=out, xhat=x)))
ev = torch.cuda.Event()
s2 = torch.cuda.Stream()
print("event record+wait        %.2f us" % t(lambda: (ev.record(), s2.wait_event(ev))))
def ctx():
    with torch.cuda.stream(s2): pass
print("stream context           %.2f us" % t(ctx))
