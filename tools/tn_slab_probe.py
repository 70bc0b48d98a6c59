"""Weight-gradient GEMM on the config-2 shapes: fp32-atomic combine of the M-splits (default) against per-split slabs + the ordered
reduce kernel (deterministic mode), stand-alone.  With ASR_HIP_LIB=.../libasr_hip_dbg.so ASR_GEMM_TN_NOATOMIC=1 the first column is the
kernel WITHOUT its combine (wrong results, timing only).  python tools/tn_slab_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K

M = int(os.environ.get("BENCH_M", "16000"))
SHAPES = [(1536, 512, "qkv"), (512, 512, "fc"), (1024, 512, "w1"), (512, 1024, "w2"), (4232, 512, "ctc_lo")]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
ws = K.Workspace("cuda")
for N, Kd, name in SHAPES:
    x = torch.randn(M, Kd, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    dw = torch.zeros(N, Kd, device="cuda")
    db = torch.zeros(N, device="cuda")
    K.set_deterministic(False)
    ta = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True))
    tab = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, dbias=db))
    K.set_deterministic(True)
    td = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, ws=ws))
    tdb = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, dbias=db, ws=ws))
    K.set_deterministic(False)
    print(f"{name:7s} N={N:5d} K={Kd:5d} | atomics {ta:6.1f} us, with bias {tab:6.1f} | slabs + reduce {td:6.1f} us, with bias {tdb:6.1f}", flush=True)
