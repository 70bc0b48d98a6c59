"""Cost of a kernel boundary: back-to-back launches of a tiny kernel on one stream (GPU-side time per launch)
and of a mid-size streaming kernel, against their kernel-trace durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K, _lib
lib = _lib.fast
def run(n_elems, reps=400):
    x = torch.randn(n_elems, device="cuda").bfloat16()
    st = K._stream()
    for _ in range(20): lib.asr_relu_fwd(x.data_ptr(), n_elems, 1, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): lib.asr_relu_fwd(x.data_ptr(), n_elems, 1, st)
    th = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    print(f"relu on {n_elems * 2 / 1e6:8.3f} MB: GPU {e0.elapsed_time(e1) / reps * 1e3:6.2f} us per launch, host {th / reps * 1e6:5.2f} us per launch")
for n in (1024, 1 << 20, 8 << 20, 32 << 20):
    run(n)
