import os, sys
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo/asr_chinese_e2e_amd") else sys.path.insert(0, os.getcwd())
import torch
from asr_chinese_e2e_amd import kernels as K
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, H, Tq, Tk, ragged) in [(32, 8, 500, 500, False), (32, 8, 500, 500, True), (32, 8, 17, 500, False), (8, 8, 333, 470, True)]:
    d = H * 64
    torch.manual_seed(1)
    q = torch.randn(B * Tq, d, device="cuda").bfloat16(); k = torch.randn(B * Tk, d, device="cuda").bfloat16(); v = torch.randn(B * Tk, d, device="cuda").bfloat16()
    klen = torch.full((B,), Tk, dtype=torch.int32, device="cuda")
    if ragged: klen = torch.randint(Tk // 3, Tk + 1, (B,), dtype=torch.int32, device="cuda")
    res = {}
    for mode in (0, 1):
        K.set_option("sdpa_pair", mode)
        o, lse = K.sdpa_fwd(q, k, v, klen, B, H, Tq, Tk, 64, False, -1)
        t = timeit(lambda: K.sdpa_fwd(q, k, v, klen, B, H, Tq, Tk, 64, False, -1, o=o, lse=lse))
        res[mode] = (o.float().clone(), lse.clone(), t)
    K.set_option("sdpa_pair", 1)
    do = (res[0][0] - res[1][0]).abs().max().item(); dl = (res[0][1] - res[1][1]).abs().max().item()
    print(f"B={B} Tq={Tq} Tk={Tk} ragged={ragged}: two-pass {res[0][2]:.1f} us, pair {res[1][2]:.1f} us; max |dO| {do:.3e} max |dlse| {dl:.3e}")
