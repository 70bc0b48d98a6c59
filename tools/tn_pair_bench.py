"""Weight gradients of one encoder layer (config 2: M = 16000): four single launches against two pair launches (asr_gemm_tn_grouped_bf16 on the
128 x 128-tile code, option tn_multi = 1) and against the 256 x 128-tile grouped kernel (tn_multi = 0), stand-alone, us per layer.
python tools/tn_pair_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K

M = int(os.environ.get("BENCH_M", "16000"))
SH = {"qkv": (1536, 512), "fc": (512, 512), "w1": (1024, 512), "w2": (512, 1024)}


def timeit(fn, reps=40):
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
P = {}
for name, (N, Kd) in SH.items():
    P[name] = (torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, Kd, device="cuda").bfloat16(), torch.zeros(N, Kd, device="cuda"),
               torch.zeros(N, device="cuda") if name in ("qkv", "w1") else None)
single = {n: timeit(lambda n=n: K.gemm_tn(P[n][0], P[n][1], P[n][2], accumulate=True, dbias=P[n][3])) for n in P}
print("single launches:", " ".join(f"{n} {t:.1f}" for n, t in single.items()), f"| layer {sum(single.values()):.1f} us")
for mode in (1, 0):
    prev = K.set_option("tn_multi", mode)
    t_ffn = timeit(lambda: K.gemm_tn_grouped([P["w2"], P["w1"]], accumulate=True))
    t_att = timeit(lambda: K.gemm_tn_grouped([P["fc"], P["qkv"]], accumulate=True))
    t_all = timeit(lambda: K.gemm_tn_grouped([P["w2"], P["w1"], P["fc"], P["qkv"]], accumulate=True))
    K.set_option("tn_multi", prev)
    print(f"tn_multi={mode}: (w2, w1) {t_ffn:.1f}  (fc, qkv) {t_att:.1f} | layer as two pairs {t_ffn + t_att:.1f} us | all four in one launch {t_all:.1f} us")
