import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
B, T, d = 32, 500, 512
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
x = torch.randn(B * T, d, device="cuda").bfloat16(); res = torch.randn_like(x); dy = torch.randn_like(x); dy2 = torch.randn_like(x)
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda"); lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
ws = K.Workspace("cuda")
y, xhat, rstd = K.add_ln_fwd(x, res, g, b, None, lens, B, T)
dg, db, dbias = (torch.zeros(d, device="cuda") for _ in range(3))
dz = torch.empty_like(x)
t = timeit(lambda: K.add_ln_fwd(x, res, g, b, None, lens, B, T, y=y, xhat=xhat, rstd=rstd))
print(f"ln fwd {t:6.1f} us {4 * x.numel() * 2 / t / 1e6:5.2f} TB/s")
t = timeit(lambda: K.add_ln_bwd(dy, dy2, xhat, rstd, g, lens, dg, db, dbias, B, T, ws, dz=dz))
print(f"ln bwd (+finalize) {t:6.1f} us {4 * x.numel() * 2 / t / 1e6:5.2f} TB/s")
a = torch.randn(B * T, 1024, device="cuda").bfloat16(); da = torch.randn_like(a); dbb = torch.zeros(1024, device="cuda")
t = timeit(lambda: K.relu_bwd_(da, a, dbb, ws))
print(f"relu_bwd+colsum {t:6.1f} us {3 * a.numel() * 2 / t / 1e6:5.2f} TB/s")
q = torch.randn(B * T, 1536, device="cuda").bfloat16(); o = torch.zeros(1536, device="cuda")
t = timeit(lambda: K.colsum(q, o, ws))
print(f"colsum 1536 {t:6.1f} us {q.numel() * 2 / t / 1e6:5.2f} TB/s")
