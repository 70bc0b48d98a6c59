"""Micro-benchmark of the fused residual-add + LayerNorm kernels at the config-2 shape (16000 x 512 bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
rows, d, T = int(os.environ.get("ROWS", "16000")), 512, 500
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
x = torch.randn(rows, d, device="cuda").bfloat16(); res = torch.randn_like(x)
g = torch.randn(d, device="cuda"); b = torch.randn(d, device="cuda")
B = rows // T
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
y, xhat, rstd = K.add_ln_fwd(x, res, g, b, None, lens, B, T)
t = timeit(lambda: K.add_ln_fwd(x, res, g, b, None, lens, B, T, y=y, xhat=xhat, rstd=rstd))
print(f"add_ln_fwd  {t:6.1f} us  {4 * rows * d * 2 / t / 1e6:5.2f} TB/s")
dy = torch.randn_like(x)
dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
ws = torch.empty(K.add_ln_bwd_workspace_bytes(rows, d), dtype=torch.uint8, device="cuda")
dz = torch.empty_like(dy)
t = timeit(lambda: K.add_ln_bwd(dy, None, xhat, rstd, g, lens, dg, db, None, B, T, ws, dz=dz, partials=ws))
print(f"add_ln_bwd (partials only)  {t:6.1f} us  {3 * rows * d * 2 / t / 1e6:5.2f} TB/s")
dy2 = torch.randn_like(x)
t = timeit(lambda: K.add_ln_bwd(dy, dy2, xhat, rstd, g, lens, dg, db, None, B, T, ws, dz=dz, partials=ws))
print(f"add_ln_bwd with the residual-path gradient (dy2)  {t:6.1f} us  {4 * rows * d * 2 / t / 1e6:5.2f} TB/s")
