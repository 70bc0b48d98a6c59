import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
B, T, V, L = 32, int(os.environ.get("T", "500")), 4232, 22
torch.manual_seed(0)
x = (torch.randn(B, T, V, device="cuda") * 2).bfloat16()
lab = torch.randint(4, V, (B, L), dtype=torch.int32, device="cuda")
ll = torch.randint(8, L + 1, (B,), dtype=torch.int32, device="cuda")
il = torch.full((B,), T, dtype=torch.int32, device="cuda")
ws = K.Workspace("cuda"); dl = torch.empty_like(x)
for _ in range(3): K.ctc_fwd_bwd(x, il, lab, ll, ws, dlogits=dl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.ctc_fwd_bwd(x, il, lab, ll, ws, dlogits=dl)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"ctc fwd+bwd (3 launches) {t:.1f} us  {3 * x.numel() * 2 / t / 1e6:.2f} TB/s algorithmic")
