"""Which kernel transitions leave the queue idle?  Run under `rocprofv3 --kernel-trace`, then tools/gap_probe_report.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
M, N, Kd = 16000, 512, 512
a = torch.randn(M, N, device="cuda").bfloat16()
w = torch.randn(N, Kd, device="cuda").bfloat16()
b = torch.randn(N, device="cuda")
x = torch.randn(M, Kd, device="cuda").bfloat16()
o1 = torch.empty(M, Kd, device="cuda", dtype=torch.bfloat16)
o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
marker = torch.zeros(1 << 20, device="cuda")
def sep():
    torch.cuda.synchronize(); marker.add_(1.0); marker.add_(1.0); marker.add_(1.0); torch.cuda.synchronize()
for rep in range(2):
    sep()
    for _ in range(40):      # phase A: own NT GEMM -> library mm
        K.gemm_nt(x, w, b, o2)
        torch.mm(a, w, out=o1)
    sep()
    for _ in range(40):      # phase B: torch elementwise -> library mm
        o2.add_(1.0)
        torch.mm(a, w, out=o1)
    sep()
    for _ in range(40):      # phase C: library mm -> library mm
        torch.mm(a, w, out=o1)
        torch.mm(a, w, out=o1)
    sep()
    for _ in range(40):      # phase D: own -> own
        K.gemm_nt(x, w, b, o2)
        K.relu_(o2)
torch.cuda.synchronize()
