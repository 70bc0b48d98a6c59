"""In-kernel time stamps of the persistent NT GEMM (diagnostic build path ASR_GEMM_DBG=5, ASR_GEMM_CFG=4):
per workgroup: start, barrier passage of the first 12 k-steps, end of first tile's store tail issue, stores drained.
s_memrealtime ticks are 10 ns."""
import os, sys
os.environ.setdefault("ASR_GEMM_DBG", "5")
os.environ.setdefault("ASR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asr_chinese_e2e_amd", "libasr_hip_dbg.so"))      # `make -C asr_chinese_e2e_amd/csrc debug`
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
M = 16000
for N, Kd, name in [(512, 512, "fc"), (1536, 512, "qkv"), (1024, 512, "w1"), (512, 1024, "w2"), (4232, 512, "ctc_lo")]:
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(256 * 16 * 2, device="cuda", dtype=torch.float32)   # 256 x 16 uint64
    for _ in range(5):
        K.gemm_nt(x, w, st, out)
    torch.cuda.synchronize()
    t = st.view(torch.int64).view(256, 16).cpu()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t - t0).double() * 0.01  # us
    print(name, "blocks", t.shape[0])
    print("  start       med %.2f max %.2f us" % (rel[:, 0].median(), rel[:, 0].max()))
    nk = Kd // 64
    for i in range(min(nk, 12)):
        d = (t[:, 1 + i] - t[:, 0]).double() * 0.01
        print("  k-step %2d passed barrier at +%.2f us (med), +%.2f (max)" % (i, d.median(), d.max()))
    d = (t[:, 13] - t[:, 0]).double() * 0.01; print("  tile-0 stores issued   +%.2f med" % d.median())
    d = (t[:, 14] - t[:, 0]).double() * 0.01; print("  tile-0 stores drained  +%.2f med" % d.median())
    print("  last block finished at %.2f us after first start" % rel[:, 14].max())
