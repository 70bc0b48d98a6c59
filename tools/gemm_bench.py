"""Micro-benchmark of the hand-written MFMA GEMMs against the library GEMM on the config-2 shapes.
python tools/gemm_bench.py [nt|tn|nn|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
from asr_chinese_e2e_amd._lib import ACT_RELU, ACT_RELU_MASK

which = sys.argv[1] if len(sys.argv) > 1 else "all"
M = int(os.environ.get("BENCH_M", "16000"))
SHAPES = [(1536, 512, "qkv"), (512, 512, "fc"), (1024, 512, "w1"), (512, 1024, "w2"), (4232, 512, "ctc_lo")]
if which == "nt":
    SHAPES = SHAPES + [(512, 4232, "ctc_dx"), (512, 1536, "qkv_dx")]      # input gradients as NT products with transposed weight copies


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
    return e0.elapsed_time(e1) / reps * 1e3  # us


torch.manual_seed(0)
for N, Kd, name in (SHAPES if which not in ("grp", "grp1", "pad", "mask") else []):
    x = torch.randn(M, Kd, device="cuda").bfloat16()
    w = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(N, Kd, device="cuda")
    dx = torch.empty(M, Kd, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * Kd
    line = f"{name:7s} N={N:5d} K={Kd:5d}"
    if which in ("nt", "all"):
        t = timeit(lambda: K.gemm_nt(x, w, b, out))
        tl = timeit(lambda: torch.addmm(b.bfloat16(), x, w.t(), out=out))
        line += f" | NT mine {t:7.1f} us {fl / t / 1e6:6.0f} TF/s  lib {tl:7.1f} us {fl / tl / 1e6:6.0f} TF/s"
    if which in ("tn", "all"):
        t = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True))
        tl = timeit(lambda: torch.mm(dy.t(), x))
        line += f" | TN mine {t:7.1f} us {fl / t / 1e6:6.0f} TF/s  lib {tl:7.1f} us {fl / tl / 1e6:6.0f} TF/s"
    if which in ("nn", "all"):
        tl = timeit(lambda: torch.mm(dy, w, out=dx))
        line += f" | NN lib {tl:7.1f} us {fl / tl / 1e6:6.0f} TF/s"
        if hasattr(K, "gemm_nn"):
            t = timeit(lambda: K.gemm_nn(dy, w, dx))
            line += f"  mine {t:7.1f} us {fl / t / 1e6:6.0f} TF/s"
    print(line, flush=True)

if which == "mask":
    # the w_2 input gradient: dH = (dY W_2) masked by the ReLU of the w_1 activations (N = 1024 outputs, reduction 512), against the
    # same product without the mask
    x = torch.randn(M, 512, device="cuda").bfloat16()
    wt = (torch.randn(1024, 512, device="cuda") * 0.05).bfloat16()
    act = torch.randn(M, 1024, device="cuda").bfloat16()
    out = torch.empty(M, 1024, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * 1024 * 512
    t0 = timeit(lambda: K.gemm_nt(x, wt, None, out))
    t1 = timeit(lambda: K.gemm_nt(x, wt, None, out, act=ACT_RELU_MASK, res=act))
    t2 = timeit(lambda: K.gemm_nt(x, wt, None, out, res=act))
    b = torch.randn(1024, device="cuda")
    t3 = timeit(lambda: K.gemm_nt(x, wt, b, out, act=ACT_RELU))
    print(f"w2_dx plain {t0:6.1f} us {fl / t0 / 1e6:5.0f} TF/s | ReLU mask {t1:6.1f} us {fl / t1 / 1e6:5.0f} TF/s | residual add {t2:6.1f} us | w1 fwd bias+ReLU {t3:6.1f} us", flush=True)

if which == "pad":
    # the CTC head (V = 4232: rows of 8464 B start 16 B further into a 128-B line each) with rows padded to 4288 columns (8576 B = 67 lines)
    V, D = 4232, 512
    for ld in (4232, 4288, 4352):
        x = torch.randn(M, D, device="cuda").bfloat16()
        w = (torch.randn(V, D, device="cuda") * 0.05).bfloat16()
        b = torch.randn(V, device="cuda")
        buf = torch.randn(M, ld, device="cuda").bfloat16()
        out = buf[:, :V]
        wt = (torch.randn(D, ld, device="cuda") * 0.05).bfloat16()[:, :V]      # transposed weight copy, padded the same way
        dx = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(V, D, device="cuda"); db = torch.zeros(V, device="cuda")
        fl = 2.0 * M * V * D
        t_lo = timeit(lambda: K.gemm_nt(x, w, b, out))
        t_dx = timeit(lambda: K.gemm_nt(out, wt, None, dx))
        t_dw = timeit(lambda: K.gemm_tn(out, x, dw, accumulate=True, dbias=db))
        print(f"ld={ld}: ctc_lo {t_lo:6.1f} us {fl / t_lo / 1e6:5.0f} TF/s | ctc_dx {t_dx:6.1f} us {fl / t_dx / 1e6:5.0f} TF/s | ctc_dw {t_dw:6.1f} us {fl / t_dw / 1e6:5.0f} TF/s", flush=True)

if which in ("grp", "all"):
    # one encoder layer's four weight gradients: four launches vs one grouped launch
    probs = []
    for N, Kd, name in SHAPES[:4]:
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, Kd, device="cuda").bfloat16()
        probs.append((dy, x, torch.zeros(N, Kd, device="cuda"), torch.zeros(N, device="cuda") if name in ("qkv", "w1") else None))
    fl = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in probs)
    t1 = timeit(lambda: [K.gemm_tn(dy, x, dw, accumulate=True, dbias=db) for dy, x, dw, db in probs])
    tg = timeit(lambda: K.gemm_tn_grouped(probs, accumulate=True))
    print(f"layer wgrads: 4 launches {t1:7.1f} us {fl / t1 / 1e6:6.0f} TF/s | grouped {tg:7.1f} us {fl / tg / 1e6:6.0f} TF/s", flush=True)
    N, Kd, _ = SHAPES[4]
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, Kd, device="cuda").bfloat16()
    dw = torch.zeros(N, Kd, device="cuda"); db = torch.zeros(N, device="cuda")
    fl = 2.0 * M * N * Kd
    t1 = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, dbias=db))
    tg = timeit(lambda: K.gemm_tn_grouped([(dy, x, dw, db)], accumulate=True))
    print(f"ctc_lo wgrad: single {t1:7.1f} us {fl / t1 / 1e6:6.0f} TF/s | grouped kernel {tg:7.1f} us {fl / tg / 1e6:6.0f} TF/s", flush=True)

if which == "grp1":
    # every projection alone: per-projection kernel (128 x 128 tiles) vs the grouped kernel's 256 x 128 tiles with one problem
    for N, Kd, name in SHAPES:
        dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, Kd, device="cuda").bfloat16()
        dw = torch.zeros(N, Kd, device="cuda"); db = torch.zeros(N, device="cuda")
        fl = 2.0 * M * N * Kd
        t1 = timeit(lambda: K.gemm_tn(dy, x, dw, accumulate=True, dbias=db))
        tg = timeit(lambda: K.gemm_tn_grouped([(dy, x, dw, db)], accumulate=True))
        print(f"{name:7s} single {t1:7.1f} us {fl / t1 / 1e6:6.0f} TF/s | 256x128 kernel {tg:7.1f} us {fl / tg / 1e6:6.0f} TF/s", flush=True)
