"""The K projection's weight gradient dW_k = dK^T x under the attention backward's delta error, emulated in fp64 with the fused kernel's rounding points at
500 keys, with components common to the rows of Q / K, of V and of the layer input x: as computed, with the mean over the keys of dK removed (zero in
exact arithmetic), and with delta from the kernel's own p and dP.  python tools/sdpa_dk_mean.py  (CPU, ~2 min)."""
import torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float64)
L2E = 1.4426950408889634
def run(T, sc, kmean, vmean, xmean, d=64, n=16):
    res = []
    for trial in range(n):
        x = torch.randn(T, d, dtype=torch.float64) + xmean * torch.randn(1, d, dtype=torch.float64)      # the layer input the K projection reads
        q = (torch.randn(T, d, dtype=torch.float64) + kmean * torch.randn(1, d, dtype=torch.float64)) * sc
        k = (torch.randn(T, d, dtype=torch.float64) + kmean * torch.randn(1, d, dtype=torch.float64)) * sc
        v = torch.randn(T, d, dtype=torch.float64) + vmean * torch.randn(1, d, dtype=torch.float64)
        do = torch.randn(T, d, dtype=torch.float64)
        def exact(q, k, v, do):
            s = (q @ k.T) / 8.0; p = torch.softmax(s, -1)
            dp = do @ v.T; dl = (p * dp).sum(-1, keepdim=True); ds = p * (dp - dl); return ds.T @ q / 8.0, p, dp, torch.logsumexp(s, -1, keepdim=True)
        dk_ref = exact(q, k, v, do)[0]
        qb, kb, vb, dob = bf(q), bf(k), bf(v), bf(do)
        dk_in, p, dp, lse = exact(qb, kb, vb, dob)
        ob = bf(bf(p) @ vb)
        kimg = bf(kb / 8.0 * L2E)
        pb = torch.exp2(qb @ kimg.T - lse * L2E)
        dl_a = (dob * ob).sum(-1, keepdim=True)
        dl_c = (pb * dp).sum(-1, keepdim=True) / pb.sum(-1, keepdim=True)
        def dk_of(dl): return bf(pb * (dp - dl)).T @ qb / 8.0
        da, dc = dk_of(dl_a), dk_of(dl_c)
        dm = da - da.mean(0, keepdim=True)
        c = lambda a, r: 1 - float(torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0))
        W = lambda g: g.T @ x      # the K projection's weight gradient
        res.append((c(W(dk_in), W(dk_ref)), c(W(da), W(dk_ref)), c(W(dm), W(dk_ref)), c(W(dc), W(dk_ref)), c(da, dk_ref), c(dm, dk_ref)))
    r = torch.tensor(res).mean(0)
    print(f"T={T} common components K,Q {kmean:3.1f} V {vmean:3.1f} x {xmean:3.1f}: 1-cos dW_k: inputs only {r[0]:.1e} | flash delta {r[1]:.1e} | flash delta, mean over keys of dK removed {r[2]:.1e} | own delta {r[3]:.1e}   (dK itself: {r[4]:.1e} -> {r[5]:.1e})")
for km, vm, xm in ((0, 0, 0), (1, 1, 0), (1, 1, 1), (3, 3, 1), (1, 1, 3), (3, 3, 3)):
    run(500, 0.3, float(km), float(vm), float(xm))
