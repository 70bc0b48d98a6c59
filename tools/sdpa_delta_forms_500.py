"""The attention backward at T = 500 keys with a component common to the rows of K and of V (a bias behind a LayerNorm), emulated in fp64 with the fused
kernel's rounding points: delta = rowsum(dO o O) from the bf16 O, from its own p and dP, from the fp32 O, from O as two bf16 pieces (what asr_sdpa_fwd's
o_lo stores) - 1 - cos(dQ) against the fp64 result of the unrounded inputs.  python tools/sdpa_delta_forms_500.py  (CPU, ~2 min)."""
import torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float64)
L2E = 1.4426950408889634
def run(T, sc, kmean, vmean, d=64, n=24):
    res = []
    for trial in range(n):
        q = torch.randn(T, d, dtype=torch.float64) * sc
        k = (torch.randn(T, d, dtype=torch.float64) + kmean * torch.randn(1, d, dtype=torch.float64)) * sc
        v = torch.randn(T, d, dtype=torch.float64) + vmean * torch.randn(1, d, dtype=torch.float64)
        do = torch.randn(T, d, dtype=torch.float64)
        def exact(q, k, v, do):
            s = (q @ k.T) / 8.0; p = torch.softmax(s, -1)
            dp = do @ v.T; dl = (p * dp).sum(-1, keepdim=True); ds = p * (dp - dl); return ds @ k / 8.0, ds.T @ q / 8.0, p, dp, torch.logsumexp(s, -1, keepdim=True)
        dq_ref, dk_ref = exact(q, k, v, do)[:2]
        qb, kb, vb, dob = bf(q), bf(k), bf(v), bf(do)
        dq_in, dk_in, p, dp, lse = exact(qb, kb, vb, dob)
        ob = bf(bf(p) @ vb)
        kimg = bf(kb / 8.0 * L2E)
        pb = torch.exp2(qb @ kimg.T - lse * L2E)
        dl_a = (dob * ob).sum(-1, keepdim=True)
        dl_c = (pb * dp).sum(-1, keepdim=True) / pb.sum(-1, keepdim=True)
        dl_f = (dob * (bf(p) @ vb)).sum(-1, keepdim=True)      # delta from the forward's fp32 accumulator (P rounded to bf16 for the P V product)
        o2 = bf(p) @ vb; ohi = bf(o2); olo = bf(o2 - ohi); dl_s = (dob * (ohi + olo)).sum(-1, keepdim=True)      # O as two bf16 pieces
        def grads(dl):
            ds = bf(pb * (dp - dl)); return (ds @ kimg) / L2E, ds.T @ qb / 8.0
        c = lambda a, r: 1 - float(torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0))
        ga, gc, gf, gs = grads(dl_a), grads(dl_c), grads(dl_f), grads(dl_s)
        res.append((c(dq_in, dq_ref), c(ga[0], dq_ref), c(gc[0], dq_ref), c(gf[0], dq_ref), c(gs[0], dq_ref), c(ga[1], dk_ref)))
    r = torch.tensor(res).mean(0)
    print(f"T={T} score scale {sc:4.2f} K mean {kmean:3.1f} V mean {vmean:3.1f}: 1-cos dQ: inputs only {r[0]:.1e} | flash delta (bf16 O) {r[1]:.1e} | own delta {r[2]:.1e} | delta from fp32 O {r[3]:.1e} | from O as hi + lo bf16 {r[4]:.1e}")
for sc in (0.3, 1.0):
    for km, vm in ((0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (3.0, 3.0)):
        run(500, sc, km, vm)
