"""Three forms of delta in the attention backward, emulated in fp64 with the fused kernel's rounding points (bf16 Q, K, V, dO; O stored bf16;
K image = bf16(K * scale * log2 e); dS rounded to bf16): delta = rowsum(dO o O) from the stored O (the flash identity), sum_j p_j dP_j, and
that over sum_j p_j - against the fp64 result of the unrounded inputs.  python tools/sdpa_delta_forms.py  (CPU, ~1 min)."""
import torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float64)
L2E = 1.4426950408889634
def run(T, sc, VN, d=64, n=200):
    res = []
    for trial in range(n):
        q = torch.randn(T, d, dtype=torch.float64) * sc; k = torch.randn(T, d, dtype=torch.float64) * sc
        v = torch.randn(1, d, dtype=torch.float64) + VN * torch.randn(T, d, dtype=torch.float64)
        do = torch.randn(T, d, dtype=torch.float64)
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))
        def exact(q, k, v, do):
            s = (q @ k.T) / 8.0; s = s.masked_fill(~mask, -1e30); p = torch.softmax(s, -1)
            dp = do @ v.T; dl = (p * dp).sum(-1, keepdim=True); ds = p * (dp - dl); return ds @ k / 8.0, p, dp, p @ v, torch.logsumexp(s, -1, keepdim=True)
        dq_ref = exact(q, k, v, do)[0]
        qb, kb, vb, dob = bf(q), bf(k), bf(v), bf(do)
        dq_in, p, dp, o, lse = exact(qb, kb, vb, dob)
        ob = bf(p.to(torch.float32).to(torch.float64).to(torch.bfloat16).to(torch.float64) @ vb)      # forward: P rounded to bf16 for P V, O stored bf16
        kimg = bf(kb / 8.0 * L2E)                                   # backward's K image
        s2 = (qb @ kimg.T).masked_fill(~mask, -1e30)
        pb = torch.exp2(s2 - lse * L2E)                             # backward's p from the forward's lse
        dl_a = (dob * ob).sum(-1, keepdim=True)
        dl_b = (pb * dp).sum(-1, keepdim=True)
        dl_c = dl_b / pb.sum(-1, keepdim=True)
        def dq_of(dl):
            ds = bf(pb * (dp - dl)); return (ds @ kimg) / L2E       # dQ = dS K scale
        c = lambda a: 1 - float(torch.nn.functional.cosine_similarity(a.flatten(), dq_ref.flatten(), dim=0))
        res.append((c(dq_in), c(dq_of(dl_a)), c(dq_of(dl_b)), c(dq_of(dl_c))))
    r = torch.tensor(res).mean(0)
    print(f"V noise {VN:5.2f} T={T:3d} score scale {sc:4.2f}: 1-cos(dQ): bf16 inputs only {r[0]:.2e} | kernel now (delta from bf16 O) {r[1]:.2e} | delta = sum p dP {r[2]:.2e} | / sum p {r[3]:.2e}")
for VN in (1.0, 0.1, 0.03):
    for T in (12, 64):
        for sc in (0.05, 1.0):
            run(T, sc, VN)
