"""Per-tensor parity of the bf16 MFMA path on the d_model = 512 reference golden case (tests/golden/model_mfma_d512.npz): cosine on
the sampled elements against the reference's values, and over EVERY element against the oracle run on the same weights.
python tools/d512_parity.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ref_model as R
from tests.helpers import mfma_golden_case
from tests.test_model_gpu import build, to_pack, cos

cfg, sd, batch, z, GI = mfma_golden_case()
ref = R.RefTrainer(sd, cfg, warmup=25).iterate(batch)
for dtype in ("fp32", "bf16"):
    model = build(cfg, GI.MFMA_CASE["V"], dtype=dtype)
    model.load_state_dict(sd)
    model = model.cuda()
    model._ensure_engine("cuda")
    model.zero_flat_grads()
    loss, _ = model.train_step(to_pack(batch))
    print(dtype, "loss rel", abs(float(loss[0]) - float(z["fwd/loss"])) / abs(float(z["fwd/loss"])))
    for n, p in model.named_parameters():
        g = p.grad.double().flatten().cpu().numpy()
        w = z["grad_s/" + n].astype(np.float64)
        gs = g[GI.sample_index(n, g.size)]
        cs = float((gs @ w) / (np.linalg.norm(gs) * np.linalg.norm(w) + 1e-300))
        print(f"  {n:55s} sampled-vs-reference cos {cs:.6f}  full-vs-oracle cos {cos(p.grad, ref['grads'][n]):.6f}  norm ratio {float(np.linalg.norm(g)) / float(z['grad_norm/' + n]):.4f}")
