"""Who is right where the fp32 HIP path and the fp32 oracle disagree (attention Q / K weights scaled by SHARP: far-from-uniform softmax rows): both against
the oracle in fp64.  FULL=1 SHARP=3 python tools/sharp_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_model as R
from tests.test_model_gpu import build, oracle_case, to_pack, cos
FULL = os.environ.get("FULL", "0") == "1"
over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=6 if FULL else 2, ctc_weight=0.3)
cfg, sd, batch = oracle_case(32, 500, 80, 4232, 17, over, seed=13) if FULL else oracle_case(4, 136, 80, 56, 12, over, seed=9)
V = 4232 if FULL else 56
sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
f = float(os.environ.get("SHARP", "3"))
for k in sd:
    if "w_qs.weight" in k or "w_ks.weight" in k:
        sd[k] = sd[k] * f
ref32 = R.RefTrainer(sd, cfg, warmup=25).iterate(batch)
ref64 = R.RefTrainer({k: v.double() for k, v in sd.items()}, cfg, warmup=25).iterate(dict(batch, wave=batch["wave"].double()))
print("loss: oracle fp32", float(ref32["loss"]), " oracle fp64", float(ref64["loss"]))
for dtype in os.environ.get("DTYPES", "fp32,bf16").split(","):
    model = build(cfg, V, "TransformerOffical", dtype=dtype).cuda()
    model.load_state_dict(sd); model._ensure_engine("cuda"); model.zero_flat_grads()
    loss, _ = model.train_step(to_pack(batch))
    print(f"{dtype}: loss {float(loss[0])}")
    gmax = max(float(g.abs().max()) for g in ref64["grads"].values())
    rows = []
    for n, p in model.named_parameters():
        g64 = ref64["grads"][n].float()
        if n.endswith("w_ks.bias") or float(g64.abs().max()) < 1e-5 * gmax: continue
        rows.append((cos(p.grad, g64), cos(ref32["grads"][n], g64), n, float(g64.abs().max()) / gmax))
    for c_ours, c_o32, n, m in sorted(rows)[:8]:
        print(f"  {n:50s} HIP {dtype} vs fp64 oracle: {c_ours:.6f}   fp32 oracle vs fp64 oracle: {c_o32:.6f}   |g|max/gmax {m:.1e}")
