"""The long-form case of tests/test_model_gpu.py (T = 2000, +-50-frame band, joint, 2 layers) in bf16 against the fp64 oracle run on (a) the fp32 weights and
(b) the weight matrices rounded to bf16: how much of the loss difference is the parameters' rounding.  python tools/long_form_parity.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from oracle import ref_model as R
from tests.test_model_gpu import build, oracle_case, to_pack, cos
over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, attn_window=50)
B, T, F, V, L = 2, 2000, 80, 56, 20
cfg, sd, batch = oracle_case(B, T, F, V, L, over, seed=13)
sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
model = build(cfg, V, "TransformerOffical", dtype="bf16").cuda(); model.load_state_dict(sd)
model._ensure_engine("cuda"); model.zero_flat_grads()
loss, _ = model.train_step(to_pack(batch))
for name, rounded in (("fp32 weights", False), ("bf16-rounded weight matrices", True)):
    sdr = {k: ((v.bfloat16().float() if (rounded and v.dim() == 2) else v).double()) for k, v in sd.items()}
    ref = R.RefTrainer(sdr, cfg, warmup=25).iterate(dict(batch, wave=batch["wave"].double()))
    rel = abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"])); relc = abs(float(loss[2]) - float(ref["out"]["ctc"])) / abs(float(ref["out"]["ctc"]))
    worst = min((cos(p.grad, ref["grads"][n].float()), n) for n, p in model.named_parameters() if not n.endswith("w_ks.bias") and float(ref["grads"][n].abs().max()) > 1e-5 * max(float(g.abs().max()) for g in ref["grads"].values()))
    print(f"oracle on {name}: loss rel {rel:.2e}  ctc rel {relc:.2e}  worst significant cosine {worst[0]:.5f} ({worst[1]})")
