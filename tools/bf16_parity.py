"""Per-tensor parity of the bf16 MFMA path against the fp32 CPU oracle (loss, gradient cosine, norm ratio):
python tools/bf16_parity.py [joint|ctc_only]   (the numbers quoted in DESIGN.md section 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_model as R
from tests.test_model_gpu import build, oracle_case, to_pack, cos

mode = sys.argv[1] if len(sys.argv) > 1 else "joint"
FULL = os.environ.get("FULL", "0") == "1"      # BASELINE configs[2] / [1] at full size (B 32, T 500, V 4232, 6 layers): ~10 s of oracle on 16 host cores
over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=6 if FULL else 2)
over.update(dict(ctc_weight=0.3) if mode == "joint" else dict(use_decoder=False, ctc_weight=1.0))
cfg, sd, batch = oracle_case(32, 500, 80, 4232, 17, over, seed=13) if FULL else oracle_case(4, 136, 80, 56, 12, over, seed=9)
V = 4232 if FULL else 56
if "decoder.tgt_word_emb.weight" in sd:
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
# SHARP=f: the attention Q / K projection weights times f (sharper attention than at initialisation: softmax rows far from uniform)
if float(os.environ.get("SHARP", "1")) != 1.0:
    for k in sd:
        if ("w_qs.weight" in k or "w_ks.weight" in k):
            sd[k] = sd[k] * float(os.environ["SHARP"])
# ORACLE_BF16_WEIGHTS=1: the oracle runs on the weight MATRICES rounded to bf16 (what the MFMA path multiplies by) - what is left of the difference
# is then the rounding of activations and of the kernels' intermediates, not of the parameters
sd_ref = {k: (v.bfloat16().float() if (os.environ.get("ORACLE_BF16_WEIGHTS") == "1" and v.dim() == 2) else v) for k, v in sd.items()}
ref = R.RefTrainer(sd_ref, cfg, warmup=25).iterate(batch)
for dtype in os.environ.get("DTYPES", "fp32,bf16").split(","):
    model = build(cfg, V, "TransformerCTC" if mode == "ctc_only" else "TransformerOffical", dtype=dtype).cuda()
    model.load_state_dict(sd)
    model._ensure_engine("cuda")
    model.zero_flat_grads()
    loss, _ = model.train_step(to_pack(batch))
    print(dtype, "loss rel", abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"])))
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    for n, p in model.named_parameters():
        g = ref["grads"][n]
        print(f"  {n:55s} cos {cos(p.grad, g):.6f}  ratio {float(p.grad.double().norm().cpu() / (g.double().norm() + 1e-30)):.4f}  |g|max/gmax {float(g.abs().max()) / gmax:.2e}")
