"""Where the bf16 long-form CTC loss differs from the fp64 oracle (BASELINE configs[4] shape of tests/test_model_gpu.py::
test_long_form_window_matches_oracle: T = 2000, +-50 band, V = 56): the encoder's bf16 rounding, the head's bf16 logits, or the loss kernels?
python tools/ctc_parity_diag.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import ref_model as R
from tests.test_model_gpu import oracle_case, build, to_pack
from asr_chinese_e2e_amd import kernels as K

over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, attn_window=50)
B, T, Fd, V, L = 2, 2000, 80, 56, 20
cfg, sd, batch = oracle_case(B, T, Fd, V, L, over, seed=13)
sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
sd64 = {k: v.double() for k, v in sd.items()}
b64 = dict(batch, wave=batch["wave"].double())
ref = R.RefTrainer(sd64, cfg, warmup=25).iterate(b64)
print("oracle fp64: loss", float(ref["loss"]), "ctc", float(ref["out"]["ctc"]))
enc_ref = ref["out"].get("enc_out") if isinstance(ref["out"], dict) else None


def ctc64(logits, wave_len, tgt, tgt_len):
    lp = F.log_softmax(logits.double(), -1).transpose(0, 1)
    return F.ctc_loss(lp, tgt, wave_len, tgt_len, blank=0, reduction="none", zero_infinity=False)


W, bvec = sd64["ctc_lo.weight"], sd64["ctc_lo.bias"]
for dtype in ("fp32", "bf16"):
    model = build(cfg, V, "TransformerOffical", dtype=dtype).cuda()
    model.load_state_dict(sd)
    pack = to_pack(batch)
    eng = model._ensure_engine("cuda")
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    with torch.no_grad():
        out = model.forward(pack)
    enc = out.encoder_out.double().cpu()                      # (B, T, d) as the kernels produced it
    logits_k = out.ctc_logits.double().cpu()                   # the head's logits as stored (bf16 in bf16 mode)
    wl, tl = batch["wave_len"], batch["tgt_len"]
    tgt = batch["tgt_for_input"]
    nll_a = ctc64(enc @ W.t() + bvec, wl, tgt, tl)             # kernels' encoder output, fp64 head + fp64 loss
    nll_b = ctc64(logits_k, wl, tgt, tl)                       # kernels' stored logits, fp64 loss
    ctc_k = float(loss[2])
    refc = float(ref["out"]["ctc"])
    print(f"{dtype}: kernel ctc {ctc_k:.6f} rel {abs(ctc_k - refc) / refc:.3e} | fp64 head+loss on the kernels' encoder output {float(nll_a.mean()):.6f} rel {abs(float(nll_a.mean()) - refc) / refc:.3e}"
          f" | fp64 loss on the kernels' logits {float(nll_b.mean()):.6f} rel {abs(float(nll_b.mean()) - refc) / refc:.3e}")
    if dtype == "bf16":
        lg64 = enc @ W.t() + bvec
        d = (logits_k - lg64)
        print(f"   stored-logit rounding: rms {float(d.pow(2).mean().sqrt()):.3e}, max {float(d.abs().max()):.3e}; logits rms {float(lg64.pow(2).mean().sqrt()):.3f} max {float(lg64.abs().max()):.3f}")

# ---- where inside the encoder: the oracle's encoder output / logits in fp64 against the bf16 kernels', and the same with the INPUT features
# rounded to bf16 first (the kernels' input cast), and with the oracle's activations rounded to bf16 after every op it can be told to round at
with torch.no_grad():
    enc64 = R.encoder_forward(sd64, cfg, b64["wave"], batch["wave_len"])[0] if isinstance(R.encoder_forward(sd64, cfg, b64["wave"], batch["wave_len"]), tuple) else R.encoder_forward(sd64, cfg, b64["wave"], batch["wave_len"])
    wave_r = batch["wave"].bfloat16().double()
    enc64_r = R.encoder_forward(sd64, cfg, wave_r, batch["wave_len"])
    enc64_r = enc64_r[0] if isinstance(enc64_r, tuple) else enc64_r
enc64 = enc64.reshape(B, T, -1)
enc64_r = enc64_r.reshape(B, T, -1)
lg_ref = enc64 @ W.t() + bvec
print(f"oracle with bf16-rounded INPUT features: ctc {float(ctc64(enc64_r @ W.t() + bvec, wl, tgt, tl).mean()):.6f} (rel {abs(float(ctc64(enc64_r @ W.t() + bvec, wl, tgt, tl).mean()) - refc) / refc:.3e})")
m = (torch.arange(T)[None, :] < wl[:, None])
de = (enc - enc64)[m]
print(f"bf16 encoder output vs oracle: rms error {float(de.pow(2).mean().sqrt()):.4e} (output rms {float(enc64[m].pow(2).mean().sqrt()):.3f}), mean error {float(de.mean()):.3e}")
dl = (logits_k - lg_ref)[m]
print(f"bf16 logits vs oracle: rms error {float(dl.pow(2).mean().sqrt()):.4e}, mean {float(dl.mean()):.3e}; per-frame mean of (error of lse - error of blank logit) follows")
lse_k, lse_r = torch.logsumexp(logits_k, -1)[m], torch.logsumexp(lg_ref, -1)[m]
print(f"   lse error mean {float((lse_k - lse_r).mean()):.4e}  blank-logit error mean {float((logits_k[..., 0] - lg_ref[..., 0])[m].mean()):.4e}")
