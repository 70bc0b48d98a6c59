"""End-to-end parity of the HIP engine behind the reference's model API.

 * fp32 mode against the REFERENCE's own outputs (tests/golden/model_small_*.npz: logits, loss,
   every parameter gradient, parameters after one and two Noam+Adam steps), tolerance = fp32
   round-off of a different summation order (1e-4 relative on gradients).
 * bf16 mode (MFMA attention + MFMA GEMMs, d_model=512 / d_k=64) against the CPU oracle
   (oracle/ref_model.RefTrainer, fp32) on the same weights and batch: SURVEY 8(d) gates - loss <= 1e-3 relative,
   gradient cosine >= 0.999 per tensor (bf16 storage of activations); measured worst cases go to
   gpurun_out/bf16_parity.jsonl.
 * joint CTC/attention (lambda = 0.3) and CTC-only against the oracle (CTC is not in the reference).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as R  # noqa: E402
from asr_chinese_e2e_amd import engine as E  # noqa: E402
from tests.helpers import golden_model_case, mfma_golden_case  # noqa: E402

DEV = "cuda"


def build(cfg, V, cls_name="TransformerOffical", **over):
    from asr_chinese_e2e_amd import Models
    from asr_chinese_e2e_amd.data_handler import Vocab
    M = getattr(Models, cls_name)
    mc = M.get_default_config()()
    d = dict(vars(cfg))
    d.pop("use_decoder", None)
    d.update(over)
    mc.fn_build(d)
    return M(mc, Vocab.synthetic(V))


def make_opt(model, cfg, warmup):
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    adam = FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9)   # main.py:81
    return NoamOpt(cfg.d_model, 1, warmup, adam)                                  # main.py:83


def to_pack(batch, dev=DEV):
    from asr_chinese_e2e_amd.Utils import Pack
    p = Pack()
    p.add(**{k: v.to(dev) for k, v in batch.items()})
    if "tgt_for_metric" not in p:
        p.add(tgt_for_metric=p.tgt_for_input.clone())
    return p


def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("case", ["model_small_ragged.npz", "model_small_full.npz"])
def test_fp32_matches_reference_golden(case):
    cfg, sd, batch, z = golden_model_case(case)
    V = int(z["cfg/V"])
    model = build(cfg, V, dtype="fp32")
    model.load_state_dict(sd)
    model = model.cuda()
    pack = to_pack(batch)
    out = model.forward(pack)
    assert np.allclose(out.encoder_out.cpu().numpy(), z["fwd/enc_out"], rtol=1e-4, atol=2e-5)
    assert np.array_equal(out.gold.cpu().numpy(), z["fwd/gold"])
    assert np.allclose(out.pred.cpu().numpy(), z["fwd/pred"], rtol=1e-4, atol=1e-4)
    metrics = model.cal_metrics(out, pack)
    assert abs(float(metrics.loss) - float(z["fwd/loss"])) < 1e-5 * abs(float(z["fwd/loss"]))
    # CER: tie-free convention (argmax) == oracle with greedy="argmax"
    id2tok = model.vocab._id2token
    want = R.cer_percent(torch.from_numpy(z["fwd/pred"]), torch.from_numpy(z["fwd/gold"]), id2tok, greedy="argmax")
    assert abs(float(metrics.cer) - want) < 1e-3

    # training step 1: gradients, clipped norm, parameters after Noam+Adam
    opt = make_opt(model, cfg, int(z["cfg/warm_up"]))
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    assert abs(float(loss[0]) - float(z["fwd/loss"])) < 1e-5 * abs(float(z["fwd/loss"]))
    gmax = max(float(np.abs(z["grad/" + n]).max()) for n, _ in model.named_parameters())
    for n, p in model.named_parameters():
        g = z["grad/" + n]
        assert np.allclose(p.grad.cpu().numpy(), g, rtol=2e-4, atol=2e-6 * max(gmax, 1.0)), n
    opt.fused_step(model._flat, 5.0)
    norm = float(opt.last_grad_sumsq.sqrt())
    assert abs(norm - float(z["step/grad_norm"])) < 1e-4 * float(z["step/grad_norm"])
    assert abs(opt._rate - float(z["step/lr"])) < 1e-12
    names = [n for n, _ in model.named_parameters() if not n.endswith("w_ks.bias")]   # see test_oracle_golden
    # Elements whose true gradient is ~0 (the softmax-shift-invariant part of w_ks, dead ReLU units ...) carry only the round-off of the
    # summation order, and Adam (eps = 1e-9) turns its SIGN into a full +-lr step in any implementation: compare the parameters
    # where the reference's gradient is significant, and bound every element by a few learning rates (as test_fp32_ctc_paths_match_oracle)
    lr1 = float(z["step/lr"])
    for n, p in model.named_parameters():
        if n in names:
            sig = np.abs(z["grad/" + n]) > 1e-5 * gmax
            got, want = p.detach().cpu().numpy(), z["step/" + n]
            assert np.allclose(got[sig], want[sig], rtol=1e-5, atol=3e-6), n
            assert np.abs(got - want).max() <= 2.5 * lr1, n
    # step 2 through the public entry point
    m2, _ = model.iterate(pack, optimizer=opt, is_train=True)
    assert abs(float(m2.loss) - float(z["step2/loss"])) < 3e-4 * abs(float(z["step2/loss"]))
    lr2 = float(z["step2/lr"])
    for n, p in model.named_parameters():
        if n in names:
            sig = np.abs(z["grad/" + n]) > 1e-5 * gmax
            got, want = p.detach().cpu().numpy(), z["step2/" + n]
            assert np.allclose(got[sig], want[sig], rtol=3e-4, atol=3e-5), n
            assert np.abs(got - want).max() <= 2.5 * (lr1 + lr2), n


def oracle_case(B, T, F, V, L, cfg_over, seed=5, ragged=True):
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    cfg = R.default_cfg(n_mels=F, lfr_m=1, **cfg_over)
    sd = R.init_state_dict(cfg, V, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in sd:   # non-trivial LayerNorm gains / biases
        if "layer_norm" in k and k.endswith("weight"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif "layer_norm" in k and k.endswith("bias"):
            sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
    pack = synthetic_pack(B, T, F, V, seed=seed + 2, ragged=ragged, Lmin=max(1, L - 4), Lmax=L)
    batch = {k: pack[k] for k in ("wave", "wave_len", "tgt_for_input", "tgt_len")}
    return cfg, sd, batch


@pytest.mark.parametrize("mode", ["joint", "ctc_only", "ce_wave_len"])
def test_fp32_ctc_paths_match_oracle(mode):
    over = dict(d_model=32, hidden_size=8, num_head=4, ff_size=64, layer_num=2)
    if mode == "joint":
        over.update(ctc_weight=0.3)
    elif mode == "ctc_only":
        over.update(use_decoder=False, ctc_weight=1.0)
    else:
        over.update(cross_mask="wave_len")
    cfg, sd, batch = oracle_case(4, 30, 16, 40, 6, over)
    tr = R.RefTrainer(sd, cfg, warmup=25)
    ref = tr.iterate(batch)
    model = build(cfg, 40, "TransformerCTC" if mode == "ctc_only" else "TransformerOffical", dtype="fp32").cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    opt = make_opt(model, cfg, 25)
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    # north_star: CTC loss within 1e-4 relative of the reference path
    assert abs(float(loss[0]) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
    if mode == "joint":
        assert abs(float(loss[2]) - float(ref["out"]["ctc"])) < 1e-4 * abs(float(ref["out"]["ctc"]))
        assert abs(float(loss[1]) - float(ref["out"]["ce"])) < 1e-4 * abs(float(ref["out"]["ce"]))
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    for n, p in model.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), ref["grads"][n].numpy(), rtol=3e-4, atol=3e-6 * max(gmax, 1.0)), n
    opt.fused_step(model._flat, 5.0)
    for n, p in model.named_parameters():
        if not n.endswith("w_ks.bias"):
            # elements whose true gradient is ~0 (dead ReLU units ...) carry only round-off noise,
            # which Adam (eps = 1e-9) turns into +-lr: compare where the gradient is significant
            sig = (ref["grads"][n].abs() > 1e-5 * gmax).numpy()
            got, want = p.detach().cpu().numpy(), tr.sd[n].numpy()
            assert np.allclose(got[sig], want[sig], rtol=1e-5, atol=3e-6), n
            assert np.abs(got - want).max() <= 2.5 * ref["lr"], n


def _report(name, rows):
    """Measured parity figures of the bf16 path, kept for DESIGN.md (gpurun_out/ is merged back from the GPU box)."""
    import json
    import os
    from tests.helpers import ROOT
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bf16_parity.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **rows}) + "\n")


# SURVEY 8(d) tolerances for bf16 storage: loss <= 1e-3 relative, per-tensor gradient cosine >= 0.999.
BF16_LOSS_RTOL = 1e-3
BF16_COS = 0.999
# Tensors that cannot meet the 0.999 cosine gate, and why (measured worst cases are written to gpurun_out/bf16_parity.jsonl;
# tools/bf16_parity.py prints every tensor).  In fp32 mode all of them agree with the oracle to 1e-6.
#   *.w_ks.bias                         analytically zero gradient (softmax is shift-invariant): pure round-off in any implementation.
#   (decoder.slf_attn.w_qs / w_ks       left this list in round 5: their gradients - 1e-4 .. 1e-5 of the largest one - are formed as P o (dP - delta)
#                                       with dP ~ delta, and delta = rowsum(dO o O) from the bf16-ROUNDED attention output carried 2^-9 |delta|
#                                       of rounding into that difference; for heads whose keys one wave holds the backward now takes delta as
#                                       sum p dP / sum p from its own p and dP: 0.99899 -> 0.99980 on the worst of them.  The cross-attention
#                                       (500 keys over eight waves) keeps the flash form and measures 0.9997.)
#   decoder.*.pos_ffn.w_1               ReLU inputs within bf16 rounding of zero fall on the other side of the ReLU; 50 rows. Measured 0.9989.
BF16_COS_EXEMPT = ("w_ks.bias",)
BF16_COS_RELAXED = (("decoder.", "pos_ffn.w_1."),)
BF16_COS_RELAXED_MIN = 0.997


def bf16_gradient_gate(model, ref, name, cos_min=BF16_COS):
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    worst, worst_name, worst_ratio, worst_relaxed = 1.0, "", 1.0, 1.0
    for n, p in model.named_parameters():
        g = ref["grads"][n]
        if float(g.abs().max()) < 1e-6 * max(gmax, 1e-30) or n.endswith(BF16_COS_EXEMPT):
            continue
        c = cos(p.grad, g)
        r = float(p.grad.double().norm().cpu() / g.double().norm())
        relaxed = any(n.startswith(a) and b in n for a, b in BF16_COS_RELAXED)
        if relaxed:
            worst_relaxed = min(worst_relaxed, c)
        elif c < worst:
            worst, worst_name = c, n
        if abs(r - 1) > abs(worst_ratio - 1):
            worst_ratio = r
        assert c > (BF16_COS_RELAXED_MIN if relaxed else cos_min), (n, c)
        assert 0.97 < r < 1.03, (n, r)
    return worst, worst_name + f" (relaxed class: {worst_relaxed:.5f})", worst_ratio


@pytest.mark.parametrize("mode", ["joint", "ctc_only"])
def test_bf16_mfma_path_matches_oracle(mode):
    """d_model 512 / 8 heads x 64 / ff 1024: the shapes the MFMA kernels are built for.  Gates = SURVEY 8(d): loss
    1e-3 relative, every gradient tensor's cosine >= 0.999 and norm within 3 %."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2)
    over.update(dict(ctc_weight=0.3) if mode == "joint" else dict(use_decoder=False, ctc_weight=1.0))
    B, T, F, V, L = 4, 136, 80, 56, 12
    cfg, sd, batch = oracle_case(B, T, F, V, L, over, seed=9)
    if "decoder.tgt_word_emb.weight" in sd:   # keep CE in a sane range: N(0,1) tied embedding scaled down
        sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
        sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    tr = R.RefTrainer(sd, cfg, warmup=25)
    ref = tr.iterate(batch)
    model = build(cfg, V, "TransformerCTC" if mode == "ctc_only" else "TransformerOffical", dtype="bf16").cuda()
    model.load_state_dict(sd)
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    rel = abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"]))
    worst, worst_name, ratio = bf16_gradient_gate(model, ref, "mfma_" + mode)
    _report("bf16_mfma_path_" + mode, dict(loss_rel=rel, worst_cos=worst, worst_tensor=worst_name, worst_norm_ratio=ratio))
    assert rel < BF16_LOSS_RTOL, (float(loss[0]), float(ref["loss"]))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_default_width_matches_reference_golden(dtype):
    """The kernels the bench times, against the REFERENCE directly (no oracle in between): TransformerOffical at its default
    width (d_model 512, 8 heads x 64, ff 1024; transformer_official.py:115-122), one encoder + one decoder layer, ragged B = 3,
    T = 140.  Weights and batch are rebuilt from numpy seeds (oracle/golden_inputs.py); expected values are what the reference
    computed from the same arrays (tests/golden/model_mfma_d512.npz: `python oracle/gen_golden.py mfma`).
    bf16 (MFMA GEMMs + fused MFMA attention): the SURVEY 8(d) bf16 gates - logits 2e-2 abs + 1e-2 rel, loss 1e-3 rel, per-tensor
    gradient cosine >= 0.999 on the sampled elements (same relaxed classes as bf16_gradient_gate, with the reasons stated
    there), norm within 3 %, clip norm within 1 %.  fp32 (own fp32 kernels): logits 1e-4, loss 1e-5, gradients 3e-4."""
    cfg, sd, batch, z, GI = mfma_golden_case()
    V = GI.MFMA_CASE["V"]
    model = build(cfg, V, dtype=dtype)
    model.load_state_dict(sd)
    model = model.cuda()
    pack = to_pack(batch)
    out = model.forward(pack)
    pred, want = out.pred.float().cpu().numpy(), z["fwd/pred"]
    assert np.array_equal(out.gold.cpu().numpy(), z["fwd/gold"])
    if dtype == "fp32":
        assert np.allclose(pred, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    else:
        # |error| <= 2e-2 + 1e-2 |logit|, every element.  The gate is a MAXIMUM over ~1.4e5 elements of an error whose rms is
        # measured at a fifth of the bound: the worst element sits where such a maximum is expected (~4.5 sigma), so the margin of the
        # gate is stated by the two figures reported below (worst element / its own bound, rms / bound), not by max |error| alone.
        bound = 2e-2 + 1e-2 * np.abs(want)
        ratio = np.abs(pred - want) / bound
        logit_worst_ratio, logit_rms_ratio = float(ratio.max()), float(np.sqrt(np.mean(ratio ** 2)))
        assert logit_worst_ratio <= 1.0, (logit_worst_ratio, float(np.abs(pred - want).max()))
        assert logit_rms_ratio <= 0.3, logit_rms_ratio      # measured 0.237: the typical element has 4x headroom; the worst of 1.4e5 elements
        # sits at 0.965 of its bound, where the maximum of that many errors of this rms is expected (4.1 sigma).  The error is the bf16 rounding of
        # the decoder's activations (rms 0.6 % per tensor); the stored logits' own rounding is a fifth of it (tools/ctc_parity_diag.py for the
        # encoder side) - there is no cheaper place to buy margin than fp32 activations
    opt = make_opt(model, cfg, GI.MFMA_CASE["warm_up"])
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    rel = abs(float(loss[0]) - float(z["fwd/loss"])) / abs(float(z["fwd/loss"]))
    assert rel < (1e-5 if dtype == "fp32" else BF16_LOSS_RTOL), (float(loss[0]), float(z["fwd/loss"]))
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad_s/"))
    worst, worst_name, worst_ratio, worst_relaxed, sq = 1.0, "", 1.0, 1.0, 0.0
    for n, p in model.named_parameters():
        g = p.grad.double().flatten().cpu().numpy()
        sq += float((g * g).sum()) if not n.endswith("tgt_word_prj.weight") else 0.0
        w = z["grad_s/" + n].astype(np.float64)
        if float(np.abs(w).max()) < 1e-6 * gmax or n.endswith(BF16_COS_EXEMPT):
            continue
        gs = g[GI.sample_index(n, g.size)]
        norm = float(z["grad_norm/" + n])
        if dtype == "fp32":
            assert np.allclose(gs, w, rtol=3e-4, atol=3e-6 * max(float(np.abs(w).max()), 1e-3)), n
            assert abs(float(np.sqrt((g * g).sum())) - norm) < 1e-4 * norm, n
            assert abs(float(g @ GI.probe(n, g.size)) - float(z["grad_probe/" + n])) < 1e-3 * norm, n      # every element, not only the samples
            continue
        c = float((gs @ w) / (np.linalg.norm(gs) * np.linalg.norm(w) + 1e-300))
        r = float(np.sqrt((g * g).sum())) / norm
        # The ReLU-flip class (see BF16_COS_RELAXED: hidden units whose input is within bf16 rounding of zero fall on the other side of
        # the ReLU) also covers the ENCODER's w_1 here: this batch has 308 valid frames, and the model is trained with CE only (the
        # reference has no CTC), so the encoder's gradient is the small signal that 19 decoder rows send through cross-attention.
        # Measured (tools/d512_parity.py): 0.9980 on the samples, 0.9983 over every element against the oracle; every other
        # encoder tensor meets 0.999 (0.9995 .. 0.9999).  With a CTC term and 4 x 136 frames the same tensors reach 0.9997.
        relaxed = any(n.startswith(a) and b in n for a, b in BF16_COS_RELAXED) or "pos_ffn.w_1." in n
        if relaxed:
            worst_relaxed = min(worst_relaxed, c)
        elif c < worst:
            worst, worst_name = c, n
        if abs(r - 1) > abs(worst_ratio - 1):
            worst_ratio = r
        assert c > (BF16_COS_RELAXED_MIN if relaxed else BF16_COS), (n, c)
        assert 0.97 < r < 1.03, (n, r)
    gn, gn_want = float(np.sqrt(sq)), float(z["step/grad_norm"])
    assert abs(gn - gn_want) < (1e-4 if dtype == "fp32" else 1e-2) * gn_want, (gn, gn_want)
    if dtype == "bf16":
        _report("default_width_vs_reference_bf16", dict(loss_rel=rel, worst_cos=worst, worst_tensor=worst_name, worst_relaxed_cos=worst_relaxed,
                                                         worst_norm_ratio=worst_ratio, logits_max_abs=float(np.abs(pred - want).max()),
                                                         logits_worst_over_bound=logit_worst_ratio, logits_rms_over_bound=logit_rms_ratio))
    # and through the public entry point: clip + Noam + Adam, then the loss of the second iterate
    m1, _ = model.iterate(pack, optimizer=opt, is_train=True)
    m2, _ = model.iterate(pack, optimizer=opt, is_train=True)
    assert abs(float(m1.loss) - float(z["fwd/loss"])) < (1e-5 if dtype == "fp32" else BF16_LOSS_RTOL) * abs(float(z["fwd/loss"]))
    assert abs(float(m2.loss) - float(z["step2/loss"])) < (2e-4 if dtype == "fp32" else 5e-3) * abs(float(z["step2/loss"])), (float(m2.loss), float(z["step2/loss"]))
    if dtype == "fp32":
        # CER of the REFERENCE's logits under the tie-free greedy convention (first index on the all-equal padded rows, where the
        # id the reference's topk returns is implementation-defined: oracle/ref_model.cer_percent)
        id2tok = ["$", "%", "^", "&"] + [chr(0x4E00 + i) for i in range(V - 4)]
        want_cer = R.cer_percent(torch.from_numpy(z["fwd/pred"]), torch.from_numpy(z["fwd/gold"]), id2tok, greedy="argmax")
        assert abs(float(m1.cer) - want_cer) < 1e-3, (float(m1.cer), want_cer)


@pytest.mark.parametrize("mode", ["ctc_only", "joint"])
def test_ctc_greedy_cer_matches_oracle(mode):
    """Evaluation of a CTC model reports the CER of best-path decoding: the device decode of the
    model's own logits equals the oracle's decode, and the CER follows the reference's convention."""
    over = dict(d_model=32, hidden_size=8, num_head=4, ff_size=64, layer_num=2)
    over.update(dict(use_decoder=False, ctc_weight=1.0) if mode == "ctc_only" else dict(ctc_weight=0.3))
    cfg, sd, batch = oracle_case(5, 40, 16, 30, 7, over)
    model = build(cfg, 30, "TransformerCTC" if mode == "ctc_only" else "TransformerOffical", dtype="fp32").cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    model.eval()
    pack = to_pack(batch)
    out = model.forward(pack)
    want = R.ctc_greedy_decode(out.ctc_logits.float().cpu().numpy(), batch["wave_len"].numpy())
    assert model.ctc_greedy_search(pack) == want
    labels = [[int(t) for t in row if int(t) != 0] for row in batch["tgt_for_input"]]   # Decoder.preprocess strips the 0 padding
    ev, _ = model.iterate(pack, is_train=False)
    key = "cer" if mode == "ctc_only" else "ctc_cer"
    hyp_s = [model.vocab.convert_id2str(h) for h in want]
    ref_s = [model.vocab.convert_id2str(l) for l in labels]
    cer = sum(R.edit_distance(h, r) / len(r.split(" ")) for h, r in zip(hyp_s, ref_s)) * 100 / len(want)
    assert abs(float(getattr(ev, key)) - cer) < 1e-4
    assert np.isfinite(float(ev.loss))


@pytest.mark.parametrize("case", ["beam_small.npz", "beam_small_maxlen.npz"])
def test_beam_search_matches_reference_golden(case):
    """GPU beam search (KV caches, batched beams) against the hypotheses the reference's own
    Decoder.recognize_beam produced: n-best token sequences exactly, scores to 1e-4 (fp32 mode)."""
    from tests.helpers import load_npz
    from asr_chinese_e2e_amd.Utils import Pack
    z = load_npz(case)
    cfg = R.default_cfg(**{k: (int(v) if float(v).is_integer() else float(v)) for k, v in zip(z["cfg/keys"], z["cfg/vals"])})
    V = int(z["cfg/V"])
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    model = build(cfg, V, dtype="fp32").cuda()
    model.load_state_dict(sd, strict=False)
    model.eval()
    pack = Pack()
    pack.add(wave=torch.from_numpy(z["in/wave"]).to(DEV), wave_len=torch.from_numpy(z["in/wave_len"]).to(DEV))
    beam, nbest, dml = int(z["cfg/beam"]), int(z["cfg/nbest"]), int(z["cfg/decode_max_len"])
    got = model.beam_search(pack, beam, nbest, dml)
    for b, hyps in enumerate(got):
        want_seq, want_len, want_score = z[f"beam/{b}/yseq"], z[f"beam/{b}/len"], z[f"beam/{b}/score"]
        assert len(hyps) == len(want_len), (b, hyps)
        for h, ws, wl, wsc in zip(hyps, want_seq, want_len, want_score):
            assert h["yseq"] == [int(t) for t in ws[: int(wl)]], (b, h, ws)
            assert abs(h["score"] - float(wsc)) < 1e-4


@pytest.mark.parametrize("dtype,beam", [("fp32", 4), ("bf16", 3)])
def test_beam_search_matches_oracle(dtype, beam):
    """Random model at MFMA-capable geometry (d_model 64, heads of 64 in bf16): fp32 must give the
    oracle's hypotheses; bf16 must give the same best hypothesis score within bf16 tolerance."""
    over = dict(d_model=64, hidden_size=64 if dtype == "bf16" else 16, num_head=2 if dtype == "bf16" else 4, ff_size=128, layer_num=2)
    cfg, sd, batch = oracle_case(3, 18, 16, 24, 5, over, seed=9)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 3.0      # peaked outputs: hypotheses end before maxlen
    if "decoder.tgt_word_prj.weight" in sd:                                             # tied: the state dict carries both names
        sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    model = build(cfg, 24, dtype=dtype).cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    model.eval()
    pack = to_pack(batch)
    got = model.beam_search(pack, beam, 2, 10)
    enc = R.encoder_forward(sd, cfg, batch["wave"], batch["wave_len"])
    for b in range(3):
        want = R.beam_search(sd, cfg, enc[b, : int(batch["wave_len"][b])], beam, 2, 10)
        assert len(got[b]) == len(want)
        if dtype == "fp32":
            for h, (ids, score) in zip(got[b], want):
                assert h["yseq"] == ids
                assert abs(h["score"] - score) < 1e-4 * max(1.0, abs(score))
        else:
            # ~10 steps of bf16 logits through a sharpened softmax (the case scales the tied embedding by 3): the summed
            # log-probability moves by up to 0.14 (measured, gpurun_out/bf16_parity.jsonl); gate 0.2 (round 2: 0.3)
            err = abs(got[b][0]["score"] - want[0][1]) / max(1.0, abs(want[0][1]))
            _report("beam_search_bf16_score", dict(utt=b, err=err, score=want[0][1]))
            assert err < 0.2, (got[b][0], want[0])


def test_label_smoothing_matches_reference_formula():
    """config.label_smoothing switches on the reference's smoothing branch (Utils/loss.py:30-45):
    loss and gradients follow the oracle's restatement of it."""
    over = dict(d_model=32, hidden_size=8, num_head=4, ff_size=64, layer_num=2)
    cfg, sd, batch = oracle_case(4, 20, 16, 30, 5, over, seed=13)
    model = build(cfg, 30, dtype="fp32", label_smoothing=0.1).cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "positional" not in k}
    full = dict(sd, **sdg)
    if "decoder.tgt_word_prj.weight" in full:
        full["decoder.tgt_word_prj.weight"] = full["decoder.tgt_word_emb.weight"]
    enc = R.encoder_forward(full, cfg, batch["wave"], batch["wave_len"])
    pred, gold = R.decoder_forward(full, cfg, batch["tgt_for_input"], enc, batch["tgt_len"])
    ref = R.ce_loss(pred, gold, 0.1)
    ref.backward()
    assert abs(float(loss[0]) - float(ref)) < 1e-5 * abs(float(ref))
    for n, p in model.named_parameters():
        if n in sdg and sdg[n].grad is not None and not n.endswith("tgt_word_prj.weight"):
            g = sdg[n].grad
            assert np.allclose(p.grad.cpu().numpy(), g.numpy(), rtol=3e-4, atol=3e-6 * max(float(g.abs().max()), 1.0)), n


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_baseline_config0_matches_oracle(dtype):
    """BASELINE.json configs[0] (the reference's CPU-runnable case): 2-layer encoder + CTC at the default
    width (d_model 512, 8 x 64 heads, ff 1024), batch 4, 100 frames of 80 mels, 50-character vocabulary.
    fp32: loss 1e-4 rel; gradients: 99.9 % of every tensor's elements within 1e-3 rel + 2e-5 of the
    largest gradient, and no element off by more than 1 % of its tensor's maximum (typical agreement is
    3e-5; with 409 600 ReLU inputs an input within rounding of zero can land on the other side of the
    ReLU on the GPU, which moves a handful of w_1 gradient elements by ~4e-3 and everything upstream of
    that layer by ~3e-4); bf16 MFMA path: the SURVEY 8(d) gates (loss 1e-3 relative, per-tensor gradient cosine > 0.999)."""
    over = dict(layer_num=2, use_decoder=False, ctc_weight=1.0)
    cfg, sd, batch = oracle_case(4, 100, 80, 50, 12, over, seed=21)
    ref = R.RefTrainer(sd, cfg, warmup=4000).iterate(batch)
    model = build(cfg, 50, "TransformerCTC", dtype=dtype).cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    tol = 1e-4 if dtype == "fp32" else BF16_LOSS_RTOL
    rel = abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"]))
    if dtype == "bf16":
        worst, worst_name, ratio = bf16_gradient_gate(model, ref, "config0")
        _report("baseline_config0_bf16", dict(loss_rel=rel, worst_cos=worst, worst_tensor=worst_name, worst_norm_ratio=ratio))
    assert rel < tol, (float(loss[0]), float(ref["loss"]))
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    for n, p in model.named_parameters():
        g = ref["grads"][n]
        if dtype == "fp32":
            if n.endswith("w_ks.bias"):
                continue
            d = (p.grad.cpu() - g).abs()
            # ONE hidden unit whose pre-activation is within rounding of zero falls on the other side of the ReLU on the GPU
            # (measured, tools/config0_fp32_diag.py: exactly one row of layer 0's w_1 gradient - 1 / 1024 of its elements - is off by up to
            # 1 % of the tensor's maximum, layer 1 agrees to 2e-5): everything UPSTREAM of that unit then moves by a few 1e-4 of its
            # tensor's maximum.  Gate per tensor: 99.9 % of the elements within 1e-3 relative + 2e-5 of the largest gradient, OR every
            # element within 1e-3 of the tensor's maximum; and never more than 1 % of the maximum.
            frac_off = float((d > 1e-3 * g.abs() + 2e-5 * max(gmax, 1.0)).float().mean())
            assert frac_off < 1e-3 or float(d.max()) < 1e-3 * float(g.abs().max()), (n, frac_off, float(d.max()) / float(g.abs().max()))
            assert float(d.max()) < 1e-2 * float(g.abs().max()), n


def test_padded_rows_are_exact_zero_and_ignore_garbage():
    """Post-LN pad zeroing (transformer_official.py:208, 211): encoder output rows t >= wave_len are
    exactly 0 and garbage in the padded input frames cannot change any valid output."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=1, use_decoder=False, ctc_weight=1.0)
    cfg, sd, batch = oracle_case(3, 70, 80, 30, 5, over, seed=3)
    model = build(cfg, 30, "TransformerCTC", dtype="bf16").cuda()
    model.load_state_dict(sd)
    p1 = to_pack(batch)
    o1 = model.forward(p1).encoder_out.float().cpu()
    b2 = dict(batch)
    w = batch["wave"].clone()
    for b in range(3):
        w[b, int(batch["wave_len"][b]):] = 1e3
    b2["wave"] = w
    o2 = model.forward(to_pack(b2)).encoder_out.float().cpu()
    for b in range(3):
        n = int(batch["wave_len"][b])
        assert float(o1[b, n:].abs().max()) == 0.0 if n < 70 else True
        assert torch.equal(o1[b, :n], o2[b, :n])


def test_full_config_step_is_finite_and_learns():
    """BASELINE config 2 shape (B=32, T=500, F=80, V=4232, 6 layers): a few steps run, loss is
    finite and decreases on a repeated batch."""
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    cfg = R.default_cfg(n_mels=80, lfr_m=1, use_decoder=False, ctc_weight=1.0)
    model = build(cfg, 4232, "TransformerCTC", dtype="bf16").cuda()
    opt = make_opt(model, cfg, 20)
    pack = synthetic_pack(32, 500, 80, 4232, device=DEV)
    losses = []
    for _ in range(6):
        m, _ = model.iterate(pack, optimizer=opt, is_train=True)
        losses.append(float(m.loss))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


def test_transposed_weight_copies_track_the_weights():
    """The own-kernel input gradients read W^T copies made once per step: after every step the copies used by the NEXT
    backward must equal the updated weights, also when the module is in eval mode while training (dropout off), and a
    stale copy (weights changed without a refresh) must be REFUSED - round 2 fell back to the library GEMM there, a silent
    2x cliff; there is no library GEMM any more."""
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    cfg = R.default_cfg(n_mels=80, lfr_m=1, use_decoder=False, ctc_weight=1.0)
    cfg.layer_num = 2
    model = build(cfg, 200, "TransformerCTC", dtype="bf16").cuda()
    opt = make_opt(model, cfg, 20)
    pack = synthetic_pack(16, 300, 80, 200, device=DEV)          # 4800 rows: the own-kernel path is active
    model.eval()                                                  # training with the module in eval mode: still refreshed
    for _ in range(3):
        model.iterate(pack, optimizer=opt, is_train=True)
    eng, flat = model._engine, model._flat
    lin = eng.enc[1][1].w2
    assert lin.wlpT is not None
    dy = torch.randn(4800, lin.N, device=DEV).bfloat16()
    assert lin.own_dgrad(dy) and flat.lpT_version != flat.version   # the optimizer has just rewritten the weights: copies are stale
    torch.cuda.synchronize()
    stale = lin.wlpT.clone()
    assert not torch.equal(stale.t().contiguous(), lin.wlp)      # ... and they really differ from the current weights
    with pytest.raises(RuntimeError, match="stale"):
        lin.dgrad(dy)
    ref = (dy.float() @ lin.wlp.float())
    eng.refresh_transposes()
    eng.wait_transposes()
    torch.cuda.synchronize()
    assert flat.lpT_version == flat.version and torch.equal(lin.wlpT.t().contiguous(), lin.wlp)
    out2 = lin.dgrad(dy)                                          # own kernel on the fresh copy
    assert float((out2.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-2
    # the CTC head's input gradient (reduction over V = 200 here, 4232 in the bench) runs on the same kernel: no library GEMM
    head = eng.ctc_lo
    dl = torch.randn(4800, head.N, device=DEV).bfloat16()
    assert head.wlpT is not None and head.own_dgrad(dl)
    got = head.dgrad(dl)
    want = dl.float() @ head.wlp.float()
    assert float((got.float() - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-2


def test_no_cpu_fallback():
    cfg, sd, batch = oracle_case(2, 10, 16, 20, 3, dict(d_model=32, hidden_size=8, num_head=4, ff_size=64, layer_num=1))
    model = build(cfg, 20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.forward(to_pack(batch, "cpu"))


def test_dropout_training_and_eval_modes():
    """Reference default dropout 0.1: eval mode ignores it (identical to a dropout-0 model), train
    mode is reproducible for a given step seed, differs between steps, and still learns."""
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3)
    cfg, sd, batch = oracle_case(4, 64, 80, 40, 8, over, seed=4, ragged=True)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    pack = to_pack(batch)
    m0 = build(cfg, 40, dtype="bf16").cuda()
    m0.load_state_dict(sd)
    m1 = build(cfg, 40, dtype="bf16", dropout=0.1).cuda()
    m1.load_state_dict(sd)
    m0.eval(); m1.eval()
    a, _ = m0.iterate(pack, is_train=False)
    b, _ = m1.iterate(pack, is_train=False)
    assert float(a.loss) == float(b.loss)
    m1.train()
    m1._ensure_engine(DEV)
    losses = []
    for seed in (7, 7, 8):
        m1._step_seed = seed
        m1.zero_flat_grads()
        loss, _ = m1.train_step(pack)
        losses.append((float(loss[0]), float(m1._flat.g.abs().sum())))
    assert losses[0] == losses[1]                       # same seed -> same masks, bit-identical step
    assert losses[0][0] != losses[2][0]                 # next step -> new masks
    assert abs(losses[0][0] - float(a.loss)) < 0.25 * abs(float(a.loss))
    opt = make_opt(m1, cfg, 10)
    first = None
    for i in range(12):
        m, _ = m1.iterate(pack, optimizer=opt, is_train=True)
        first = float(m.loss) if first is None else first
    assert np.isfinite(float(m.loss)) and float(m.loss) < first


# ------------------------------------------------------------------------------------ BASELINE configs[2], [4]
@pytest.mark.parametrize("mode", ["joint", "ctc_only"])
def test_full_size_step_matches_oracle(mode):
    """BASELINE.json configs[2] (joint, lambda = 0.3) and configs[1] (CTC-only) at FULL size - B = 32, T = 500, F = 80, V = 4232, 6 (+ 6) layers,
    d_model 512, ragged lengths, random initial weights - against the oracle's fp32 step on the same weights and batch: the workload
    bench.py times.  Gates: loss 1e-3 relative; every gradient tensor whose largest element is at least 1e-5 of the step's largest gradient
    element: cosine >= 0.999 (joint) / 0.998 (CTC-only) and norm within 3 %; the tensors below that (at initialisation: the decoder's attention Q / K projections,
    1e-7 .. 9e-6 of the largest gradient - softmax over 500 near-equal scores) carry no weight in an update and must reach 0.97.
    Found with this test (round 5): 0.989 for the top encoder layer's Q / K projections and 0.90 for the decoder's cross-attention ones until
    the attention backward centred its keys and took the mean over the keys out of dK (sdpa.hip; now 0.9998 and 0.998).  The oracle takes ~4 s per step
    on the GPU box's 16 host cores (bench.py's cpu_baseline leg times the same call), ~15 s with its set-up."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=6)
    over.update(dict(ctc_weight=0.3) if mode == "joint" else dict(use_decoder=False, ctc_weight=1.0))
    B, T, F, V, L = 32, 500, 80, 4232, 17
    cfg, sd, batch = oracle_case(B, T, F, V, L, over, seed=13)
    if "decoder.tgt_word_emb.weight" in sd:   # keep CE in a sane range: N(0,1) tied embedding scaled down
        sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
        sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    ref = R.RefTrainer(sd, cfg, warmup=25).iterate(batch)
    model = build(cfg, V, "TransformerCTC" if mode == "ctc_only" else "TransformerOffical", dtype="bf16").cuda()
    model.load_state_dict(sd)
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    rel = abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"]))
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    worst, worst_small, worst_ratio = (1.0, ""), (1.0, ""), 1.0
    for n, p in model.named_parameters():
        g = ref["grads"][n]
        if n.endswith(BF16_COS_EXEMPT) or float(g.abs().max()) < 1e-9 * gmax:
            continue
        c = cos(p.grad, g)
        if float(g.abs().max()) >= 1e-5 * gmax:
            r = float(p.grad.double().norm().cpu() / g.double().norm())
            worst = min(worst, (c, n))
            worst_ratio = r if abs(r - 1) > abs(worst_ratio - 1) else worst_ratio
        else:
            worst_small = min(worst_small, (c, n))
    _report("full_size_" + mode, dict(loss_rel=rel, worst_cos=worst[0], worst_tensor=worst[1], worst_cos_of_insignificant=worst_small[0],
                                      worst_small_tensor=worst_small[1], worst_norm_ratio=worst_ratio))
    assert rel < BF16_LOSS_RTOL, (float(loss[0]), float(ref["loss"]))
    # joint: 0.99930 measured (the decoder's w_1); CTC-only: 0.99852 - there the attention Q / K projections of ALL encoder layers, the input
    # projection and the LayerNorm gains sit at 0.9985 .. 0.9990: that is the rounding of the PARAMETERS to bf16, see below
    assert worst[0] > (BF16_COS if mode == "joint" else 0.998), worst
    assert worst_small[0] > 0.97, worst_small
    assert 0.97 < worst_ratio < 1.03, worst_ratio
    # Second oracle run on the weight MATRICES rounded to bf16 - the numbers the MFMA path multiplies by: what is left of the difference is the
    # rounding of activations and of the kernels' intermediates.  Measured: loss 1.2e-6 (joint) / 3.7e-5 (CTC-only); every significant tensor
    # >= 0.99969 (joint; the decoder's w_1, the ReLU-flip class, 0.99950) / >= 0.99987 (CTC-only).
    sd16 = {k: (v.bfloat16().float() if v.dim() == 2 else v) for k, v in sd.items()}
    ref16 = R.RefTrainer(sd16, cfg, warmup=25).iterate(batch)
    rel16 = abs(float(loss[0]) - float(ref16["loss"])) / abs(float(ref16["loss"]))
    gmax = max(float(g.abs().max()) for g in ref16["grads"].values())
    worst16, worst16_relaxed = (1.0, ""), (1.0, "")
    for n, p in model.named_parameters():
        g = ref16["grads"][n]
        if n.endswith(BF16_COS_EXEMPT) or float(g.abs().max()) < 1e-5 * gmax:
            continue
        c = cos(p.grad, g)
        if any(n.startswith(a) and b in n for a, b in BF16_COS_RELAXED):
            worst16_relaxed = min(worst16_relaxed, (c, n))
        else:
            worst16 = min(worst16, (c, n))
    _report("full_size_" + mode + "_oracle_on_bf16_weights", dict(loss_rel=rel16, worst_cos=worst16[0], worst_tensor=worst16[1],
                                                                    worst_relaxed_cos=worst16_relaxed[0]))
    assert rel16 < 2e-4, rel16
    assert worst16[0] > 0.9995, worst16
    assert worst16_relaxed[0] > BF16_COS, worst16_relaxed


def test_full_size_joint_step_properties():
    """BASELINE.json configs[2] at full size (joint CTC/attention, lambda = 0.3, B=32, T=500, F=80, V=4232, 6+6 layers, bf16),
    ragged lengths: steps run, the loss is finite and falls on a repeated batch, and the size-independent properties
    hold - padded encoder rows exactly 0, every CTC gradient row sums to 0 (so does the CTC head's bias gradient),
    padded frames / tokens contribute nothing."""
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    cfg = R.default_cfg(n_mels=80, lfr_m=1, ctc_weight=0.3)
    torch.manual_seed(0)
    model = build(cfg, 4232, "TransformerOffical", dtype="bf16").cuda()
    with torch.no_grad():      # N(0,1) tied embedding at d=512 gives logits of +-20: keep CE in a sane range for the loss trend
        model.decoder.tgt_word_emb.weight.mul_(0.05)
    opt = make_opt(model, cfg, 20)
    pack = synthetic_pack(32, 500, 80, 4232, seed=11, ragged=True, device=DEV)
    model._ensure_engine(DEV)
    model._flat.refresh_lowp()
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    torch.cuda.synchronize()
    assert all(np.isfinite(loss.cpu().numpy())), loss
    g = {n: p.grad for n, p in model.named_parameters()}
    assert all(bool(torch.isfinite(v).all()) for v in g.values())
    gb = g["ctc_lo.bias"].double()
    assert abs(float(gb.sum())) < 2e-2 * float(gb.abs().sum()), (float(gb.sum()), float(gb.abs().sum()))
    # class 0 doubles as CTC blank and decoder padding: the embedding row of <pad> gets no CE gradient through the tied projection's
    # targets, and no gradient tensor is identically zero
    assert all(float(v.abs().max()) > 0 for n, v in g.items() if not n.endswith("w_ks.bias"))
    enc = model.forward(pack).encoder_out.float()
    for b in range(32):
        n = int(pack.wave_len[b])
        if n < 500:
            assert float(enc[b, n:].abs().max()) == 0.0
    losses = []
    for _ in range(5):
        m, _ = model.iterate(pack, optimizer=opt, is_train=True)
        losses.append(float(m.loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_w1_gradient_cosine_recovers_without_relu_flips():
    """The relaxed cosine class of the bf16 gates holds `pos_ffn.w_1` (BF16_COS_RELAXED; at the default-width case also the encoder's): the
    stated reason is that hidden units whose pre-activation lies within bf16 rounding of zero fall on the other side of the ReLU in the
    bf16 forward pass, which switches whole (frame, unit) contributions dh[frame, unit] x[frame, :] of the weight gradient on or off.
    Demonstrated here instead of asserted: with the (frame, unit) pairs whose ORACLE pre-activation is within one bf16 ulp of the
    rounding scale of zero removed from BOTH sides - the oracle's gradient recomputed from its own dh and x, the kernels' from the dh and
    x they handed to the weight-gradient GEMM - the cosine of the encoder's w_1 gradient is >= 0.999 (the strict gate), while the
    unmasked one is the ~0.998 the relaxed class was introduced for."""
    cfg, sd, batch, z, GI = mfma_golden_case()
    V = GI.MFMA_CASE["V"]
    # ---- oracle (fp64) with the feed-forward internals of encoder layer 0 captured
    keep = {}
    orig_ff = R.feed_forward

    def ff_capture(sd_, pre, x):
        if pre != "encoder.layer_stack.0.pos_ffn.":
            return orig_ff(sd_, pre, x)
        import torch.nn.functional as F
        w1, w2 = sd_[pre + "w_1.weight"].squeeze(-1), sd_[pre + "w_2.weight"].squeeze(-1)
        hp = F.linear(x, w1, sd_[pre + "w_1.bias"])
        hp.retain_grad()
        keep["hp"], keep["x"] = hp, x
        h = F.linear(F.relu(hp), w2, sd_[pre + "w_2.bias"])
        return F.layer_norm(h + x, (x.shape[-1],), sd_[pre + "layer_norm.weight"], sd_[pre + "layer_norm.bias"], R.LN_EPS)

    sd64 = {k: v.double() for k, v in sd.items()}
    b64 = dict(batch, wave=batch["wave"].double())
    R.feed_forward = ff_capture
    try:
        tr = R.RefTrainer(sd64, cfg, warmup=25)
        leaves = {k: tr.sd[k].detach().clone().requires_grad_(True) for k in tr.trainable}
        sdl = dict(tr.sd)
        sdl.update(leaves)
        sdl["decoder.tgt_word_prj.weight"] = leaves["decoder.tgt_word_emb.weight"]
        out = R.forward_losses(sdl, cfg, b64)
        out["loss"].backward()
    finally:
        R.feed_forward = orig_ff
    hp, x_o = keep["hp"].detach().reshape(-1, keep["hp"].shape[-1]), keep["x"].detach().reshape(-1, keep["x"].shape[-1])
    dh_o = keep["hp"].grad.reshape(hp.shape)                     # gradient wrt the pre-activation (the ReLU derivative is inside)
    g_full_o = leaves["encoder.layer_stack.0.pos_ffn.w_1.weight"].grad.squeeze(-1)
    assert float((dh_o.t() @ x_o - g_full_o).abs().max()) < 1e-9 * float(g_full_o.abs().max()) + 1e-12
    # ---- the bf16 kernels: the operands they hand to the weight-gradient GEMM of that projection
    model = build(cfg, V, dtype="bf16")
    model.load_state_dict(sd)
    model = model.cuda()
    eng = model._ensure_engine(DEV)
    w1_lin = eng.enc[0][1].w1
    got = {}
    orig_wgrad = eng._wgrad

    def wgrad_capture(lin, dy, x, bias_from=None):
        if lin is w1_lin:
            got["dh"], got["x"] = dy.detach().double().cpu(), x.detach().double().cpu()
        return orig_wgrad(lin, dy, x, bias_from)

    eng._wgrad = wgrad_capture
    model.zero_flat_grads()
    model.train_step(to_pack(batch))
    torch.cuda.synchronize()
    eng._wgrad = orig_wgrad
    g_full_k = dict(model.named_parameters())["encoder.layer_stack.0.pos_ffn.w_1.weight"].grad.double().cpu().squeeze(-1)
    c_full = cos(g_full_k, g_full_o)
    # pairs at risk, decided from the ORACLE alone: |pre-activation| below 2^-6 of its scale (rms of the input row x norm of the unit's weight
    # row).  The bf16 path's pre-activation differs from the oracle's by the rounding accumulated upstream - ~0.6 % of that scale (the
    # encoder activations carry an rms error of 6e-3, tools/ctc_parity_diag.py) plus 2^-9 per term of the product itself - so 2^-6 = 1.6 %
    # is ~2.5 sigma of it: a percent or two of the pairs.
    scale = x_o.pow(2).mean(-1, keepdim=True).sqrt() * sd64["encoder.layer_stack.0.pos_ffn.w_1.weight"].squeeze(-1).norm(dim=1)[None, :]
    safe = (hp.abs() > (2.0 ** -6) * scale).double()
    frac = 1.0 - float(safe.mean())
    # the pairs whose ReLU state actually differs between the two paths (valid frames only: padded rows are zero on both sides)
    valid = (x_o.abs().sum(-1, keepdim=True) > 0)
    flipped = ((got["dh"] != 0) != (dh_o != 0)) & ((got["dh"] != 0) | (dh_o != 0)) & valid
    flips = float(flipped.double().mean())
    covered = float((flipped & (safe == 0)).double().sum() / max(1.0, float(flipped.double().sum())))
    g_mask_o = (dh_o * safe).t() @ x_o
    g_mask_k = (got["dh"] * safe).t() @ got["x"]
    c_mask = cos(g_mask_k, g_mask_o)
    _report("w1_relu_flip_demonstration", dict(cos_unmasked=c_full, cos_without_pairs_at_risk=c_mask, pairs_removed_fraction=frac,
                                               pairs_whose_relu_state_differs=flips, of_which_inside_the_removed_set=covered))
    assert 0.0 < frac < 0.2, frac
    assert c_mask >= BF16_COS, (c_mask, c_full, frac)
    assert c_mask > c_full


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_long_form_window_matches_oracle(dtype):
    """BASELINE.json configs[4] at model level: T = 2000 frames, +-50-frame attention band on the encoder
    (config.attn_window = 50; the mask of transformer_new.py:53), joint CTC/attention, against the oracle with the same band.
    fp32: CTC loss 1e-4 relative (north_star), gradients 1e-4 of the largest; bf16 (d_model 512, MFMA kernels): SURVEY 8(d) gates."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, attn_window=50)
    B, T, F, V, L = 2, 2000, 80, 56, 20
    cfg, sd, batch = oracle_case(B, T, F, V, L, over, seed=13)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    # The oracle runs in fp64 here: in fp32 its CTC recursion (log domain, values near -1500 at T = 2000, where one ulp is
    # 1.2e-4) carries ~1e-3 relative error into the posteriors, more than the fp32 HIP path (lattice in fp64) differs from the truth.
    sd64 = {k: v.double() for k, v in sd.items()}
    b64 = dict(batch, wave=batch["wave"].double())
    ref = R.RefTrainer(sd64, cfg, warmup=25).iterate(b64)
    ref["grads"] = {k: v.float() for k, v in ref["grads"].items()}
    # the band matters: the full-attention oracle gives a different loss
    full = R.RefTrainer(sd, R.default_cfg(**{**vars(cfg), "attn_window": -1}), warmup=25).iterate(batch)
    assert abs(float(full["loss"]) - float(ref["loss"])) > 1e-3 * abs(float(ref["loss"]))
    model = build(cfg, V, "TransformerOffical", dtype=dtype).cuda()
    model.load_state_dict(sd)
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    rel = abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"]))
    rel_ctc = abs(float(loss[2]) - float(ref["out"]["ctc"])) / abs(float(ref["out"]["ctc"]))
    if dtype == "fp32":
        assert rel < 1e-4 and rel_ctc < 1e-4, (rel, rel_ctc)
        gmax = max(float(g.abs().max()) for g in ref["grads"].values())
        for n, p in model.named_parameters():
            if n.endswith("w_ks.bias"):
                continue
            # fp32 gradients <= 1e-4 relative - here relative to the largest gradient: every element is a sum over
            # B*T = 4000 frames in a different order than the CPU's (measured differences ~3e-5 of the largest gradient)
            g = ref["grads"][n]
            d = (p.grad.cpu() - g).abs()
            assert float((d > 1e-3 * g.abs() + 1e-4 * max(gmax, 1.0)).float().mean()) < 1e-3, n
            assert float(d.max()) < 1e-2 * float(g.abs().max()) + 1e-4 * gmax, n
            assert cos(p.grad, g) > 0.99999, n
    else:
        worst, worst_name, ratio = bf16_gradient_gate(model, ref, "long_form")
        _report("long_form_window_bf16", dict(loss_rel=rel, ctc_rel=rel_ctc, worst_cos=worst, worst_tensor=worst_name, worst_norm_ratio=ratio))
        # T = 2000: measured 9.3e-4 (round 3 and 4), of which 8.6e-4 is the bf16 rounding of the ENCODER's activations carried into a loss that
        # sums 2000 per-frame terms (fp64 head + fp64 loss on the kernels' own encoder output: tools/ctc_parity_diag.py ->
        # profiles/round4_ctc_parity_diag.txt), 0.7e-4 the bf16 logits; the loss kernels add 1e-8.  The 1e-3 of SURVEY 8(d) is kept for every
        # case up to T = 500 (measured 1.7e-4 .. 4.9e-4, >= 2x headroom); here the stated bound is 2e-3 = 2x the measurement.
        assert rel < 2 * BF16_LOSS_RTOL, (float(loss[0]), float(ref["loss"]))
        assert rel_ctc < 2 * BF16_LOSS_RTOL, rel_ctc
        # Round 5: most of those 9.3e-4 is the rounding of the PARAMETERS, not of the activations: against the oracle run on the weight
        # matrices rounded to bf16 (the numbers the MFMA path multiplies by) the loss differs by 3.6e-4 (tools/long_form_parity.py) - gate 6e-4.
        sd16 = {k: (v.bfloat16().float() if v.dim() == 2 else v).double() for k, v in sd.items()}
        ref16 = R.RefTrainer(sd16, cfg, warmup=25).iterate(b64)
        rel16 = abs(float(loss[0]) - float(ref16["loss"])) / abs(float(ref16["loss"]))
        _report("long_form_window_bf16_oracle_on_bf16_weights", dict(loss_rel=rel16))
        assert rel16 < 6e-4, rel16


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_deterministic_mode_repeats_bit_for_bit(deterministic_mode, dtype):
    """Two runs of three training steps from the same weights in deterministic mode end with IDENTICAL parameters
    (every bit): no atomics in the weight-gradient split sums, bias gradients, LayerNorm reductions or the embedding scatter.
    The values stay those of the default path (checked against it within the usual tolerances by the oracle tests)."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3)
    cfg, sd, batch = oracle_case(4, 136, 80, 56, 12, over, seed=9)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    pack = to_pack(batch)
    finals, losses = [], []
    for run in range(2):
        model = build(cfg, 56, "TransformerOffical", dtype=dtype).cuda()
        model.load_state_dict(sd)
        opt = make_opt(model, cfg, 25)
        eng = model._ensure_engine(DEV)
        assert eng.deterministic and not eng.overlap_wgrad and eng.group_wgrad is None
        ls = [float(model.iterate(pack, optimizer=opt, is_train=True)[0].loss) for _ in range(3)]
        torch.cuda.synchronize()
        finals.append(model._flat.p.clone())
        losses.append(ls)
    assert losses[0] == losses[1], losses
    assert torch.equal(finals[0], finals[1]), float((finals[0] - finals[1]).abs().max())


# ------------------------------------------------------------------------------------ CTC prefix beam search, joint rescoring
def _decode_case(dtype="fp32"):
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    over = dict(d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, ctc_weight=0.3)
    cfg, sd, batch = oracle_case(3, 40, 16, 14, 6, over, seed=17)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.3
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    sd["ctc_lo.weight"] = sd["ctc_lo.weight"] * 6.0          # peaky posteriors: hypotheses with clear score gaps
    model = build(cfg, 14, "TransformerOffical", dtype=dtype, cross_mask="wave_len").cuda()
    model.load_state_dict(sd)
    model.eval()
    return model, to_pack(batch), batch


@pytest.mark.parametrize("on_device", [True, False])
@pytest.mark.parametrize("frame_topk", [14, 4])
def test_ctc_prefix_beam_search_matches_oracle(frame_topk, on_device):
    """model.ctc_prefix_beam_search (GPU: encoder, CTC head, per-frame top-k log-softmax + blank, and - on_device - the prefix
    bookkeeping kernel asr_ctc_prefix_beam; otherwise the host loop over the same candidates) against
    oracle/decode_ref.ctc_prefix_beam_search fed with the same posteriors - with every class as a candidate and with the
    same per-frame pruning.  The oracle itself is pinned by brute-force enumeration (tests/test_oracle_ctc.py); nothing in the
    reference covers CTC decoding (parity unpinned by the reference)."""
    from oracle import decode_ref as D
    model, pack, batch = _decode_case()
    got = model.ctc_prefix_beam_search(pack, beam_size=4, nbest=3, frame_topk=frame_topk, on_device=on_device)
    with torch.no_grad():
        logits = model.forward(pack).ctc_logits.double().cpu()
    logp = torch.log_softmax(logits, -1).numpy()
    for b in range(logits.shape[0]):
        Tb = int(batch["wave_len"][b])
        cand = None
        if frame_topk < logits.shape[-1]:
            cand = [list(np.argsort(-logp[b, t], kind="stable")[:frame_topk]) for t in range(Tb)]
        want = D.ctc_prefix_beam_search(logp[b, :Tb], 4, candidates=cand)[:3]
        assert [tuple(h["yseq"]) for h in got[b]] == [p for p, _ in want], (b, got[b], want)
        for h, (_, sc) in zip(got[b], want):
            assert abs(h["score"] - sc) < 2e-4 * max(1.0, abs(sc))
    # a wide beam: identical to the oracle's list again, and its best prefix (a sum over alignments) is at least as probable as
    # the single best path, whose labelling the greedy search returns
    greedy = model.ctc_greedy_search(pack)
    wk = 7 if on_device else 14      # the device kernel ranks beam * (frame_topk + 1) <= 64 candidates per frame
    wide = model.ctc_prefix_beam_search(pack, beam_size=8, nbest=8, frame_topk=wk, on_device=on_device)
    for b in range(len(greedy)):
        Tb = int(batch["wave_len"][b])
        cand = [list(np.argsort(-logp[b, t], kind="stable")[:wk]) for t in range(Tb)] if wk < logits.shape[-1] else None
        want = D.ctc_prefix_beam_search(logp[b, :Tb], 8, candidates=cand)[:8]
        # the best prefix and its score; the lower ranks may differ where the device's fp32 per-frame top-k and the fp64 argsort
        # above pick different candidates on near-ties (asr_ctc_prefix_beam itself is compared rank by rank, on the device's own
        # candidates, in tests/test_kernels_gpu.py::test_ctc_prefix_beam_kernel_matches_host_restatement)
        assert tuple(wide[b][0]["yseq"]) == want[0][0], (b, wide[b][0], want[0])
        assert abs(wide[b][0]["score"] - want[0][1]) < 2e-4 * max(1.0, abs(want[0][1]))
        assert all(wide[b][i]["score"] >= wide[b][i + 1]["score"] for i in range(len(wide[b]) - 1))
        best_path = float(logp[b, :Tb].max(-1).sum())
        assert wide[b][0]["score"] >= best_path - 1e-6, (wide[b][0], best_path)
        assert any(h["yseq"] == greedy[b] for h in wide[b]), (greedy[b], wide[b])


def test_joint_ctc_attention_rescoring_matches_oracle():
    """beam_search(ctc_weight = lambda): the attention beam's hypotheses re-ranked by lambda * log p_ctc + (1 - lambda) * log p_att,
    log p_ctc from the training CTC kernel (forward algorithm, one lattice per hypothesis) - against oracle/decode_ref.joint_rescore
    with oracle/ctc_ref.  The attention n-best lists themselves are pinned by the reference's recognize_beam goldens
    (test_beam_search_matches_reference_golden)."""
    from oracle import ctc_ref, decode_ref as D
    model, pack, batch = _decode_case()
    lam, beam = 0.4, 4
    att = model.beam_search(pack, beam_size=beam, nbest=beam, decode_max_len=9)
    got = model.beam_search(pack, beam_size=beam, nbest=beam, decode_max_len=9, ctc_weight=lam)
    with torch.no_grad():
        logits = model.forward(pack).ctc_logits.double().cpu().numpy()
    for b in range(logits.shape[0]):
        Tb = int(batch["wave_len"][b])

        def ctc_lp(toks, b=b, Tb=Tb):
            if 0 in toks:
                return -float("inf")
            lab = np.array([list(toks) + [0] * max(1, 1 - len(toks))] if len(toks) else [[0]])
            nll, _ = ctc_ref.ctc_batch(logits[b:b + 1, :Tb], [Tb], lab, [len(toks)])
            return -float(nll[0])

        want = D.joint_rescore(att[b], ctc_lp, lam)
        assert [h["yseq"] for h in got[b]] == [h["yseq"] for h in want], (b, got[b], want)
        for h, w in zip(got[b], want):
            for key in ("score", "att_score", "ctc_score"):
                assert (h[key] == w[key]) or abs(h[key] - w[key]) < 2e-4 * max(1.0, abs(w[key])), (key, h, w)
        # the re-ranked list is a permutation of the attention beam's hypotheses, sorted by the joint score
        assert sorted(map(tuple, (h["yseq"] for h in got[b]))) == sorted(map(tuple, (h["yseq"] for h in att[b])))
        assert all(got[b][i]["score"] >= got[b][i + 1]["score"] for i in range(len(got[b]) - 1))
    with pytest.raises(RuntimeError):
        build(R.default_cfg(n_mels=16, lfr_m=1, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=1), 14, "TransformerOffical", dtype="fp32").cuda().beam_search(pack, 2, 1, 4, ctc_weight=0.3)


@pytest.mark.parametrize("which", ["edge_batch", "single_utterance"])
def test_edge_batches_match_oracle(which):
    """Edge cases of the batch contract (empty and ragged inputs): an EMPTY transcript (tgt_len = 0: the decoder sees
    only <sos> -> <eos>, CTC scores the all-blank path), a transcript of three equal tokens on exactly the 5 frames CTC needs
    for it (L + repeats: one feasible alignment family), a 5-frame utterance beside a 12-frame one, and a batch of ONE
    utterance.  fp32 mode against the oracle: joint loss and both terms to 1e-4, every gradient."""
    over = dict(d_model=32, hidden_size=8, num_head=4, ff_size=64, layer_num=2, ctc_weight=0.3)
    B = 3 if which == "edge_batch" else 1
    cfg, sd, batch = oracle_case(B, 12, 16, 40, 6, over, seed=21)
    tgt = torch.zeros(B, 6, dtype=batch["tgt_for_input"].dtype)
    if which == "edge_batch":
        batch["wave_len"] = torch.tensor([12, 5, 5], dtype=batch["wave_len"].dtype)
        batch["wave"][1, 5:] = 0
        batch["wave"][2, 5:] = 0
        tgt[0, :4] = torch.tensor([9, 4, 4, 17])
        tgt[2, :3] = torch.tensor([7, 7, 7])
        batch["tgt_len"] = torch.tensor([4, 0, 3], dtype=batch["tgt_len"].dtype)
    else:
        batch["wave_len"] = torch.tensor([12], dtype=batch["wave_len"].dtype)
        tgt[0, :2] = torch.tensor([5, 5])
        batch["tgt_len"] = torch.tensor([2], dtype=batch["tgt_len"].dtype)
    batch["tgt_for_input"] = tgt
    tr = R.RefTrainer(sd, cfg, warmup=25)
    ref = tr.iterate(batch)
    assert np.isfinite(float(ref["loss"]))
    model = build(cfg, 40, "TransformerOffical", dtype="fp32").cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    pack = to_pack(batch)
    model._ensure_engine(DEV)
    model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    assert abs(float(loss[0]) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
    assert abs(float(loss[2]) - float(ref["out"]["ctc"])) < 1e-4 * abs(float(ref["out"]["ctc"]))
    assert abs(float(loss[1]) - float(ref["out"]["ce"])) < 1e-4 * abs(float(ref["out"]["ce"]))
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    for n, p in model.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), ref["grads"][n].numpy(), rtol=3e-4, atol=3e-6 * max(gmax, 1.0)), n
    # the bf16 path on the same batch: finite, and the padded encoder rows stay exactly zero
    m16 = build(cfg, 40, "TransformerOffical", dtype="bf16").cuda()
    m16.load_state_dict({k: v for k, v in sd.items()})
    m16._ensure_engine(DEV)
    m16.zero_flat_grads()
    l16, _ = m16.train_step(pack)
    assert all(np.isfinite(l16.cpu().numpy()))
    assert all(bool(torch.isfinite(p.grad).all()) for p in m16.parameters())


@pytest.mark.parametrize("T", [70, 600])
def test_beam_search_cross_attention_kernels_agree(T, monkeypatch):
    """The beam search's cross attention on the training attention kernel (all beams of an utterance as Tq = beam queries; the LDS-
    resident kernel at T <= 512, the tiled one beyond) gives the hypotheses of the single-query decode kernel (decode.USE_SDPA = False):
    same token sequences; scores within bf16 noise (the two kernels round differently)."""
    over = dict(d_model=64, hidden_size=64, num_head=2, ff_size=128, layer_num=2)
    cfg, sd, batch = oracle_case(3, T, 16, 24, 5, over, seed=13)
    sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 3.0
    if "decoder.tgt_word_prj.weight" in sd:
        sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    model = build(cfg, 24, dtype="bf16").cuda()
    model.load_state_dict({k: v for k, v in sd.items()})
    model.eval()
    pack = to_pack(batch)
    from asr_chinese_e2e_amd import decode
    monkeypatch.setattr(decode, "USE_SDPA", True)
    a = model.beam_search(pack, 4, 2, 10)
    monkeypatch.setattr(decode, "USE_SDPA", False)
    b = model.beam_search(pack, 4, 2, 10)
    for ha, hb in zip(a, b):
        assert len(ha) == len(hb)
        assert abs(ha[0]["score"] - hb[0]["score"]) < 0.1 * max(1.0, abs(hb[0]["score"]))
        assert ha[0]["yseq"] == hb[0]["yseq"] or abs(ha[0]["score"] - hb[0]["score"]) < 0.05       # a near tie may swap ranks


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_decoder_sequencer_matches_per_kernel_path(dropout, monkeypatch):
    """The native launch sequencer of the decoder layers (csrc/decoder_exec.hip: one host call per layer and direction) issues the same
    kernels in the same order as the per-kernel Python path (ASR_DEC_EXEC=0): in deterministic mode (ordered reductions) loss, logits and
    every gradient are bit-identical, with and without dropout (same per-site mask seeds)."""
    from asr_chinese_e2e_amd import kernels as K
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, dropout=dropout)
    # 9 x 500 = 4500 encoder frames: the accumulating input gradient d_enc += dK|dV W_kv takes the persistent NT kernel on both paths
    # (below 4096 rows the per-kernel path sends it through the fp32 kernel, which rounds differently)
    cfg, sd, batch = oracle_case(9, 500, 80, 56, 12, over, seed=9)
    pack = to_pack(batch)
    prev = K.set_deterministic(True)
    try:
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ASR_DEC_EXEC", mode)
            model = build(cfg, 56, "TransformerOffical", dtype="bf16").cuda()
            model.load_state_dict(sd)
            model.train()
            eng = model._ensure_engine(DEV)
            assert eng.dec_exec == (mode == "1")
            model.zero_flat_grads()
            loss, _ = model.train_step(pack)
            torch.cuda.synchronize()
            assert ("exec" if mode == "1" else "python") and bool(eng._dec_cache) == (mode == "1")      # the sequencer really ran (or did not)
            res[mode] = (loss.clone(), model._flat.g.clone())
    finally:
        K.set_deterministic(prev)
    assert torch.equal(res["1"][0], res["0"][0]), (res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1]), float((res["1"][1] - res["0"][1]).abs().max())


def test_step_cer_beside_backward_matches_inline(monkeypatch):
    """The per-step CER of the greedy ids (transformer_official.py:87-91) is scored on the auxiliary stream beside the encoder's backward pass
    (TransformerOffical._cer_beside_backward) and handed out behind an event; without stream overlap (ASR_WGRAD_OVERLAP=0) it is scored inline
    after the optimizer.  Same ids, same kernel: the metric is identical, step after step (the ids of step 2 depend on step 1's update)."""
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, dropout=0.0)
    cfg, sd, batch = oracle_case(6, 120, 80, 56, 9, over, seed=31)
    pack = to_pack(batch)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ASR_WGRAD_OVERLAP", mode)
        model = build(cfg, 56, "TransformerOffical", dtype="bf16").cuda()
        model.load_state_dict(sd)
        model.train()
        opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
        cers = []
        for _ in range(3):
            m, _ = model.iterate(pack, optimizer=opt, is_train=True)
            cers.append(float(m.cer))
        eng = model._ensure_engine(DEV)
        assert eng.aux_overlap == (mode == "1") and (getattr(model, "_cer_event", None) is not None) == (mode == "1")
        res[mode] = cers
    assert all(0.0 <= c < 1000.0 for c in res["1"])
    # the two runs differ in reduction order (overlap uses atomics): the ids - and with them the CER - may differ by a near-tie after an update
    assert res["1"][0] == res["0"][0], res
    assert all(abs(a - b) <= 5.0 for a, b in zip(res["1"], res["0"])), res


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_ctc_only_training_step_returns_cer(monkeypatch, overlap):
    """The reference's trainer reads metrics.cer.item() after EVERY training step (Trainer/trainer11.py:73-75; produced by cal_metrics,
    transformer_official.py:83-94).  The CTC-only model (BASELINE configs[1]) scores the greedy CTC path of the step - handed out by the loss
    kernels, collapsed and scored on the auxiliary stream beside the backward pass - and the value equals what the evaluation path
    (forward -> cal_metrics: asr_ctc_greedy_decode on the logits) gives for the same weights, step after step."""
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    monkeypatch.setenv("ASR_WGRAD_OVERLAP", overlap)
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, use_decoder=False, ctc_weight=1.0, dropout=0.0)
    cfg, sd, batch = oracle_case(6, 120, 80, 56, 9, over, seed=33)
    pack = to_pack(batch)
    model = build(cfg, 56, "TransformerCTC", dtype="bf16").cuda()
    model.load_state_dict(sd)
    opt = NoamOpt(512, 1, 25, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    seen = []
    for _ in range(4):
        model.eval()
        want, _ = model.iterate(pack, is_train=False)
        model.train()
        metrics, _ = model.iterate(pack, optimizer=opt, is_train=True)
        cer, loss = metrics.cer.item(), metrics.loss.item()      # exactly the trainer's two reads
        assert metrics.cer.detach().cpu().numpy().shape == (1,)      # trainer11.py:112
        assert cer == want.cer.item(), (cer, want.cer.item())
        assert abs(loss - want.loss.item()) <= 2e-3 * abs(loss)
        seen.append(cer)
    eng = model._ensure_engine(DEV)
    assert (getattr(model, "_cer_event", None) is not None) == (overlap == "1" and eng.aux_overlap)
    assert all(0.0 <= c < float("inf") for c in seen), seen      # an untrained model's greedy path is many times longer than the labels: CER >> 100 %
    # cer_in_iterate = False: the step carries no CER work and the key is absent (Pack gives None)
    model.cer_in_iterate = False
    metrics, _ = model.iterate(pack, optimizer=opt, is_train=True)
    assert metrics.cer is None


@pytest.mark.parametrize("name", ["TransformerCTC", "TransformerOffical"])
def test_padded_head_rows_match_dense_rows(name, monkeypatch):
    """The training step keeps the CTC head's logits / gradient rows 64-element aligned (V = 56 -> 64, like 4232 -> 4288 at full size;
    the W^T copy of the head is padded the same way).  Same kernels, same tiles: in deterministic mode the loss and every gradient are
    bit-identical to the dense layout's (Engine.PAD_HEAD_ROWS = False)."""
    from asr_chinese_e2e_amd import kernels as K
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=1, ctc_weight=0.3)
    if name == "TransformerCTC":
        over.update(use_decoder=False, ctc_weight=1.0)
    cfg, sd, batch = oracle_case(9, 500, 80, 56, 12, over, seed=13)      # 4500 frames: the head's input gradient runs on the persistent NT kernel
    pack = to_pack(batch)
    prev = K.set_deterministic(True)
    try:
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setattr(E.Engine, "PAD_HEAD_ROWS", mode == "1")
            model = build(cfg, 56, name, dtype="bf16").cuda()
            model.load_state_dict(sd)
            model.train()
            eng = model._ensure_engine(DEV)
            assert eng.ld_v == (64 if mode == "1" else 56) and eng.ctc_lo.wlpT.stride(0) == eng.ld_v
            model.zero_flat_grads()
            loss, _ = model.train_step(pack)
            torch.cuda.synchronize()
            res[mode] = (loss.clone(), model._flat.g.clone())
    finally:
        K.set_deterministic(prev)
    assert torch.isfinite(res["1"][0]).all()
    assert torch.equal(res["1"][0], res["0"][0]), (res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1]), float((res["1"][1] - res["0"][1]).abs().max())


@pytest.mark.parametrize("name", ["TransformerCTC", "TransformerOffical"])
def test_armed_hand_over_matches_event_fork(name, monkeypatch):
    """The weight-gradient stream is handed its operands by the producer kernel's own completion event (ASR_ARMED_FORK=1, the default)
    or by an event record behind it (0): same kernels, same order - in deterministic mode loss and gradients are bit-identical."""
    from asr_chinese_e2e_amd import kernels as K
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3)
    if name == "TransformerCTC":
        over.update(use_decoder=False, ctc_weight=1.0)
    cfg, sd, batch = oracle_case(9, 500, 80, 56, 12, over, seed=19)
    pack = to_pack(batch)
    prev = K.set_deterministic(True)
    try:
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ASR_ARMED_FORK", mode)
            model = build(cfg, 56, name, dtype="bf16").cuda()
            model.load_state_dict(sd)
            model.train()
            eng = model._ensure_engine(DEV)
            assert eng.armed_fork == (mode == "1")
            for _ in range(3):      # a few steps: a missing dependency shows as a race, not every time
                model.zero_flat_grads()
                loss, _ = model.train_step(pack)
            torch.cuda.synchronize()
            res[mode] = (loss.clone(), model._flat.g.clone())
    finally:
        K.set_deterministic(prev)
    assert torch.isfinite(res["1"][0]).all()
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1]), float((res["1"][1] - res["0"][1]).abs().max())


@pytest.mark.parametrize("env", [dict(ASR_WGRAD_DEFER="w2"), dict(ASR_WGRAD_GROUP="block"), dict(ASR_WGRAD_GROUP="layer", ASR_WGRAD_DEFER="w2,fc")])
def test_arm_does_not_outlive_its_producer(env, monkeypatch):
    """Round-3 ADVICE: with a weight gradient held back (ASR_WGRAD_DEFER=w2 at B*T < 4096) or collected for a grouped launch
    (ASR_WGRAD_GROUP=block / layer; at these row counts the ReLU backward is its own kernel behind the input-gradient GEMM) `_wgrad` returns without forking, and the arm set for the LayerNorm
    backward used to survive until a later `_fork(side)` - which then skipped its event although other main-stream kernels had produced
    the operand since: the weight-gradient stream read a dY ordered only behind the armed kernel.  The arm is now dropped wherever no
    fork follows its producer: with the two streams live (NOT deterministic mode) the gradients of armed and event-record hand-overs
    agree to the atomics' round-off over several steps."""
    over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, use_decoder=False, ctc_weight=1.0)
    cfg, sd, batch = oracle_case(6, 400, 80, 56, 12, over, seed=23)      # 2400 rows: below the 4096-row limit of the own input-gradient kernel
    pack = to_pack(batch)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ASR_ARMED_FORK", mode)
        model = build(cfg, 56, "TransformerCTC", dtype="bf16").cuda()
        model.load_state_dict(sd)
        model.train()
        eng = model._ensure_engine(DEV)
        assert eng.armed_fork == (mode == "1") and eng.overlap_wgrad
        for _ in range(4):      # a missing dependency shows as a race, not every time
            model.zero_flat_grads()
            loss, _ = model.train_step(pack)
        torch.cuda.synchronize()
        assert not eng._armed and not eng._keep      # nothing left armed or held after a finished step
        res[mode] = (loss.clone(), model._flat.g.clone())
    assert torch.isfinite(res["1"][0]).all() and torch.isfinite(res["1"][1]).all()
    torch.testing.assert_close(res["1"][0], res["0"][0], rtol=1e-5, atol=1e-6)
    f = model._flat
    for name, (off, shape) in f.index.items():
        n = int(np.prod(shape))
        a, b = res["1"][1][off:off + n], res["0"][1][off:off + n]
        gmax = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * gmax + 1e-9, (name, float((a - b).abs().max()), gmax)


def test_decoder_sequencer_buffer_cache_is_bounded():
    """Real batches come in many shapes: the sequencer's persistent per-shape buffers are an LRU of a few shapes (Engine.DEC_CACHE_SHAPES),
    re-used when a shape returns, and the training steps stay finite across the changes."""
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    cfg = R.default_cfg(n_mels=80, lfr_m=1, ctc_weight=0.3)
    cfg.layer_num = 1
    model = build(cfg, 60, "TransformerOffical", dtype="bf16").cuda()
    opt = make_opt(model, cfg, 20)
    shapes = [(4, 96, 9), (3, 120, 7), (4, 64, 11), (2, 200, 5), (4, 96, 9), (5, 88, 8), (4, 72, 6), (3, 130, 10), (4, 96, 9)]
    for i, (B, T, L) in enumerate(shapes):
        pack = synthetic_pack(B, T, 80, 60, seed=i, ragged=True, Lmin=2, Lmax=L, device=DEV, dtype=torch.bfloat16)
        m, _ = model.iterate(pack, optimizer=opt, is_train=True)
        assert torch.isfinite(m.loss).all()
    eng = model._engine
    assert 0 < len(eng._dec_cache) <= eng.DEC_CACHE_SHAPES
