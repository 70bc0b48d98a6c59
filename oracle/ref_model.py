"""CPU oracle: functional restatement of the reference training step (plain PyTorch, fp32/fp64).

TEST INFRASTRUCTURE ONLY.  Nothing under asr_chinese_e2e_amd/ imports this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker / the timed
CPU baseline - never as a product code path.

What it restates (file:line are into the reference, zqs01/ASR_chinese_e2e):
  * TransformerOffical.forward / iterate           Predictor/Models/transformer_official.py:68-104
  * Encoder / EncoderLayer                          transformer_official.py:158-213
  * Decoder.preprocess / forward / DecoderLayer     transformer_official.py:260-328, 446-458
  * MultiHeadAttention / ScaledDotProductAttention  Predictor/Models/attention.py:33-86
  * PositionalEncoding / FFN (Conv1d k=1)           Predictor/Models/module.py:8-33, 58-75
  * mask builders                                   Predictor/Models/utils.py:100-144
  * cal_loss (CE, ignore_index 0, smoothing)        Predictor/Utils/loss.py:26-51
  * clip_grad_norm_(5.0) + NoamOpt + Adam           transformer_official.py:100-103,
                                                    Trainer/optimizer.py:15-28, main.py:81-83
  * CER string convention                           Predictor/Utils/score.py:4-13, vocab.py:75-79
It is written over a plain {name: tensor} state dict with the reference's own state_dict keys,
so reference weights (tests/golden/*.npz, made by oracle/gen_golden.py) load directly.

Pinned by: tests/test_oracle_golden.py against vectors produced by the reference itself.

Not in the reference (BASELINE.json north_star asks for it): a CTC head `ctc_lo` on the encoder
output and the joint loss  lambda*CTC + (1-lambda)*CE.  Its oracle is torch's own
F.ctc_loss (ATen LossCTC.cpp, torch 2.10.0) cross-checked by oracle/ctc_ref.py; "parity unpinned"
by the reference (it has no CTC code), pinned by build-generated known-answer vectors.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

PAD_ID, UNK_ID, SOS_ID, EOS_ID = 0, 1, 2, 3  # vocab.py:10-17; transformer_official.py:53-54
LN_EPS = 1e-5                                 # torch.nn.LayerNorm default
CLIP_NORM = 5.0                               # transformer_official.py:102


def default_cfg(**over):
    """Merged model config (transformer_official.py:115-122 + data_config.py:11-16)."""
    cfg = dict(n_mels=80, lfr_m=4, d_model=512, hidden_size=64, ff_size=1024, num_head=8,
               dropout=0.0, layer_num=6, ctc_weight=0.0, cross_mask="ref_compat",
               use_decoder=True, attn_window=-1)
    cfg.update(over)
    return SimpleNamespace(**cfg)


# ----------------------------------------------------------------------------- building blocks
def positional_encoding(length, d_model, dtype=torch.float32):
    """module.py:16-24.  pe[p, 2i] = sin(p * w_i), pe[p, 2i+1] = cos(p * w_i)."""
    pos = torch.arange(0, length).unsqueeze(1).float()
    w = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(pos * w)
    pe[:, 1::2] = torch.cos(pos * w)
    return pe.to(dtype)


def valid_mask(lengths, T, loop=False):
    """True where t < lengths[b].  utils.py:100-115 builds the same thing with a Python loop
    over the batch (`loop=True` keeps that loop, for the faithful CPU baseline timing)."""
    if loop:
        m = torch.ones(len(lengths), T, dtype=torch.bool)
        for i in range(len(lengths)):
            m[i, int(lengths[i]):] = False
        return m
    return torch.arange(T).unsqueeze(0) < lengths.view(-1, 1)


def multi_head_attention(sd, pre, q_in, kv_in, masked, n_head, d_k):
    """attention.py:33-62.  `masked` is (B, Lq, Lk) bool, True = excluded (score -> -inf)."""
    B, Lq, _ = q_in.shape
    Lk = kv_in.shape[1]
    q = F.linear(q_in, sd[pre + "w_qs.weight"], sd[pre + "w_qs.bias"]).view(B, Lq, n_head, d_k)
    k = F.linear(kv_in, sd[pre + "w_ks.weight"], sd[pre + "w_ks.bias"]).view(B, Lk, n_head, d_k)
    v = F.linear(kv_in, sd[pre + "w_vs.weight"], sd[pre + "w_vs.bias"]).view(B, Lk, n_head, d_k)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) / (d_k ** 0.5)          # attention.py:76-77
    # a query row with NO admissible key (only possible with the long-form band: a padded frame more than `window` frames past
    # the end of its utterance) would be softmax(all -inf) = NaN, and NaN * 0 survives the pad zeroing; such a row attends to
    # nothing: context 0, no gradient.  Without a band every row sees the valid keys, as in the reference.
    dead = masked.all(dim=-1, keepdim=True)                              # (B, Lq, 1)
    s = s.masked_fill((masked & ~dead).unsqueeze(1), float("-inf"))      # attention.py:80
    p = torch.softmax(s, dim=-1) * (~dead).unsqueeze(1).to(s.dtype)
    ctx = torch.einsum("bhqk,bkhd->bqhd", p, v).reshape(B, Lq, n_head * d_k)
    out = F.linear(ctx, sd[pre + "fc.weight"], sd[pre + "fc.bias"])
    d = q_in.shape[-1]
    return F.layer_norm(out + q_in, (d,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], LN_EPS)


def feed_forward(sd, pre, x):
    """module.py:68-75.  Conv1d(k=1) weights are (out, in, 1): squeeze -> Linear."""
    w1, w2 = sd[pre + "w_1.weight"].squeeze(-1), sd[pre + "w_2.weight"].squeeze(-1)
    h = F.linear(F.relu(F.linear(x, w1, sd[pre + "w_1.bias"])), w2, sd[pre + "w_2.bias"])
    d = x.shape[-1]
    return F.layer_norm(h + x, (d,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], LN_EPS)


def encoder_forward(sd, cfg, wave, wave_len, loop_masks=False):
    """transformer_official.py:158-189."""
    B, T, _ = wave.shape
    keep = valid_mask(wave_len, T, loop_masks)                           # (B,T)
    non_pad = keep.unsqueeze(-1).to(wave.dtype)
    masked = (~keep).unsqueeze(1).expand(B, T, T)                        # key-pad mask only
    w = int(getattr(cfg, "attn_window", -1))
    if w >= 0:   # long-form config: +-w frame band, the mask of transformer_new.py:53 (t.triu(t.tril(mask, 50), -50))
        idx = torch.arange(T)
        masked = masked | ((idx.view(T, 1) - idx.view(1, T)).abs() > w).unsqueeze(0)
    d = cfg.d_model
    x = F.linear(wave, sd["encoder.linear_in.weight"], sd["encoder.linear_in.bias"])
    x = F.layer_norm(x, (d,), sd["encoder.layer_norm_in.weight"], sd["encoder.layer_norm_in.bias"], LN_EPS)
    x = x + positional_encoding(T, d, wave.dtype).unsqueeze(0)           # :175-177 (no pad zeroing here)
    for i in range(cfg.layer_num):
        pre = f"encoder.layer_stack.{i}."
        x = multi_head_attention(sd, pre + "slf_attn.", x, x, masked, cfg.num_head, cfg.hidden_size) * non_pad
        x = feed_forward(sd, pre + "pos_ffn.", x) * non_pad
    return x


def decoder_preprocess(tgt):
    """transformer_official.py:260-275: strip pad 0; ys_in = [sos]+y padded with EOS;
    ys_out = y+[eos] padded with 0.  Width = max stripped length + 1."""
    rows = [r[r != PAD_ID] for r in tgt]
    To = max(len(r) for r in rows) + 1
    ys_in = torch.full((len(rows), To), EOS_ID, dtype=tgt.dtype)
    ys_out = torch.full((len(rows), To), PAD_ID, dtype=tgt.dtype)
    for i, r in enumerate(rows):
        ys_in[i, 0] = SOS_ID
        ys_in[i, 1:len(r) + 1] = r
        ys_out[i, :len(r)] = r
        ys_out[i, len(r)] = EOS_ID
    return ys_in, ys_out


def decoder_forward(sd, cfg, tgt, enc_out, cross_len, loop_masks=False):
    """transformer_official.py:277-328.  `cross_len` is what the reference passes as
    encoder_input_lengths: the TEXT lengths (quirk, :78) - or wave_len for the corrected mask."""
    ys_in, ys_out = decoder_preprocess(tgt)
    B, To = ys_in.shape
    Ti = enc_out.shape[1]
    d = cfg.d_model
    non_pad = ys_in.ne(EOS_ID).unsqueeze(-1).to(enc_out.dtype)           # :292
    causal = torch.triu(torch.ones(To, To, dtype=torch.bool), diagonal=1)
    self_masked = ys_in.eq(EOS_ID).unsqueeze(1) | causal.unsqueeze(0)     # :294-298
    cross_masked = (~valid_mask(cross_len, Ti, loop_masks)).unsqueeze(1).expand(B, To, Ti)  # :301-303
    emb = sd["decoder.tgt_word_emb.weight"]
    x = emb[ys_in] * (d ** -0.5) + positional_encoding(To, d, enc_out.dtype).unsqueeze(0)   # :306-307
    for i in range(cfg.layer_num):
        pre = f"decoder.layer_stack.{i}."
        x = multi_head_attention(sd, pre + "slf_attn.", x, x, self_masked, cfg.num_head, cfg.hidden_size) * non_pad
        x = multi_head_attention(sd, pre + "enc_attn.", x, enc_out, cross_masked, cfg.num_head, cfg.hidden_size) * non_pad
        x = feed_forward(sd, pre + "pos_ffn.", x) * non_pad
    pred = F.linear(x, emb)                                              # :321 tied, no bias, no scaling
    return pred, ys_out


def decoder_step_logits(sd, cfg, ys, enc_out):
    """The decoder pass recognize_beam runs for ONE hypothesis (transformer_official.py:360-379):
    ys (1, i) ids, causal self-attention mask only, NO cross-attention mask, all-ones non_pad;
    returns log_softmax of the last position (V,)."""
    i = ys.shape[1]
    d = cfg.d_model
    emb = sd["decoder.tgt_word_emb.weight"]
    causal = torch.triu(torch.ones(i, i, dtype=torch.bool), diagonal=1).unsqueeze(0)
    nomask = torch.zeros(1, i, enc_out.shape[1], dtype=torch.bool)
    x = emb[ys] * (d ** -0.5) + positional_encoding(i, d, enc_out.dtype).unsqueeze(0)
    for l in range(cfg.layer_num):
        pre = f"decoder.layer_stack.{l}."
        x = multi_head_attention(sd, pre + "slf_attn.", x, x, causal, cfg.num_head, cfg.hidden_size)
        x = multi_head_attention(sd, pre + "enc_attn.", x, enc_out, nomask, cfg.num_head, cfg.hidden_size)
        x = feed_forward(sd, pre + "pos_ffn.", x)
    return F.log_softmax(F.linear(x[:, -1], emb), dim=1)[0]


def beam_search(sd, cfg, enc_out, beam, nbest=1, decode_max_len=0):
    """Decoder.recognize_beam (transformer_official.py:331-434) for one utterance: enc_out (T, d).
    Every live hypothesis is extended by its `beam` best tokens, the best `beam` of all extensions
    survive, those that end in eos leave the beam for the ended list; at the last step eos is
    appended to whatever is left.  Scores are sums of log-softmax values, no length normalisation.
    Returns [(ids incl. sos/eos, score)] sorted by score, at most nbest.  Pinned by the golden
    vectors tests/golden/beam_*.npz (the reference's own search, see oracle/gen_golden.py)."""
    T = enc_out.shape[0]
    maxlen = T if decode_max_len == 0 else decode_max_len
    enc = enc_out.unsqueeze(0)
    hyps = [(0.0, [SOS_ID])]
    ended = []
    for i in range(maxlen):
        kept = []
        for score, seq in hyps:
            lp = decoder_step_logits(sd, cfg, torch.tensor([seq]), enc)
            vals, ids = torch.topk(lp, beam)
            for j in range(beam):
                kept.append((score + float(vals[j]), seq + [int(ids[j])]))
            kept = sorted(kept, key=lambda h: h[0], reverse=True)[:beam]      # stable, as the reference's sorted()
        hyps = kept
        if i == maxlen - 1:
            hyps = [(s_, q + [EOS_ID]) for s_, q in hyps]
        remained = []
        for h in hyps:
            (ended if h[1][-1] == EOS_ID else remained).append(h)
        hyps = remained
        if not hyps:
            break
    ended = sorted(ended, key=lambda h: h[0], reverse=True)[: min(len(ended), nbest)]
    return [(q, s_) for s_, q in ended]


def ce_loss(pred, gold, smoothing=0.0):
    """Utils/loss.py:26-51."""
    pred = pred.reshape(-1, pred.shape[-1])
    gold = gold.reshape(-1)
    if smoothing > 0.0:
        n_class = pred.shape[1]
        one_hot = torch.zeros_like(pred).scatter(1, gold.view(-1, 1), 1)
        one_hot = one_hot * (1 - smoothing) + (1 - one_hot) * smoothing / n_class
        logp = F.log_softmax(pred, dim=1)
        keep = gold.ne(PAD_ID)
        return -(one_hot * logp).sum(dim=1).masked_select(keep).sum() / keep.sum()
    return F.cross_entropy(pred, gold, ignore_index=PAD_ID, reduction="mean")


def ctc_logits(sd, enc_out):
    return F.linear(enc_out, sd["ctc_lo.weight"], sd["ctc_lo.bias"])


def ctc_loss(logits, in_len, labels, lab_len, zero_infinity=False):
    """Sum over utterances of -log p(labels | x), divided by the batch size; blank = 0.
    (New functionality - not in the reference; oracle = torch F.ctc_loss.)"""
    logp = F.log_softmax(logits, dim=-1).transpose(0, 1)                  # (T,B,V)
    total = F.ctc_loss(logp, labels, in_len, lab_len, blank=PAD_ID, reduction="sum",
                       zero_infinity=zero_infinity)
    return total / logits.shape[0]


def forward_losses(sd, cfg, batch, loop_masks=False):
    """TransformerOffical.forward + cal_performance (+ the added CTC branch).
    batch: dict(wave, wave_len, tgt_for_input, tgt_len).  Returns dict of tensors."""
    out = {}
    enc = encoder_forward(sd, cfg, batch["wave"], batch["wave_len"], loop_masks)
    out["enc_out"] = enc
    loss = 0.0
    lam = float(cfg.ctc_weight)
    if cfg.use_decoder:
        cross_len = batch["tgt_len"] if cfg.cross_mask == "ref_compat" else batch["wave_len"]
        pred, gold = decoder_forward(sd, cfg, batch["tgt_for_input"], enc, cross_len, loop_masks)
        out["pred"], out["gold"] = pred, gold
        out["ce"] = ce_loss(pred, gold)
        loss = (1.0 - lam) * out["ce"] if lam > 0 else out["ce"]
    if lam > 0 or not cfg.use_decoder:
        logits = ctc_logits(sd, enc)
        out["ctc_logits"] = logits
        out["ctc"] = ctc_loss(logits, batch["wave_len"], batch["tgt_for_input"], batch["tgt_len"])
        loss = loss + (lam * out["ctc"] if cfg.use_decoder else out["ctc"])
    out["loss"] = loss
    return out


# ----------------------------------------------------------------------------- metrics
def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def ids_to_str(ids, id2token):
    """vocab.py:75-79: drop PAD, join tokens with single spaces."""
    return " ".join(id2token[int(i)] for i in ids if int(i) != PAD_ID)


def cer_percent(pred, gold, id2token, greedy="topk"):
    """transformer_official.py:87-91 + score.py:11-13: character edit distance over the
    SPACE-JOINED strings (spaces count), divided by the token count of the reference."""
    # NOTE: the reference uses pred.topk(1) (transformer_official.py:87).  Decoder rows at padded
    # positions are exactly 0 (masked, no projection bias), so every logit ties and the id topk
    # returns there is implementation-defined (CPU torch 2.10 gives a non-zero id; argmax gives
    # 0 = PAD, which convert_id2str then drops).  The oracle keeps topk to match the reference
    # bit-for-bit on the same torch build; `greedy="argmax"` is the tie-free convention the HIP
    # path implements (first index wins).
    hyp_ids = pred.topk(1)[1].squeeze(-1) if greedy == "topk" else pred.argmax(-1)
    tot = 0.0
    for h, g in zip(hyp_ids, gold):
        hs, gs = ids_to_str(h, id2token), ids_to_str(g, id2token)
        tot += edit_distance(hs, gs) / len(gs.split(" "))
    return tot * 100.0 / len(gold)


def ctc_greedy_decode(logits, lens, blank=0):
    """Best-path CTC decoding (Graves 2006, B(.)): argmax per frame (first index on ties), merge
    repeats, drop blanks.  NOT in the reference (no CTC there): parity unpinned by the reference,
    pinned by hand-made paths in tests/test_oracle_ctc.py.  logits (B,T,V) array-like; returns a
    list of id lists."""
    import numpy as np
    x = np.asarray(logits, dtype=np.float64)
    out = []
    for b in range(x.shape[0]):
        path = x[b, : int(lens[b])].argmax(-1)
        hyp, prev = [], blank
        for p in path:
            p = int(p)
            if p != blank and p != prev:
                hyp.append(p)
            prev = p
        out.append(hyp)
    return out


def ctc_cer_percent(hyps, labels, lab_len, id2token):
    """CER of collapsed CTC hypotheses against the label sequences, with the reference's CER
    convention (score.py:11-13: edit distance over the space-joined strings / reference tokens)."""
    tot = 0.0
    for h, g, n in zip(hyps, labels, lab_len):
        hs, gs = ids_to_str(h, id2token), ids_to_str(list(g[: int(n)]), id2token)
        tot += edit_distance(hs, gs) / max(len(gs.split(" ")), 1)
    return tot * 100.0 / len(hyps)


# ----------------------------------------------------------------------------- optimizer
def noam_rate(step, model_size, warmup, factor=1.0):
    """Trainer/optimizer.py:24-28."""
    return factor * (model_size ** -0.5) * min(step ** -0.5, step * warmup ** -1.5)


def clip_grad_norm(grads, max_norm=CLIP_NORM):
    """torch.nn.utils.clip_grad_norm_ semantics (transformer_official.py:102)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def adam_update(p, g, m, v, step, lr, b1=0.9, b2=0.98, eps=1e-9):
    """torch.optim.Adam single-tensor math with main.py:81's hyper-parameters."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


class RefTrainer:
    """iterate() of the reference, restated over a flat {name: tensor} dict (CPU, autograd).

    Used (a) by parity tests as the expected result of one/two training steps and
    (b) by bench.py as the CPU baseline ("port" of TransformerOffical.iterate incl. the Python
    loop masks and the host CER loop)."""

    def __init__(self, sd, cfg, warmup, id2token=None, factor=1.0):
        self.cfg = cfg
        self.sd = {k: v.clone() for k, v in sd.items()}
        # tied embedding / projection (transformer_official.py:253-256): one tensor, two keys
        self.trainable = [k for k in self.sd if not k.endswith("positional_encoding.pe")
                          and k != "decoder.tgt_word_prj.weight"]
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.trainable}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.trainable}
        self.step_no = 0
        self.warmup, self.factor = warmup, factor
        self.id2token = id2token
        self.last = {}

    def loss_and_grads(self, batch, loop_masks=False):
        leaves = {k: self.sd[k].detach().clone().requires_grad_(True) for k in self.trainable}
        sd = dict(self.sd)
        sd.update(leaves)
        if "decoder.tgt_word_emb.weight" in leaves:
            sd["decoder.tgt_word_prj.weight"] = leaves["decoder.tgt_word_emb.weight"]
        out = forward_losses(sd, self.cfg, batch, loop_masks)
        grads = torch.autograd.grad(out["loss"], [leaves[k] for k in self.trainable], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(self.trainable, grads)}
        return out, grads

    def iterate(self, batch, loop_masks=False, with_cer=False):
        out, grads = self.loss_and_grads(batch, loop_masks)
        assert not torch.isinf(out["loss"])                               # transformer_official.py:88
        cer = None
        if with_cer and self.id2token is not None and "pred" in out:
            cer = cer_percent(out["pred"].detach(), out["gold"], self.id2token)
        total, clipped = clip_grad_norm([grads[k] for k in self.trainable])
        self.step_no += 1
        lr = noam_rate(self.step_no, self.cfg.d_model, self.warmup, self.factor)
        for k, g in zip(self.trainable, clipped):
            self.sd[k], self.m[k], self.v[k] = adam_update(self.sd[k], g, self.m[k], self.v[k], self.step_no, lr)
        if "decoder.tgt_word_emb.weight" in self.sd:
            self.sd["decoder.tgt_word_prj.weight"] = self.sd["decoder.tgt_word_emb.weight"]
        self.last = dict(loss=out["loss"].detach(), grad_norm=total, lr=lr, cer=cer, grads=grads, out=out)
        return self.last


def init_state_dict(cfg, vocab_size, seed=0, dtype=torch.float32):
    """Random-init weights with the reference's key names, shapes and init distributions
    (attention.py:16-28, transformer_official.py:147-156, 242-256).  Used for synthetic benches."""
    g = torch.Generator().manual_seed(seed)
    d, dk, H, ff = cfg.d_model, cfg.hidden_size, cfg.num_head, cfg.ff_size
    d_in = cfg.n_mels * cfg.lfr_m
    sd = {}

    def normal(shape, std):
        return torch.randn(*shape, generator=g, dtype=dtype) * std

    def xavier(o, i):
        return normal((o, i), math.sqrt(2.0 / (o + i)))

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(*shape, generator=g, dtype=dtype) * 2 - 1) * b

    def mha(pre):
        for n in ("w_qs", "w_ks", "w_vs"):
            sd[pre + n + ".weight"] = normal((H * dk, d), math.sqrt(2.0 / (d + dk)))
            sd[pre + n + ".bias"] = uni((H * dk,), d)
        sd[pre + "layer_norm.weight"] = torch.ones(d, dtype=dtype)
        sd[pre + "layer_norm.bias"] = torch.zeros(d, dtype=dtype)
        sd[pre + "fc.weight"] = xavier(d, H * dk)
        sd[pre + "fc.bias"] = uni((d,), H * dk)

    def ffn(pre):
        sd[pre + "w_1.weight"] = uni((ff, d, 1), d)
        sd[pre + "w_1.bias"] = uni((ff,), d)
        sd[pre + "w_2.weight"] = uni((d, ff, 1), ff)
        sd[pre + "w_2.bias"] = uni((d,), ff)
        sd[pre + "layer_norm.weight"] = torch.ones(d, dtype=dtype)
        sd[pre + "layer_norm.bias"] = torch.zeros(d, dtype=dtype)

    sd["encoder.linear_in.weight"] = xavier(d, d_in)
    sd["encoder.linear_in.bias"] = uni((d,), d_in)
    sd["encoder.layer_norm_in.weight"] = torch.ones(d, dtype=dtype)
    sd["encoder.layer_norm_in.bias"] = torch.zeros(d, dtype=dtype)
    sd["encoder.positional_encoding.pe"] = positional_encoding(5000, d, dtype).unsqueeze(0)
    for i in range(cfg.layer_num):
        mha(f"encoder.layer_stack.{i}.slf_attn.")
        ffn(f"encoder.layer_stack.{i}.pos_ffn.")
    if cfg.use_decoder:
        # nn.Embedding default N(0,1); the tied projection's own xavier init is discarded when the
        # weights are shared (transformer_official.py:250-256)
        sd["decoder.tgt_word_emb.weight"] = normal((vocab_size, d), 1.0)
        sd["decoder.positional_encoding.pe"] = positional_encoding(5000, d, dtype).unsqueeze(0)
        for i in range(cfg.layer_num):
            mha(f"decoder.layer_stack.{i}.slf_attn.")
            mha(f"decoder.layer_stack.{i}.enc_attn.")
            ffn(f"decoder.layer_stack.{i}.pos_ffn.")
        sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    if cfg.ctc_weight > 0 or not cfg.use_decoder:
        sd["ctc_lo.weight"] = xavier(vocab_size, d)
        sd["ctc_lo.bias"] = uni((vocab_size,), d)
    return sd
