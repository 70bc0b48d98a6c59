#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself on CPU (this container only).

TEST INFRASTRUCTURE - never imported by the product package.

The reference (zqs01/ASR_chinese_e2e, mounted read-only at /root/reference) has no tests and
no fixtures of its own, so every parity claim is pinned by vectors this
script produces from the reference's own CPU PyTorch path.  The reference never travels to the
GPU box: only the small .npz files written under tests/golden/ do.

Missing third-party modules of the reference that are NOT on the Transformer/CE hot path are
replaced by inert stubs: torchaudio, python_speech_features, librosa,
seaborn, fire, tensorboard.  `Levenshtein` is replaced by a real pure-Python edit distance
(only CER uses it).

Usage:  python oracle/gen_golden.py            # writes tests/golden/{model_small_*,ops}.npz
        python oracle/gen_golden.py beam | augment | mfma     # the other fixtures, one group each
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    ta = mod("torchaudio")
    ta.transforms = mod("torchaudio.transforms")
    mod("python_speech_features", mfcc=None, delta=None, logfbank=None)
    mod("librosa")
    mod("seaborn")
    mod("fire")
    tb = mod("torch.utils.tensorboard")

    class SummaryWriter:  # never used by the generator
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter

    def distance(a, b):
        # classic two-row Levenshtein over characters
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]

    mod("Levenshtein", distance=distance)


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: later in-place ops (clip, Adam) must not alias


def make_vocab(Vocab, size):
    v = Vocab()
    # ids 0-3 are PAD/UNK/BOS/EOS (vocab.py:10-17); fill the rest with distinct CJK chars
    for i in range(size - 4):
        v._token2id[chr(0x4E00 + i)] = len(v._token2id)
    v._id2token = [k for k in v._token2id]
    assert v.vocab_size == size
    return v


def make_batch(Pack, B, T, F, Lmax, V, wave_len, tgt_len, seed):
    g = torch.Generator().manual_seed(seed)
    wave = torch.randn(B, T, F, generator=g)
    tgt = torch.zeros(B, Lmax, dtype=torch.long)
    for b in range(B):
        wave[b, wave_len[b]:] = 0.0
        tgt[b, :tgt_len[b]] = torch.randint(4, V, (tgt_len[b],), generator=g)
    p = Pack()
    p.add(wave=wave, tgt_for_input=tgt, tgt_for_metric=tgt.clone(),
          wave_len=torch.tensor(wave_len, dtype=torch.long),
          tgt_len=torch.tensor(tgt_len, dtype=torch.long))
    return p


def model_case(name, cfg, B, T, Lmax, V, wave_len, tgt_len, seed, warm_up=25):
    from Predictor import Models
    from Predictor.data_handler import Vocab
    from Predictor.Utils import Pack
    from Trainer.optimizer import NoamOpt

    torch.manual_seed(seed)
    Model = Models.TransformerOffical
    ModelConfig = Model.get_default_config()
    config = ModelConfig()
    config.fn_build(dict(cfg))
    vocab = make_vocab(Vocab, V)
    model = Model(config, vocab)
    model.train()  # dropout=0 in cfg, so train() is deterministic
    # make LayerNorm gains/biases and linear biases non-trivial so parity tests see them
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))

    F = cfg["n_mels"] * cfg["lfr_m"]
    pack = make_batch(Pack, B, T, F, Lmax, V, wave_len, tgt_len, seed + 1)

    out = {}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        if k.endswith("positional_encoding.pe"):
            out["pe_head/" + k] = _np(v[:, :64])  # full buffer is formula-defined; keep 64 rows
        else:
            out["sd/" + k] = _np(v)
    for k in ("wave", "tgt_for_input", "wave_len", "tgt_len"):
        out["in/" + k] = _np(pack[k])

    # forward pieces
    enc_out = model.encoder(pack.wave, pack.wave_len)[0]
    ys_in, ys_out = model.decoder.preprocess(pack.tgt_for_input)
    output = model.forward(pack)
    metrics = model.cal_metrics(output, pack)
    out["fwd/enc_out"] = _np(enc_out)
    out["fwd/ys_in"] = _np(ys_in)
    out["fwd/ys_out"] = _np(ys_out)
    out["fwd/pred"] = _np(output.pred)
    out["fwd/gold"] = _np(output.gold)
    out["fwd/loss"] = _np(metrics.loss)
    out["fwd/cer"] = _np(metrics.cer)

    # one full training iterate: zero_grad, backward, clip 5.0, Noam+Adam step
    adam = torch.optim.Adam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9)
    opt = NoamOpt(config.d_model, 1, warm_up, adam)
    opt.zero_grad()
    output = model.forward(pack)
    metrics = model.cal_metrics(output, pack)
    metrics.loss.backward()
    seen = set()
    for n, p in model.named_parameters():
        out["grad/" + n] = _np(p.grad)
        seen.add(n)
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    out["step/grad_norm"] = _np(total_norm)
    opt.step()
    out["step/lr"] = np.float64(opt._rate)
    for n, p in model.named_parameters():
        out["step/" + n] = _np(p)
    # second iterate through the public entry point, to pin Adam state handling
    m2, _ = model.iterate(pack, optimizer=opt, is_train=True)
    out["step2/loss"] = _np(m2.loss)
    out["step2/lr"] = np.float64(opt._rate)
    for n, p in model.named_parameters():
        out["step2/" + n] = _np(p)
    out["cfg/keys"] = np.array(sorted(cfg.keys()))
    out["cfg/vals"] = np.array([float(cfg[k]) for k in sorted(cfg.keys())])
    out["cfg/warm_up"] = np.int64(warm_up)
    out["cfg/V"] = np.int64(V)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB  loss={float(metrics.loss):.6f} "
          f"cer={float(metrics.cer):.3f} gnorm={float(total_norm):.5f}")


def mfma_case():
    """The MFMA geometry (the reference's DEFAULT width, transformer_official.py:115-122: d_model 512, 8 x 64 heads, ff 1024; one
    encoder + one decoder layer, B = 3, T = 140 ragged) through the reference: weights and batch are functions of numpy seeds
    (oracle/golden_inputs.py), only what the reference computes from them is stored - logits, sampled encoder rows, loss, CER,
    per parameter SAMPLES gradient elements + norm + a full-tensor probe checksum, the clip norm, and the loss after one step."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_inputs as GI
    from Predictor import Models
    from Predictor.data_handler import Vocab
    from Predictor.Utils import Pack
    from Trainer.optimizer import NoamOpt

    case = GI.MFMA_CASE
    Model = Models.TransformerOffical
    config = Model.get_default_config()()
    config.fn_build(dict(case["cfg"]))
    model = Model(config, make_vocab(Vocab, case["V"]))
    model.train()
    sd = {k: torch.from_numpy(v) for k, v in GI.mfma_state_dict(case).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("positional_encoding.pe") for k in missing), (missing, unexpected)
    assert model.decoder.tgt_word_prj.weight is model.decoder.tgt_word_emb.weight      # still tied after loading
    b = GI.mfma_batch(case)
    pack = Pack()
    pack.add(wave=torch.from_numpy(b["wave"]), tgt_for_input=torch.from_numpy(b["tgt_for_input"]), tgt_for_metric=torch.from_numpy(b["tgt_for_input"]).clone(),
             wave_len=torch.from_numpy(b["wave_len"]), tgt_len=torch.from_numpy(b["tgt_len"]))
    out = {}
    enc_out = model.encoder(pack.wave, pack.wave_len)[0]
    output = model.forward(pack)
    metrics = model.cal_metrics(output, pack)
    out["fwd/enc_rows"] = np.arange(0, case["T"], 7)
    out["fwd/enc_out_rows"] = _np(enc_out[:, ::7])
    out["fwd/enc_out_abs_sum"] = np.float64(enc_out.double().abs().sum())
    out["fwd/pred"] = _np(output.pred)
    out["fwd/gold"] = _np(output.gold)
    out["fwd/loss"] = _np(metrics.loss)
    out["fwd/cer"] = _np(metrics.cer)
    adam = torch.optim.Adam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9)
    opt = NoamOpt(config.d_model, 1, case["warm_up"], adam)
    opt.zero_grad()
    metrics.loss.backward()
    for n, p in model.named_parameters():
        g = p.grad.detach().double().flatten().numpy()
        out["grad_s/" + n] = g[GI.sample_index(n, g.size, case)].astype(np.float32)
        out["grad_norm/" + n] = np.float64(np.sqrt((g * g).sum()))
        out["grad_probe/" + n] = np.float64(g @ GI.probe(n, g.size, case))
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    out["step/grad_norm"] = _np(total_norm)
    opt.step()
    out["step/lr"] = np.float64(opt._rate)
    m2, _ = model.iterate(pack, optimizer=opt, is_train=True)
    out["step2/loss"] = _np(m2.loss)
    path = os.path.join(OUT, case["name"] + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB  loss={float(metrics.loss):.6f} cer={float(metrics.cer):.3f} "
          f"gnorm={float(total_norm):.5f} loss2={float(m2.loss):.6f}")


def op_cases():
    """Per-op goldens: attention, FFN, PE, mask builders, preprocess, loss, Noam, LFR, CER, config."""
    from Predictor.Models.attention import MultiHeadAttention, ScaledDotProductAttention
    from Predictor.Models.module import PositionalEncoding, PositionwiseFeedForwardUseConv
    from Predictor.Models import utils as U
    from Predictor.Utils.loss import cal_loss, cal_performance
    from Predictor.Utils.score import calculate_cer
    from Predictor.data_handler.processor import build_LFR_features
    from Predictor.data_handler.padder import Padder
    from Predictor.data_handler import Vocab
    from Predictor.Bases import BaseConfig
    from Trainer.optimizer import NoamOpt

    out = {}
    torch.manual_seed(7)
    # --- ScaledDotProductAttention with a key-pad mask (attention.py:74-86)
    q, k, v = torch.randn(6, 9, 16), torch.randn(6, 11, 16), torch.randn(6, 11, 16)
    klen = [11, 7, 1, 11, 7, 1]
    mask = torch.zeros(6, 9, 11, dtype=torch.bool)
    for i, l in enumerate(klen):
        mask[i, :, l:] = True
    sdpa = ScaledDotProductAttention(temperature=4.0, attn_dropout=0.0)
    o, a = sdpa(q, k, v, mask=mask)
    out.update({"sdpa/q": _np(q), "sdpa/k": _np(k), "sdpa/v": _np(v), "sdpa/klen": np.array(klen),
                "sdpa/out": _np(o), "sdpa/attn": _np(a)})
    # --- MultiHeadAttention (attention.py:33-62), self-attention with key-pad mask
    mha = MultiHeadAttention(4, 32, 8, 8, dropout=0.0)
    x = torch.randn(3, 10, 32, requires_grad=True)
    lens = [10, 6, 1]
    m = torch.zeros(3, 10, 10, dtype=torch.bool)
    for i, l in enumerate(lens):
        m[i, :, l:] = True
    y, _ = mha(x, x, x, mask=m)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    out.update({"mha/x": _np(x), "mha/lens": np.array(lens), "mha/y": _np(y), "mha/w": _np(w),
                "mha/dx": _np(x.grad)})
    for n, p in mha.named_parameters():
        out["mha/sd/" + n] = _np(p)
        out["mha/grad/" + n] = _np(p.grad)
    # --- FFN with Conv1d k=1 (module.py:58-75)
    ffn = PositionwiseFeedForwardUseConv(32, 48, dropout=0.0)
    x = torch.randn(3, 10, 32, requires_grad=True)
    y = ffn(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    out.update({"ffn/x": _np(x), "ffn/y": _np(y), "ffn/w": _np(w), "ffn/dx": _np(x.grad)})
    for n, p in ffn.named_parameters():
        out["ffn/sd/" + n] = _np(p)
        out["ffn/grad/" + n] = _np(p.grad)
    # --- PositionalEncoding (module.py:8-33)
    pe = PositionalEncoding(32, max_len=200)
    out["pe/d32"] = _np(pe.pe[0, :200])
    pe512 = PositionalEncoding(512, max_len=5000)
    out["pe/d512_rows"] = np.array([0, 1, 2, 499, 1999, 4999])
    out["pe/d512"] = _np(pe512.pe[0, [0, 1, 2, 499, 1999, 4999]])
    # --- masks (utils.py:100-144)
    padded = torch.randn(3, 7, 5)
    L = torch.tensor([7, 4, 1])
    out["mask/lens"] = _np(L)
    out["mask/non_pad"] = _np(U.get_non_pad_mask(padded, input_lengths=L))
    out["mask/attn_pad"] = _np(U.get_attn_pad_mask(padded, L, 4)).astype(np.uint8)
    seq = torch.tensor([[2, 5, 6, 3, 3], [2, 9, 3, 3, 3], [2, 4, 5, 6, 7]])
    out["mask/seq"] = _np(seq)
    out["mask/subseq"] = _np(U.get_subsequent_mask(seq))
    out["mask/keypad"] = _np(U.get_attn_key_pad_mask(seq, seq, 3)).astype(np.uint8)
    out["mask/non_pad_idx"] = _np(U.get_non_pad_mask(seq, pad_idx=3))
    # --- cal_loss (Utils/loss.py:26-51), both branches
    pred = torch.randn(12, 17)
    gold = torch.tensor([5, 0, 3, 16, 0, 0, 1, 2, 9, 4, 0, 8])
    out.update({"loss/pred": _np(pred), "loss/gold": _np(gold),
                "loss/ce": _np(cal_loss(pred, gold, 0.0)),
                "loss/ce_smooth01": _np(cal_loss(pred, gold, 0.1))})
    l, nc = cal_performance(pred.view(3, 4, 17), gold.view(3, 4))
    out["loss/n_correct"] = np.int64(nc)
    # --- NoamOpt.rate (Trainer/optimizer.py:24-28)
    steps = [1, 2, 100, 3999, 4000, 4001, 100000]
    no = NoamOpt(512, 1, 4000, None)
    out["noam/steps"] = np.array(steps)
    out["noam/rate_512_4000"] = np.array([no.rate(s) for s in steps], dtype=np.float64)
    no = NoamOpt(32, 2.0, 25, None)
    out["noam/rate_32_25_f2"] = np.array([no.rate(s) for s in steps], dtype=np.float64)
    # --- LFR (processor.py:74-100)
    for T in (1, 2, 3, 4, 7, 10, 11, 12):
        x = np.arange(T * 3, dtype=np.float32).reshape(T, 3) + 0.5
        out[f"lfr/T{T}_m4n3"] = build_LFR_features(x, 4, 3)
    x = np.random.RandomState(0).randn(9, 4).astype(np.float32)
    out["lfr/x9"] = x
    out["lfr/x9_m1n1"] = build_LFR_features(x, 1, 1)
    out["lfr/x9_m3n1"] = build_LFR_features(x, 3, 1)
    out["lfr/x9_m1n2"] = build_LFR_features(x, 1, 2)
    # --- CER string convention (score.py:4-13 + vocab.py:75-79)
    vocab = make_vocab(Vocab, 12)
    hyp = [[4, 5, 6, 0, 0], [7, 7, 8, 9, 3], [4, 0, 0, 0, 0], [5, 6, 3, 0, 0]]
    ref = [[4, 5, 7, 3, 0], [7, 8, 9, 3, 0], [10, 11, 3, 0, 0], [5, 6, 3, 0, 0]]
    hs = [vocab.convert_id2str(i) for i in hyp]
    rs = [vocab.convert_id2str(i) for i in ref]
    out["cer/hyp"] = np.array(hyp)
    out["cer/ref"] = np.array(ref)
    out["cer/vals"] = np.array([calculate_cer(a, b) for a, b in zip(hs, rs)], dtype=np.float64)
    # --- Padder (padder.py:7-27)
    o2, l2 = Padder.pad_two([[4, 5, 6], [7], [8, 9]], 0)
    out["pad/two"] = _np(o2)
    out["pad/two_len"] = np.array(l2)
    o3, l3 = Padder.pad_tri([torch.ones(3, 2), 2 * torch.ones(1, 2), 3 * torch.ones(2, 2)], 0)
    out["pad/tri"] = _np(o3)
    out["pad/tri_len"] = np.array(l3)
    # --- Vocab id conventions (vocab.py:10-17, 55-66)
    v2 = Vocab()
    v2.consume_sentance_list(["你好你", "好的"])
    v2.build()
    out["vocab/size"] = np.int64(v2.vocab_size)
    out["vocab/ids_plain"] = np.array(v2.convert_str("你好吗", use_bos=False, use_eos=False))
    out["vocab/ids_boseos"] = np.array(v2.convert_str("你好吗"))

    # --- BaseConfig semantics (base_config.py:7-15, 37-46): unknown keys are added, not rejected
    class C(BaseConfig):
        a = 1
        b = 2

    class D(BaseConfig):
        b = 5
        c = 7

    c = C()
    c.fn_build({"a": 3, "zzz": 9})
    c.fn_combine(D())
    out["config/abc_zzz"] = np.array([c.a, c.b, c.c, c.zzz])
    path = os.path.join(OUT, "ops.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB")


def beam_case(name, cfg, B, T, V, wave_len, seed, beam, nbest, decode_max_len):
    """Golden hypotheses of the reference's own beam search (Decoder.recognize_beam,
    transformer_official.py:331-434) on a small random model, one utterance at a time as the
    reference does.  Stored: weights, inputs, encoder output, and per utterance the n-best token
    sequences (with sos/eos) and scores."""
    import contextlib
    import io
    from Predictor import Models
    from Predictor.data_handler import Vocab
    from Predictor.Utils import Pack

    torch.manual_seed(seed)
    Model = Models.TransformerOffical
    config = Model.get_default_config()()
    config.fn_build(dict(cfg))
    vocab = make_vocab(Vocab, V)
    model = Model(config, vocab)
    model.eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
        # sharpen the output distribution a little so that hypotheses end (eos) before maxlen sometimes
        model.decoder.tgt_word_emb.weight.mul_(3.0)
    F = cfg["n_mels"] * cfg["lfr_m"]
    pack = make_batch(Pack, B, T, F, 3, V, wave_len, [1] * B, seed + 1)
    out = {}
    for k, v in model.state_dict().items():
        if k.endswith("positional_encoding.pe"):
            out["pe_head/" + k] = _np(v[:, :64])
        else:
            out["sd/" + k] = _np(v)
    out["in/wave"] = _np(pack.wave)
    out["in/wave_len"] = _np(pack.wave_len)
    # torch >= 1.2 rejects the uint8 mask recognize_beam builds (masked_fill wants bool; the training
    # path converts with .gt(0), this one does not): cast that ONE mask to bool here - same values,
    # the reference's files are untouched.  Without it the reference's beam search cannot run at all.
    import Predictor.Models.transformer_official as TO
    orig_mask = TO.get_subsequent_mask
    TO.get_subsequent_mask = lambda seq: orig_mask(seq).bool()
    char_list = [str(i) for i in range(V)]
    args = types.SimpleNamespace(beam_size=beam, nbest=nbest, decode_max_len=decode_max_len)
    with torch.no_grad():
        enc_out = model.encoder(pack.wave, pack.wave_len)[0]
        out["fwd/enc_out"] = _np(enc_out)
        for b in range(B):
            with contextlib.redirect_stdout(io.StringIO()):
                hyps = model.decoder.recognize_beam(enc_out[b, : int(pack.wave_len[b])], char_list, args)
            L = max(len(h["yseq"]) for h in hyps)
            seqs = np.zeros((len(hyps), L), dtype=np.int64)
            for i, h in enumerate(hyps):
                seqs[i, : len(h["yseq"])] = h["yseq"]
            out[f"beam/{b}/yseq"] = seqs
            out[f"beam/{b}/len"] = np.array([len(h["yseq"]) for h in hyps], dtype=np.int64)
            out[f"beam/{b}/score"] = np.array([float(h["score"]) for h in hyps], dtype=np.float64)
            print(f"  utt {b}: " + "; ".join(f"{h['yseq']} {float(h['score']):.4f}" for h in hyps))
    TO.get_subsequent_mask = orig_mask
    out["cfg/keys"] = np.array(sorted(cfg.keys()))
    out["cfg/vals"] = np.array([float(cfg[k]) for k in sorted(cfg.keys())])
    out["cfg/V"] = np.int64(V)
    out["cfg/beam"] = np.int64(beam)
    out["cfg/nbest"] = np.int64(nbest)
    out["cfg/decode_max_len"] = np.int64(decode_max_len)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB")


def augment_cases():
    """Golden SpecAugment outputs from the reference's own AudioParser.augment (processor.py:52-58:
    augments.time_mask then augments.freq_mask) on normalised random features, seeded `random`."""
    import random
    from Predictor.data_handler.augments import freq_mask, time_mask
    out = {}
    g = torch.Generator().manual_seed(77)
    cases = [(80, 300, 1), (80, 57, 2), (40, 120, 3), (80, 41, 4), (80, 500, 5), (32, 200, 6)]   # fewer than 30 channels / 40 frames: the reference's randrange can raise
    for n_mels, T, seed in cases:
        f = torch.randn(n_mels, T, generator=g)
        f = (f - f.mean()) / f.std()                    # what normalize() hands to augment()
        random.seed(seed)
        y = freq_mask(time_mask(f.unsqueeze(0))).squeeze(0)
        out[f"aug/{n_mels}_{T}_{seed}/in"] = _np(f)
        out[f"aug/{n_mels}_{T}_{seed}/out"] = _np(y)
    path = os.path.join(OUT, "augment.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB")


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    base = dict(n_mels=20, lfr_m=1, d_model=32, hidden_size=8, ff_size=64, num_head=4,
                dropout=0.0, layer_num=2)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":   # only the SpecAugment fixtures
        augment_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mfma":      # only the d_model = 512 fixture (the others stay byte-identical)
        mfma_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "beam":     # only the beam-search fixtures (the others stay byte-identical)
        beam_case("beam_small", base, B=3, T=20, V=12, wave_len=[20, 13, 7], seed=31, beam=3, nbest=3, decode_max_len=0)
        beam_case("beam_small_maxlen", dict(base, layer_num=1), B=2, T=16, V=9, wave_len=[16, 10], seed=37, beam=4, nbest=2, decode_max_len=6)
        return
    # ragged case: one full-length row, one length-1 target, short utterances
    model_case("model_small_ragged", base, B=4, T=24, Lmax=7, V=30,
               wave_len=[24, 17, 9, 13], tgt_len=[7, 3, 1, 5], seed=11)
    # all-full-length case with a different head geometry (d_k != d_model / n_head)
    cfg2 = dict(base, d_model=48, hidden_size=16, num_head=2, ff_size=40, layer_num=1, n_mels=8, lfr_m=2)
    model_case("model_small_full", cfg2, B=3, T=12, Lmax=4, V=21,
               wave_len=[12, 12, 12], tgt_len=[4, 4, 4], seed=23)
    op_cases()


if __name__ == "__main__":
    main()
