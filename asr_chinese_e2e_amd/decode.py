"""Beam search of the attention decoder on the GPU.

Same search as the reference's Decoder.recognize_beam (Predictor/Models/transformer_official.py:
331-434) - see include/asr_hip.h for the exact rules - but batched over utterances x beams, one
token per live hypothesis per step through key/value caches, instead of re-running the whole
decoder over the growing prefix for every hypothesis in a Python loop.

Device state per step: scores / alive flags / last tokens (B, beam); per layer a self-attention
K|V cache (B*beam, Lcap, 2*H*dk) in two copies (the step's survivors are gathered by parent into
the other copy); the encoder-side K|V of every layer are projected once.  The hypotheses
themselves are (token, parent) records per step; the n-best lists are rebuilt on the host by
following the parents, in the order the reference's lists would have (steps ascending, beam order
inside a step, stable sort by score).
"""
import math
import os

import torch

from . import kernels as K

# cross attention of the beam search on the training attention kernel (all beams of an utterance as Tq = beam queries against the same K / V:
# 11 us) instead of the single-query decode kernel (one wave per (hypothesis, head): ~220 us per layer and step); False: the latter (tests)
USE_SDPA = True

SOS_ID, EOS_ID = 2, 3   # transformer_official.py:53-54
BLANK_ID = 0            # CTC blank = <pad>, as in the training loss


def beam_search(model, input, beam_size=5, nbest=1, decode_max_len=0, check_every=8):
    """input: the reference's batch Pack (wave, wave_len).  Returns, per utterance, a list of at most
    `nbest` dicts {'yseq': [sos, ..., eos], 'score': float} - recognize_beam's result format."""
    if beam_size < 1 or beam_size > 8:
        raise ValueError("beam_size must be in 1..8 (asr_beam_step merges beam*beam <= 64 candidates per wave)")
    eng = model._ensure_engine(input.wave.device)
    if not eng.use_decoder:
        raise RuntimeError("this model has no attention decoder; use ctc_greedy_search")
    was_training, eng.training = eng.training, False
    try:
        with torch.no_grad():
            return _search(model, eng, input, int(beam_size), int(nbest), int(decode_max_len), int(check_every))
    finally:
        eng.training = was_training


def _search(model, eng, input, beam, nbest, decode_max_len, check_every):
    x = input.wave.to(eng.dtype).contiguous()
    wave_len = input.wave_len.to(torch.int32).contiguous()
    B, T, _ = x.shape
    dev = x.device
    d, H, dk, L, V = eng.d, eng.H, eng.dk, eng.L, eng.V
    hd = H * dk
    enc, _ = eng.encoder_fwd(x, wave_len, model.attn_window)            # (B*T, d)
    maxlen = wave_len.clone() if decode_max_len == 0 else torch.full_like(wave_len, decode_max_len)
    Lcap = int(maxlen.max())
    if Lcap > eng.pe.shape[0]:
        raise ValueError(f"decode length {Lcap} exceeds the positional-encoding table ({eng.pe.shape[0]})")
    R = B * beam
    # encoder-side keys / values of every layer, once
    cross_kv = [cross.kv.fwd(enc) for _, cross, _ in eng.dec]            # (B*T, 2hd)
    caches = [torch.zeros(L, R, Lcap, 2 * hd, dtype=eng.dtype, device=dev) for _ in range(2)]
    score = torch.zeros(B, beam, dtype=torch.float32, device=dev)
    alive = torch.zeros(B, beam, dtype=torch.int32, device=dev)
    alive[:, 0] = 1                                                       # one hypothesis [sos] per utterance
    last_tok = torch.full((B, beam), SOS_ID, dtype=torch.int32, device=dev)
    parent = torch.zeros(B, beam, dtype=torch.int32, device=dev)
    rec_tok = torch.zeros(Lcap, B, beam, dtype=torch.int32, device=dev)
    rec_par = torch.zeros_like(rec_tok)
    rec_end = torch.zeros_like(rec_tok)
    rec_score = torch.full((Lcap, B, beam), float("-inf"), dtype=torch.float32, device=dev)
    alive_total = torch.zeros(Lcap, dtype=torch.int32, device=dev)
    row_bytes = 2 * hd * caches[0].element_size()
    cur = 0
    steps_done = 0
    use_sdpa = dk == 64 and eng.dtype == torch.bfloat16 and USE_SDPA
    o_buf = torch.empty(R, hd, dtype=eng.dtype, device=dev) if use_sdpa else None
    lse_buf = None
    for i in range(Lcap):
        cache = caches[cur]
        y = K.embed_pe_fwd(last_tok.reshape(-1), eng.emb32, eng.pe[i:i + 1], d ** -0.5, R, 1, eng.dtype)   # :369-371
        for l, (slf, cross, ffn) in enumerate(eng.dec):
            q = slf.q.fwd(y)
            slf.kv.fwd(y, out=cache[l, :, i, :])                          # this step's key | value straight into the cache
            kv = cache[l].view(R * Lcap, 2 * hd)
            o = K.decode_attn(q, kv[:, :hd], kv[:, hd:], H, dk, Lcap, kv_div=1, k_len_uniform=i + 1)
            y, _, _ = K.add_ln_fwd(slf.fc.fwd(o), y, slf.ln.g, slf.ln.b, None, None, R, 1)
            q = cross.q.fwd(y)
            ckv = cross_kv[l]
            if use_sdpa:
                # the beams of an utterance are `beam` query rows against the SAME encoder keys / values: that is the training
                # attention kernel at Tq = beam (one workgroup per (utterance, head), K / V fetched once for all beams, MFMA) -
                # 11 us against ~220 us for the one-wave-per-(row, head) decode kernel, which was 48 of the 65 ms of a search
                o, lse_buf = K.sdpa_fwd(q, ckv[:, :hd], ckv[:, hd:], wave_len, B, H, beam, T, dk, o=o_buf, lse=lse_buf)
            else:
                o = K.decode_attn(q, ckv[:, :hd], ckv[:, hd:], H, dk, T, kv_div=beam, k_len=wave_len, len_div=beam)
            y, _, _ = K.add_ln_fwd(cross.fc.fwd(o), y, cross.ln.g, cross.ln.b, None, None, R, 1)
            h = ffn.w1.fwd(y, act=1)
            y, _, _ = K.add_ln_fwd(ffn.w2.fwd(h), y, ffn.ln.g, ffn.ln.b, None, None, R, 1)
        logits = eng.prj.fwd(y)                                           # tied projection, no bias (:379)
        vals, ids = K.logsoftmax_topk(logits, beam)
        K.beam_step(vals, ids, score, alive, last_tok, parent, rec_tok, rec_par, rec_end, rec_score, maxlen, alive_total[i:i + 1],
                    B, beam, i, EOS_ID)
        steps_done = i + 1
        if i + 1 < Lcap:
            K.cache_gather(caches[cur], caches[cur ^ 1], parent.reshape(-1), L, R, beam, Lcap, i + 1, row_bytes)
            cur ^= 1
        if (i % check_every) == check_every - 1 and int(alive_total[i]) == 0:   # the only host sync of the loop
            break
    return _backtrace(rec_tok[:steps_done].cpu(), rec_par[:steps_done].cpu(), rec_end[:steps_done].cpu(), rec_score[:steps_done].cpu(), B, beam, nbest)


def _backtrace(rec_tok, rec_par, rec_end, rec_score, B, beam, nbest):
    steps = rec_tok.shape[0]
    # plain nested lists: element access on a tensor costs ~1 us each, and this walk touches ~50 k of them per batch
    # (33 of the 97 ms of a B = 32, beam 5 search before)
    rec_tok, rec_par, rec_end, rec_score = rec_tok.tolist(), rec_par.tolist(), rec_end.tolist(), rec_score.tolist()
    out = []
    for b in range(B):
        ended = []                                           # in the order the reference appends to ended_hyps
        for i in range(steps):
            end_i = rec_end[i][b]
            for k in range(beam):
                e = end_i[k]
                if not e:
                    continue
                seq, kk = [], k
                for s in range(i, -1, -1):
                    seq.append(rec_tok[s][b][kk])
                    kk = rec_par[s][b][kk]
                seq = [SOS_ID] + seq[::-1] + ([EOS_ID] if e == 2 else [])
                ended.append((rec_score[i][b][k], seq))
        ended = sorted(ended, key=lambda h: h[0], reverse=True)[: min(len(ended), nbest)]
        out.append([{"yseq": seq, "score": sc} for sc, seq in ended])
    return out


# --------------------------------------------------------------------------------------------- CTC prefix beam search
def _logadd(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    m = a if a > b else b
    return m + math.log(math.exp(a - m) + math.exp(b - m))


def ctc_prefix_beam_search(model, input, beam_size=5, nbest=1, frame_topk=10, on_device=None):
    """CTC prefix beam search (Hannun et al. 2014, algorithm 1 without a language model) over the CTC head's posteriors:
    per utterance a list of at most `nbest` dicts {'yseq': [ids], 'score': log p(yseq | x)}, best first.
    Everything runs on the GPU: encoder, CTC projection, per frame the `frame_topk` best classes with their log-softmax values
    plus the blank's (asr_ctc_frame_topk), and the prefix bookkeeping itself (asr_ctc_prefix_beam: one wave per utterance,
    prefixes as trie nodes; 2 ms per batch of 32 x 500 frames against 630 ms for the host loop below).  on_device=False (or a
    beam / top-k beyond the kernel's beam * (frame_topk + 1) <= 64) selects the host loop over the same candidates - the
    restatement the kernel is tested against, written like the reference's own search (a host loop, transformer_official.py:358-420).
    SURVEY.md 8(f) rank 1; the reference has no CTC (its greedy_search / beam_search are empty stubs, :106-110)."""
    eng = model._ensure_engine(input.wave.device)
    if not eng.use_ctc:
        raise RuntimeError("this model has no CTC head (config.ctc_weight = 0)")
    was_training, eng.training = eng.training, False
    try:
        with torch.no_grad():
            out = model.forward(input)
    finally:
        eng.training = was_training
    logits = out.ctc_logits                                    # (B, T, V)
    B, T, V = logits.shape
    k = max(1, min(int(frame_topk), V))
    vals, ids, blank_lp = K.ctc_frame_topk(logits.reshape(B * T, V), k, BLANK_ID)
    fits = beam_size * (k + 1) <= 64 and beam_size <= 16 and nbest <= beam_size
    if on_device is None:
        on_device = fits
    if on_device:
        if not fits:
            raise ValueError(f"the device search ranks beam * (frame_topk + 1) <= 64 candidates per frame (beam {beam_size}, frame_topk {k})")
        tok, ln, sc = K.ctc_prefix_beam(vals, ids, blank_lp, input.wave_len.to(torch.int32).contiguous(), B, T, beam_size, nbest, BLANK_ID)
        tok, ln, sc = tok.cpu().tolist(), ln.cpu().tolist(), sc.cpu().tolist()
        return [[{"yseq": tok[b][r][:ln[b][r]], "score": sc[b][r]} for r in range(nbest) if ln[b][r] >= 0] for b in range(B)]
    vals, ids, blank_lp = vals.view(B, T, k).cpu().tolist(), ids.view(B, T, k).cpu().tolist(), blank_lp.view(B, T).cpu().tolist()
    lens = input.wave_len.cpu().tolist()
    results = []
    for b in range(B):
        beam = {(): (0.0, -math.inf)}                        # prefix -> (log p ending in blank, log p ending in a symbol)
        for t in range(int(lens[b])):
            lb = blank_lp[b][t]
            nxt = {}
            for prefix, (pb, pnb) in beam.items():
                tot = _logadd(pb, pnb)
                cur = nxt.setdefault(prefix, [-math.inf, -math.inf])
                cur[0] = _logadd(cur[0], tot + lb)
                last = prefix[-1] if prefix else None
                for c, lp in zip(ids[b][t], vals[b][t]):
                    if c == BLANK_ID:
                        continue
                    if c == last:
                        cur = nxt.setdefault(prefix, [-math.inf, -math.inf])
                        cur[1] = _logadd(cur[1], pnb + lp)       # repeated symbol, no blank in between: same prefix
                        new = nxt.setdefault(prefix + (c,), [-math.inf, -math.inf])
                        new[1] = _logadd(new[1], pb + lp)        # after a blank: a new symbol
                    else:
                        new = nxt.setdefault(prefix + (c,), [-math.inf, -math.inf])
                        new[1] = _logadd(new[1], tot + lp)
            ranked = sorted(nxt.items(), key=lambda kv: _logadd(kv[1][0], kv[1][1]), reverse=True)[:beam_size]
            beam = {p: (v[0], v[1]) for p, v in ranked}
        final = sorted(((p, _logadd(v[0], v[1])) for p, v in beam.items()), key=lambda kv: kv[1], reverse=True)[:nbest]
        results.append([{"yseq": list(p), "score": sc} for p, sc in final])
    return results


# --------------------------------------------------------------------------------------------- joint CTC / attention rescoring
def joint_beam_search(model, input, beam_size=5, nbest=1, decode_max_len=0, ctc_weight=0.3):
    """Two-pass joint decoding (Watanabe et al. 2017): the attention decoder's beam search proposes `beam_size` hypotheses per
    utterance, the CTC head scores each of them with the forward algorithm (the training kernel asr_ctc_fwd_bwd without the
    gradient, one lattice per hypothesis), and the list is re-ranked by ctc_weight * log p_ctc + (1 - ctc_weight) * log p_att.
    Returns per utterance at most `nbest` dicts {'yseq', 'score', 'att_score', 'ctc_score'}.  Hypotheses that contain the blank
    id (the attention decoder may emit <pad>) cannot be spelled by CTC: ctc_score = -inf."""
    eng = model._ensure_engine(input.wave.device)
    if not (eng.use_ctc and eng.use_decoder):
        raise RuntimeError("joint rescoring needs a model with both the attention decoder and the CTC head (0 < config.ctc_weight < 1)")
    hyps = beam_search(model, input, beam_size, beam_size, decode_max_len)
    was_training, eng.training = eng.training, False
    try:
        with torch.no_grad():
            logits = model.forward(input).ctc_logits          # (B, T, V)
    finally:
        eng.training = was_training
    B, T, V = logits.shape
    dev = logits.device
    n = beam_size
    labels, ok = [], []
    for b in range(B):
        for j in range(n):
            toks = hyps[b][j]["yseq"][1:-1] if j < len(hyps[b]) else []
            good = j < len(hyps[b]) and BLANK_ID not in toks and len(toks) <= 255
            labels.append(toks if good else [])
            ok.append(good)
    Lmax = max(1, max(len(l) for l in labels))
    lab = torch.zeros(B * n, Lmax, dtype=torch.int32)
    for r, l in enumerate(labels):
        if l:
            lab[r, : len(l)] = torch.tensor(l, dtype=torch.int32)
    lab_len = torch.tensor([len(l) for l in labels], dtype=torch.int32)
    rep = logits.repeat_interleave(n, dim=0).contiguous()     # one copy of the utterance's lattice input per hypothesis
    in_len = input.wave_len.to(torch.int32).repeat_interleave(n).contiguous()
    nll, _ = K.ctc_fwd_bwd(rep, in_len, lab.to(dev), lab_len.to(dev), eng.ws, blank=BLANK_ID, want_grad=False)
    nll = nll.cpu().tolist()
    out = []
    for b in range(B):
        cands = []
        for j, h in enumerate(hyps[b]):
            ctc = -nll[b * n + j] if ok[b * n + j] else -math.inf
            cands.append(dict(yseq=h["yseq"], att_score=h["score"], ctc_score=ctc, score=ctc_weight * ctc + (1.0 - ctc_weight) * h["score"]))
        out.append(sorted(cands, key=lambda c: c["score"], reverse=True)[:nbest])
    return out
