"""CPU restatement of the decoding algorithms added beyond the reference (TEST INFRASTRUCTURE: only tests/ may import it).

The reference has no CTC at all and leaves `greedy_search` / `beam_search` as empty stubs
(Predictor/Models/transformer_official.py:106-110); its only search is the attention beam search
`Decoder.recognize_beam` (:331-434), restated in oracle/ref_model.beam_search and pinned by the reference's own n-best lists.
SURVEY.md 8(f) rank 1 asks for CTC prefix beam search and joint CTC/attention rescoring on top of it: there is nothing in the
reference to pin them to ("parity unpinned by the reference"); this file restates the published algorithms

  * CTC prefix beam search: Hannun et al., "First-Pass Large Vocabulary Continuous Speech Recognition using Bi-Directional
    Recurrent DNNs" (2014), algorithm 1 without a language model: per prefix the probabilities of ending in blank / non-blank;
  * joint rescoring: Watanabe et al., "Hybrid CTC/Attention Architecture for End-to-End Speech Recognition" (2017), the
    two-pass form: score = lambda * log p_ctc(y | x) + (1 - lambda) * log p_att(y | x) over the attention decoder's n-best list,

and tests/test_oracle_ctc.py pins the prefix search against brute-force enumeration of every alignment on tiny lattices.
"""
import itertools
import math

import numpy as np

NEG = -float("inf")


def logadd(*xs):
    m = max(xs)
    if m == NEG:
        return NEG
    return m + math.log(sum(math.exp(x - m) for x in xs))


def ctc_prefix_beam_search(logp, beam_size, blank=0, candidates=None):
    """logp: (T, V) log-probabilities of one utterance.  candidates: optional list per frame of class ids to extend with (the
    device path prunes to the k best classes per frame; None = every class).  Returns [(prefix tuple, log p(prefix))] best first."""
    T, V = logp.shape
    beam = {(): (0.0, NEG)}          # prefix -> (log p ending in blank, log p ending in non-blank)
    for t in range(T):
        nxt = {}

        def acc(prefix, idx, val):
            cur = nxt.setdefault(prefix, [NEG, NEG])
            cur[idx] = logadd(cur[idx], val)

        cand = range(V) if candidates is None else candidates[t]
        for prefix, (pb, pnb) in beam.items():
            acc(prefix, 0, logadd(pb, pnb) + logp[t, blank])
            for c in cand:
                c = int(c)
                if c == blank:
                    continue
                lp = logp[t, c]
                if prefix and c == prefix[-1]:
                    acc(prefix, 1, pnb + lp)                 # repeated symbol without a blank in between: same prefix
                    acc(prefix + (c,), 1, pb + lp)           # after a blank: a new symbol
                else:
                    acc(prefix + (c,), 1, logadd(pb, pnb) + lp)
        ranked = sorted(nxt.items(), key=lambda kv: logadd(*kv[1]), reverse=True)[:beam_size]
        beam = {k: tuple(v) for k, v in ranked}
    return [(k, logadd(*v)) for k, v in sorted(beam.items(), key=lambda kv: logadd(*kv[1]), reverse=True)]


def ctc_label_logprob_bruteforce(logp, labels, blank=0):
    """log p(labels | x) by enumerating every alignment (tiny T and V only)."""
    T, V = logp.shape
    tot = NEG
    for path in itertools.product(range(V), repeat=T):
        col, prev = [], None
        for c in path:
            if c != prev and c != blank:
                col.append(c)
            prev = c
        if tuple(col) == tuple(labels):
            tot = logadd(tot, sum(logp[t, c] for t, c in enumerate(path)))
    return tot


def best_labelling_bruteforce(logp, blank=0, max_len=None):
    """argmax over label sequences of the total alignment probability (tiny cases)."""
    T, V = logp.shape
    scores = {}
    for path in itertools.product(range(V), repeat=T):
        col, prev = [], None
        for c in path:
            if c != prev and c != blank:
                col.append(c)
            prev = c
        key = tuple(col)
        scores[key] = logadd(scores.get(key, NEG), sum(logp[t, c] for t, c in enumerate(path)))
    return sorted(scores.items(), key=lambda kv: kv[1], reverse=True)


def joint_rescore(att_nbest, ctc_logprob_of, ctc_weight):
    """att_nbest: [{'yseq': [sos, ..., eos], 'score': s_att}]; ctc_logprob_of(tokens) -> log p_ctc(tokens | x).
    Returns the list re-ranked by ctc_weight * ctc + (1 - ctc_weight) * att (stable), each entry with att_score / ctc_score."""
    out = []
    for h in att_nbest:
        toks = h["yseq"][1:-1]
        ctc = ctc_logprob_of(toks)
        out.append(dict(yseq=list(h["yseq"]), att_score=float(h["score"]), ctc_score=float(ctc),
                        score=float(ctc_weight * ctc + (1.0 - ctc_weight) * h["score"])))
    return sorted(out, key=lambda h: h["score"], reverse=True)
