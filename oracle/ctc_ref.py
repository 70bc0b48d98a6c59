"""CPU oracle for the CTC forward-backward (numpy float64, log domain).

TEST INFRASTRUCTURE ONLY - never imported by the product package.

The reference repository contains NO CTC code: the arithmetic that
BASELINE.json's north_star asks for lives in a third-party dependency of the build,
PyTorch 2.10.0 `torch.nn.functional.ctc_loss` (ATen native/LossCTC.cpp), which restates
Graves et al. 2006, "Connectionist Temporal Classification", eqs. (6)-(16).  This file restates
that published algorithm independently:

  extended label sequence l' = (blank, l1, blank, l2, ..., lL, blank), S = 2L+1
  alpha_1(0) = y_1(blank), alpha_1(1) = y_1(l1), otherwise 0
  alpha_t(s) = y_t(l'_s) * (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [l'_s != blank and l'_s != l'_{s-2}] alpha_{t-1}(s-2))
  beta mirrored from t = T;   p(l|x) = alpha_T(S-1) + alpha_T(S-2)
  d(-ln p)/d logit_t(v) = y_t(v) - (1/p) * sum_{s: l'_s = v} alpha_t(s) beta_t(s) / y_t(l'_s)
      (ATen LossCTC.cpp ctc_loss_backward: res = exp(lp) - exp(log_sum_alpha_beta + nll - lp))
  frames t >= input_length contribute zero gradient.

"Parity unpinned" by the reference (nothing to pin against); pinned by tests/test_oracle_ctc.py:
hand-computed known answers (T=3/L=1, repeated label, empty label, infeasible) and agreement with
torch F.ctc_loss (fp64) on random cases.
"""
import numpy as np

NEG_INF = -np.inf


def _logaddexp3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def ctc_one(logits, labels, blank=0):
    """logits (T,V) float64 (already cut to the utterance's input length), labels (L,) ints.
    Returns (nll, dlogits (T,V), log_alpha (T,S), log_beta (T,S))."""
    T, V = logits.shape
    L = len(labels)
    S = 2 * L + 1
    lp = log_softmax(logits.astype(np.float64))
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = labels
    skip = np.zeros(S, dtype=bool)                       # may take the s-2 transition
    for s in range(2, S):
        skip[s] = ext[s] != blank and ext[s] != ext[s - 2]
    la = np.full((T, S), NEG_INF)
    lb = np.full((T, S), NEG_INF)
    if T == 0:
        return (0.0 if L == 0 else np.inf), np.zeros_like(lp), la, lb
    la[0, 0] = lp[0, blank]
    if S > 1:
        la[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        a0 = la[t - 1]
        a1 = np.concatenate(([NEG_INF], a0[:-1]))
        a2 = np.where(skip, np.concatenate(([NEG_INF, NEG_INF], a0))[:S], NEG_INF)
        la[t] = _logaddexp3(a0, a1, a2) + lp[t, ext]
    lb[T - 1, S - 1] = lp[T - 1, blank]
    if S > 1:
        lb[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    skip_fwd = np.zeros(S, dtype=bool)                   # s may go to s+2
    skip_fwd[:max(S - 2, 0)] = skip[2:]
    for t in range(T - 2, -1, -1):
        b0 = lb[t + 1]
        b1 = np.concatenate((b0[1:], [NEG_INF]))
        b2 = np.where(skip_fwd, np.concatenate((b0, [NEG_INF, NEG_INF]))[2:], NEG_INF)
        lb[t] = _logaddexp3(b0, b1, b2) + lp[t, ext]
    tail = la[T - 1, S - 1] if S == 1 else np.logaddexp(la[T - 1, S - 1], la[T - 1, S - 2])
    nll = -tail
    grad = np.exp(lp)
    if np.isfinite(nll):
        ab = la + lb                                      # (T,S) contains y_t(l'_s)^2
        for t in range(T):
            acc = np.full(V, NEG_INF)
            for s in range(S):
                acc[ext[s]] = np.logaddexp(acc[ext[s]], ab[t, s])
            with np.errstate(invalid="ignore"):
                grad[t] -= np.exp(acc + nll - lp[t])
    else:
        grad[:] = np.nan                                  # caller decides (zero_infinity)
    return nll, grad, la, lb


def ctc_batch(logits, in_len, labels, lab_len, blank=0, zero_infinity=False):
    """logits (B,T,V); labels (B,Lmax) padded; returns nll (B,), dlogits (B,T,V) of sum_b nll_b."""
    B, T, V = logits.shape
    nll = np.zeros(B)
    grad = np.zeros((B, T, V))
    for b in range(B):
        Tb, Lb = int(in_len[b]), int(lab_len[b])
        n, g, _, _ = ctc_one(np.asarray(logits[b, :Tb], dtype=np.float64), np.asarray(labels[b, :Lb]), blank)
        if not np.isfinite(n):
            if zero_infinity:
                n, g = 0.0, np.zeros((Tb, V))
        nll[b] = n
        grad[b, :Tb] = g
    return nll, grad
