"""oracle/ctc_ref.py (numpy restatement of the CTC alpha/beta recursion) pinned by hand-computed
known answers and by torch's F.ctc_loss (fp64).  The reference has no CTC: 'parity unpinned' by it."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctc_ref as C


def torch_ctc(logits, in_len, labels, lab_len, zero_infinity=False):
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = F.log_softmax(x, -1).transpose(0, 1)
    nll = F.ctc_loss(lp, torch.tensor(labels), torch.tensor(in_len), torch.tensor(lab_len), blank=0,
                     reduction="none", zero_infinity=zero_infinity)
    nll.sum().backward()
    return nll.detach().numpy(), x.grad.numpy()


def test_known_answer_T3_L1():
    # uniform over V=2 -> y=0.5 everywhere; paths for 'a' over T=3: a--, -a-, --a, aa-, -aa, aaa = 6
    # wait: collapse(aa-)=a, (a-a) = 'aa' no. valid: a__,_a_,__a,aa_,_aa,aaa => 6 of 8
    logits = np.zeros((1, 3, 2))
    nll, g = C.ctc_batch(logits, [3], np.array([[1]]), [1])
    assert abs(nll[0] - (-np.log(6 / 8))) < 1e-12
    assert abs(g.sum()) < 1e-12          # softmax-grad rows sum to 0


def test_known_answer_repeated_label():
    # label 'aa' needs a blank between: T=3 only path a_a -> p = (1/3)^3 with V=3 uniform
    logits = np.zeros((1, 3, 3))
    nll, _ = C.ctc_batch(logits, [3], np.array([[1, 1]]), [2])
    assert abs(nll[0] - 3 * np.log(3)) < 1e-12
    # T=2 is infeasible for 'aa'
    nll, g = C.ctc_batch(np.zeros((1, 2, 3)), [2], np.array([[1, 1]]), [2])
    assert np.isinf(nll[0])
    nll, g = C.ctc_batch(np.zeros((1, 2, 3)), [2], np.array([[1, 1]]), [2], zero_infinity=True)
    assert nll[0] == 0 and not g.any()


def test_known_answer_empty_label():
    # L=0: only the all-blank path
    rng = np.random.RandomState(0)
    logits = rng.randn(1, 4, 5)
    nll, g = C.ctc_batch(logits, [4], np.zeros((1, 1), dtype=np.int64), [0])
    lp = C.log_softmax(logits[0])
    assert abs(nll[0] + lp[:, 0].sum()) < 1e-12


@pytest.mark.parametrize("seed", range(6))
def test_matches_torch_fp64(seed):
    rng = np.random.RandomState(seed)
    B, T, V, Lmax = 5, int(rng.randint(8, 30)), int(rng.randint(4, 40)), 6
    logits = rng.randn(B, T, V) * 2
    in_len = rng.randint(T // 2, T + 1, size=B); in_len[0] = T
    lab_len = rng.randint(0, Lmax + 1, size=B); lab_len[1] = Lmax
    labels = rng.randint(1, V, size=(B, Lmax))
    if seed % 2:
        labels[:, 1] = labels[:, 0]       # force repeats
    labels = np.where(np.arange(Lmax)[None] < lab_len[:, None], labels, 0)
    nll, g = C.ctc_batch(logits, in_len, labels, lab_len, zero_infinity=True)
    tn, tg = torch_ctc(logits, in_len, labels, lab_len, zero_infinity=True)
    assert np.allclose(nll, tn, rtol=1e-10, atol=1e-10)
    assert np.allclose(g, tg, rtol=1e-8, atol=1e-10)


def test_greedy_decode_known_answers():
    """Best-path decoding on hand-made paths: repeats merge, blanks split repeats, padded frames ignored."""
    import numpy as np
    from oracle import ref_model as R
    V = 6
    paths = [[0, 3, 3, 0, 3, 4, 4, 0, 0, 5], [2, 2, 2, 2, 2, 2, 2, 2, 2, 2], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 0, 1, 1, 0, 0, 2, 3, 3, 1]]
    lens = [10, 10, 10, 7]
    x = np.full((4, 10, V), -1.0)
    for b, p in enumerate(paths):
        for t, c in enumerate(p):
            x[b, t, c] = 2.0
    assert R.ctc_greedy_decode(x, lens) == [[3, 3, 4, 5], [2], [], [1, 1, 2]]
    x[0, 1, 4] = 2.0      # tie between ids 3 and 4 at frame 1: the first index (3) wins
    assert R.ctc_greedy_decode(x, lens)[0] == [3, 3, 4, 5]
    toks = {i: str(i) for i in range(V)}
    assert R.ctc_cer_percent([[3, 3, 4, 5], [2]], [[3, 4, 5, 0], [2, 0, 0, 0]], [3, 1], toks) == pytest.approx(100.0 * (2 / 3 + 0) / 2)


def test_prefix_beam_search_oracle_against_brute_force():
    """oracle/decode_ref.ctc_prefix_beam_search (the checker of the GPU path; nothing in the reference to pin it to) against
    the enumeration of every alignment on tiny lattices: with a beam wide enough to hold every prefix the scores are the
    exact label probabilities and the ranking is the brute-force ranking."""
    from oracle import decode_ref as D
    rng = np.random.RandomState(3)
    for T, V in ((3, 3), (4, 3), (5, 3), (4, 4)):
        x = rng.randn(T, V) * 1.5
        logp = x - np.log(np.exp(x).sum(-1, keepdims=True))
        brute = D.best_labelling_bruteforce(logp)
        got = D.ctc_prefix_beam_search(logp, beam_size=10_000)
        bd = dict(brute)
        assert got[0][0] == brute[0][0]
        for prefix, sc in got[:8]:
            assert abs(sc - bd[prefix]) < 1e-9, (prefix, sc, bd[prefix])
        assert abs(D.ctc_label_logprob_bruteforce(logp, brute[0][0]) - brute[0][1]) < 1e-9
        # and the label probability equals -nll of the CTC oracle
        lab = np.array([list(brute[1][0]) + [0] * (T - len(brute[1][0]))])
        nll, _ = C.ctc_batch(logp[None], [T], lab, [len(brute[1][0])])
        assert abs(-nll[0] - brute[1][1]) < 1e-9


def test_prefix_beam_search_known_cases():
    from oracle import decode_ref as D
    # one dominant class per frame: the best prefix is the collapsed greedy path
    logp = np.log(np.array([[0.05, 0.9, 0.05], [0.05, 0.9, 0.05], [0.9, 0.05, 0.05], [0.05, 0.05, 0.9]]))
    assert D.ctc_prefix_beam_search(logp, 4)[0][0] == (1, 2)
    # the classic case where the best PATH (all blank) is not the best LABELLING: p(blank) = 0.6 per frame
    logp = np.log(np.array([[0.6, 0.4], [0.6, 0.4]]))
    best = D.ctc_prefix_beam_search(logp, 4)
    assert best[0][0] == (1,) and abs(math.exp(best[0][1]) - 0.64) < 1e-12 and abs(math.exp(dict(best)[()]) - 0.36) < 1e-12
    # rescoring: ranks by the weighted sum, keeps both parts
    nb = [dict(yseq=[2, 5, 3], score=-1.0), dict(yseq=[2, 6, 3], score=-1.2)]
    out = D.joint_rescore(nb, lambda toks: {5: -9.0, 6: -1.0}[toks[0]], 0.5)
    assert [h["yseq"][1] for h in out] == [6, 5] and abs(out[0]["score"] - (-1.1)) < 1e-12 and out[1]["ctc_score"] == -9.0
