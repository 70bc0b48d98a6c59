"""oracle/ctc_ref.py (numpy restatement of the CTC alpha/beta recursion) pinned by hand-computed
known answers and by torch's F.ctc_loss (fp64).  The reference has no CTC: 'parity unpinned' by it."""
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
