"""Character error rate with the reference's string convention (Predictor/Utils/score.py:4-13):
edit distance over the SPACE-JOINED token strings (spaces count as characters), divided by the
number of space-separated tokens of the reference string.  python-Levenshtein is replaced by a
two-row dynamic programme."""


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def calculate_cer(s1, s2):
    """s1: hypothesis, s2: gold, both space-separated token strings."""
    return edit_distance(s1, s2) / len(s2.split(" "))
