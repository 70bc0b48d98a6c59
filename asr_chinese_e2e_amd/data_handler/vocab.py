"""Character vocabulary with the reference's id contract (Predictor/data_handler/vocab.py:8-84):
ids 0-3 are PAD '$', UNK '%', BOS '^', EOS '&'; convert_id2str drops PAD and joins with spaces."""
from collections import Counter

import torch


class Vocab:
    def __init__(self, PAD="$", UNK="%", BOS="^", EOS="&", tokenize_fn=list):
        self._counter = Counter()
        self.PAD, self.UNK, self.BOS, self.EOS = PAD, UNK, BOS, EOS
        self._token2id = {v: i for i, v in enumerate([PAD, UNK, BOS, EOS]) if v is not None}
        self._id2token = None
        self._tokenize_fn = tokenize_fn

    @classmethod
    def synthetic(cls, size):
        """ids 4.. are distinct CJK characters (no dataset in the container)."""
        v = cls()
        for i in range(size - 4):
            v._token2id[chr(0x4E00 + i)] = len(v._token2id)
        v._id2token = list(v._token2id)
        return v

    def consume_sentance(self, sentance):
        self._counter.update(self._tokenize_fn(sentance))

    def consume_sentance_list(self, sentance_list):
        for s in sentance_list:
            self.consume_sentance(s)

    def build(self, min_count=1, max_vocab=20000):
        for tok, cnt in self._counter.most_common(max_vocab):
            if cnt >= min_count:
                self._token2id[tok] = len(self._token2id)
        self._id2token = list(self._token2id)

    def save(self, path):
        assert self._id2token is not None
        torch.save((self._id2token, self._token2id, self.PAD, self.UNK, self.BOS, self.EOS), path)

    @classmethod
    def load(cls, path):
        obj = cls()
        obj._id2token, obj._token2id, obj.PAD, obj.UNK, obj.BOS, obj.EOS = torch.load(path, weights_only=True)
        return obj

    def convert_str(self, string, use_bos=True, use_eos=True):
        token = self._tokenize_fn(string)
        if use_bos:
            token = [self.BOS] + token
        if use_eos:
            token = token + [self.EOS]
        return self.convert_token(token)

    def convert_token(self, token):
        unk = self._token2id[self.UNK]
        return [self._token2id.get(t, unk) for t in token]

    def convert_id(self, id):
        return [self._id2token[i] for i in id]

    def convert_id2str(self, id):
        pad = self._token2id[self.PAD]
        return " ".join(self._id2token[int(i)] for i in id if int(i) != pad)

    @property
    def vocab_size(self):
        return len(self._token2id)
