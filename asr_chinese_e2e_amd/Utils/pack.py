"""Batch container of the reference (Predictor/Utils/pack.py:3-27): a dict with attribute access
that returns None for missing keys, and a .cuda() that moves every tensor."""


class Pack(dict):
    def __getattr__(self, name):
        return self.get(name)

    def add(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = v

    def cuda(self, non_blocking=False):
        out = Pack()
        for k, v in self.items():
            if isinstance(v, tuple):
                out[k] = tuple(x.cuda(non_blocking=non_blocking) for x in v)
            else:
                out[k] = v.cuda(non_blocking=non_blocking)
        return out
