"""Zero-padding of ragged batches (Predictor/data_handler/padder.py:4-27)."""
import torch


class Padder:
    @staticmethod
    def pad_two(inputs, pad_value, lengths=None):
        lengths = [len(i) for i in inputs] if lengths is None else lengths
        out = torch.full((len(inputs), max(lengths)), float(pad_value))
        for row, (seq, l) in enumerate(zip(inputs, lengths)):
            out[row, :l] = torch.as_tensor(seq, dtype=out.dtype)
        return out, lengths

    @staticmethod
    def pad_tri(inputs, pad_value, lengths=None):
        lengths = [len(i) for i in inputs] if lengths is None else lengths
        out = torch.full((len(inputs), max(lengths), inputs[0].size(-1)), float(pad_value))
        for row, (seq, l) in enumerate(zip(inputs, lengths)):
            out[row, :l, :] = seq
        return out, lengths
