"""Log-mel front end on the GPU (Predictor/data_handler/processor.py:18-100 of the reference ran
torchaudio on the CPU): waveform batch -> log-mel -> scalar mean/std normalisation -> LFR."""
import math

import numpy as np
import random

import torch

from .. import kernels as K

SR, N_FFT, HOP = 16000, 400, 160


def mel_filterbank(n_mels, f_min=40.0, f_max=SR / 2 - 200.0):
    """HTK-mel triangular filters over FFT-bin frequencies; f_max = sr/2 - 200 Hz is the build's
    reading of the reference's `f_max=-200` (processor.py:24) - see DESIGN.md."""
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    freqs = np.arange(N_FFT // 2 + 1, dtype=np.float64) * SR / N_FFT
    m_pts = np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
    return torch.from_numpy(fb.astype(np.float32))


def build_LFR_features(inputs, m, n):
    """Host-side LFR with the reference's rule (processor.py:74-100), index form."""
    inputs = np.asarray(inputs)
    T = inputs.shape[0]
    T_out = int(np.ceil(T / n))
    idx = np.minimum(np.arange(T_out)[:, None] * n + np.arange(m)[None, :], T - 1)
    return inputs[idx].reshape(T_out, m * inputs.shape[1])


def sample_spec_augment(n_mels, n_frames, rng=random, F=30, T=40):
    """The mask ranges AudioParser.augment would draw (processor.py:52-58 -> augments.time_mask then
    augments.freq_mask, one mask each, F=30, T=40), with the SAME sequence of randrange calls, so a
    seeded `random` gives the reference's masks.  Returns [t0, t1, f0, f1] (empty ranges = no mask).
    Where the reference would raise (randrange on an empty range: fewer frames than the drawn width)
    no mask is applied."""
    t0 = t1 = f0 = f1 = 0
    t = rng.randrange(0, T)                      # augments.py:29
    if n_frames - t > 0:
        tz = rng.randrange(0, n_frames - t)      # :30
        if t > 0:                                # :33 early return when the width is 0
            t0, t1 = tz, rng.randrange(tz, tz + t)   # :36 (mask_end may equal t_zero: empty)
    f = rng.randrange(0, F)                      # :9
    if n_mels - f > 0:
        fz = rng.randrange(0, n_mels - f)        # :10
        if f > 0:
            f0, f1 = fz, rng.randrange(fz, fz + f)   # :15
    return [t0, t1, f0, f1]


class AudioParser:
    """Batched device front end: parse_batch(wav (B,S) f32 cuda, wav_len (B) int) ->
    (features (B, T_lfr, lfr_m*n_mels), feature_len (B) int32)."""

    def __init__(self, sample_rate=SR, n_mels=80, window_size=N_FFT, hop=HOP, lfr_m=4, lfr_n=3, device="cuda"):
        assert sample_rate == SR and window_size == N_FFT and hop == HOP, "kernel is specialised to 16 kHz / 400 / 160"
        self.n_mels, self.lfr_m, self.lfr_n = n_mels, lfr_m, lfr_n
        self.window = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32).to(device)
        self.melfb = mel_filterbank(n_mels).to(device)

    def parse_batch(self, wav, wav_len, dtype=torch.float32, augment=False, rng=random):
        """augment=True: SpecAugment as AudioParser.parse(path, augment=True) of the reference, masks
        drawn on the host from `rng` (the `random` module by default, as the reference), applied on the
        device between normalisation and frame stacking."""
        B, S = wav.shape
        Tmax = 1 + S // HOP
        wl = wav_len.to(torch.int32)
        feat = K.logmel(wav.contiguous(), wl, self.window, self.melfb, Tmax)
        Tl = (Tmax + self.lfr_n - 1) // self.lfr_n
        masks = None
        if augment:
            frames = [min(1 + int(l) // HOP, Tmax) if int(l) > 0 else 0 for l in wav_len.tolist()]
            masks = torch.tensor([sample_spec_augment(self.n_mels, fr, rng) for fr in frames], dtype=torch.int32).to(wav.device)
        return K.utt_norm_lfr(feat, wl, self.lfr_m, self.lfr_n, Tl, dtype, masks=masks)
