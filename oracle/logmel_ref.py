"""CPU oracle for the log-mel front end + utterance normalisation + LFR (numpy float64).

TEST INFRASTRUCTURE ONLY - never imported by the product package.

Follows the STRUCTURE of the reference's AudioParser (Predictor/data_handler/processor.py):
  transform  :33-40   MelSpectrogram(sr=16000, ws=400, hop=160, f_min=40, f_max=-200, pad=0,
                      n_mels) -> log(x + 1e-20)
  normalize  :42-46   (f - f.mean()) / f.std()      scalar mean, UNBIASED std over all (n_mels,T)
  LFR        :74-100  build_LFR_features(inputs, m, n): stack m frames, stride n, tail frames
                      padded by repeating the last input frame

The mel-spectrogram arithmetic itself lives in torchaudio (un-vendored, version un-pinned; the
kwarg names sr=/ws=/hop= imply torchaudio 0.2-0.3, 2019), which is NOT installed here, and the
reference holds no fixture for it => "parity unpinned" for the spectrogram.  The restatement is of
the published definition that API documents: centred STFT (reflect padding), periodic Hann
window of 400 samples, n_fft = 400, hop 160, power spectrum, HTK-mel triangular filterbank
(mel = 2595 log10(1 + f/700)) between f_min and f_max, no filter normalisation.
Build choice (documented in DESIGN.md): f_max = sr/2 - 200 = 7800 Hz (the reference passes
f_max = -200, degenerate if taken literally), FFT-bin frequencies k*sr/n_fft.
LFR IS pinned: tests compare build_lfr() with golden vectors made by the reference's own
build_LFR_features (tests/golden/ops.npz, lfr/*).
"""
import numpy as np

SR, N_FFT, WIN, HOP = 16000, 400, 400, 160
F_MIN, F_MAX = 40.0, SR / 2 - 200.0
LOG_FLOOR = 1e-20


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_mels, n_fft=N_FFT, sr=SR, f_min=F_MIN, f_max=F_MAX):
    """(n_fft//2+1, n_mels) triangular filters, unnormalised, HTK mel scale."""
    freqs = np.arange(n_fft // 2 + 1, dtype=np.float64) * sr / n_fft
    f_pts = mel_to_hz(np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def hann_periodic(n=WIN):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def num_frames(n_samples, hop=HOP):
    return 1 + n_samples // hop


def log_mel(wav, n_mels=80):
    """wav (n_samples,) float -> (T, n_mels) log-mel, T = 1 + n_samples // hop."""
    wav = np.asarray(wav, dtype=np.float64)
    pad = N_FFT // 2
    x = np.pad(wav, (pad, pad), mode="reflect")
    T = num_frames(len(wav))
    idx = np.arange(T)[:, None] * HOP + np.arange(WIN)[None, :]
    frames = x[idx] * hann_periodic()[None, :]
    spec = np.abs(np.fft.rfft(frames, n=N_FFT, axis=1)) ** 2
    return np.log(spec @ mel_filterbank(n_mels) + LOG_FLOOR)


def utt_normalize(feat):
    """processor.py:44: scalar mean and unbiased (N-1) std over the whole matrix."""
    feat = np.asarray(feat, dtype=np.float64)
    return (feat - feat.mean()) / feat.std(ddof=1)


def build_lfr(x, m, n):
    """processor.py:74-100 restated with index arithmetic: output frame i stacks input frames
    i*n .. i*n+m-1, indices past the end clamp to the last frame; T_out = ceil(T/n)."""
    x = np.asarray(x)
    T = x.shape[0]
    T_out = int(np.ceil(T / n))
    idx = np.minimum(np.arange(T_out)[:, None] * n + np.arange(m)[None, :], T - 1)
    return x[idx].reshape(T_out, m * x.shape[1])


def front_end(wav, n_mels=80, lfr_m=4, lfr_n=3):
    """parse() without augmentation: log-mel -> normalise -> LFR (processor.py:61-71)."""
    return build_lfr(utt_normalize(log_mel(wav, n_mels)), lfr_m, lfr_n)


def sample_spec_augment(n_mels, n_frames, rng, F=30, T=40):
    """The randrange calls of augments.time_mask (augments.py:24-42) then augments.freq_mask (:4-21),
    one mask each as AudioParser.augment uses them (processor.py:52-58).  Returns [t0, t1, f0, f1];
    empty ranges where the reference returns early (width 0).  A range the reference cannot draw
    (fewer frames / channels than the width: its randrange raises) gives no mask."""
    t0 = t1 = f0 = f1 = 0
    t = rng.randrange(0, T)
    if n_frames - t > 0:
        tz = rng.randrange(0, n_frames - t)
        if t > 0:
            t0, t1 = tz, rng.randrange(tz, tz + t)
    f = rng.randrange(0, F)
    if n_mels - f > 0:
        fz = rng.randrange(0, n_mels - f)
        if f > 0:
            f0, f1 = fz, rng.randrange(fz, fz + f)
    return [t0, t1, f0, f1]


def spec_augment(feature, masks):
    """feature (n_mels, T) normalised; time mask filled with the mean, then mel mask filled with the
    mean of the time-masked feature (cloned.mean() at each call, replace_with_zero=False).
    Pinned by tests/golden/augment.npz (the reference's own functions)."""
    t0, t1, f0, f1 = masks
    y = np.array(feature, dtype=np.float64, copy=True)
    if t1 > t0:
        y[:, t0:t1] = y.mean()
    if f1 > f0:
        y[f0:f1, :] = y.mean()
    return y
