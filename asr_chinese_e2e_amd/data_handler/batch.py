"""Batch layout contract of the reference (data/data_loader/ai_shell_1.py:67-88): a Pack of
zero-padded dense tensors  wave (B,Tmax,F) f32, tgt_for_input / tgt_for_metric (B,Lmax) i64,
wave_len / tgt_len (B,) i64 - plus a synthetic generator of AISHELL-1-shaped batches
(wave ~ N(0,1), labels U{4..V-1}, lengths U{8..22})."""
import torch

from ..Bases import BaseConfig
from ..Utils import Pack
from .padder import Padder


class DataConfigAiShell1(BaseConfig):   # Predictor/data_handler/data_config.py:6-19
    sample_rate = 16000
    n_mels = 80
    window_size = 400
    augment = False
    lfr_m = 4
    lfr_n = 3


class collat:
    def __init__(self, use_cuda=True):
        self.use_cuda = use_cuda

    def __call__(self, batch):
        wave, wave_len = Padder.pad_tri([b[0] for b in batch], 0)
        tgt_in, tgt_len = Padder.pad_two([b[1] for b in batch], 0)
        tgt_metric, _ = Padder.pad_two([b[2] for b in batch], 0)
        pack = Pack()
        pack.add(wave=wave, tgt_for_input=tgt_in.long(), wave_len=torch.tensor(wave_len).long(), tgt_len=torch.tensor(tgt_len).long(),
                 tgt_for_metric=tgt_metric.long())
        return pack.cuda() if self.use_cuda else pack


def synthetic_pack(B, T, F, V, seed=1234, ragged=False, Lmin=8, Lmax=22, device="cpu", dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    wave = torch.randn(B, T, F, generator=g)
    if ragged:
        wave_len = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
        wave_len[0] = T
    else:
        wave_len = torch.full((B,), T, dtype=torch.long)
    tgt_len = torch.randint(Lmin, Lmax + 1, (B,), generator=g)
    L = int(tgt_len.max())
    tgt = torch.randint(4, V, (B, L), generator=g)
    ar = torch.arange(L).unsqueeze(0)
    tgt = torch.where(ar < tgt_len.unsqueeze(1), tgt, torch.zeros_like(tgt))
    wave = torch.where(torch.arange(T).view(1, T, 1) < wave_len.view(B, 1, 1), wave, torch.zeros_like(wave))
    pack = Pack()
    pack.add(wave=wave.to(dtype), tgt_for_input=tgt, tgt_for_metric=tgt.clone(), wave_len=wave_len.long(), tgt_len=tgt_len.long())
    if torch.device(device).type == "cuda":
        pack = Pack({k: v.to(device) for k, v in pack.items()})
    return pack
