"""Front-end throughput: (B, 5 s) waveforms -> log-mel -> normalise (+ SpecAugment) -> frame stacking on one MI355X."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd.data_handler import AudioParser
B, S = 32, 16000 * 5
wav = (torch.randn(B, S, device="cuda") * 0.1)
wl = torch.full((B,), S, dtype=torch.int32, device="cuda")
p = AudioParser(n_mels=80, lfr_m=1, lfr_n=1, device="cuda")
for aug in (False, True):
    for _ in range(3): p.parse_batch(wav, wl, torch.bfloat16, augment=aug, rng=random.Random(0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): feat, fl = p.parse_batch(wav, wl, torch.bfloat16, augment=aug, rng=random.Random(0))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"augment={aug}: {1e3 * dt:.3f} ms per batch of {B} x 5 s ({feat.shape[1]} frames): {B / dt:.0f} utterances/s, {B * 5 / dt:.0f} x real time")
