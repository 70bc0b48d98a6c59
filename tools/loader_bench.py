"""Training from WAVEFORMS: in-memory 5-s utterances -> BucketedWaveLoader (pinned copy + log-mel / normalisation / SpecAugment on a side stream) ->
model.iterate, against the same model fed one resident batch (what bench.py times).  python tools/loader_bench.py [joint]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import AudioParser, BucketedWaveLoader, Vocab, WaveDataset
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
JOINT = len(sys.argv) > 1 and sys.argv[1] == "joint"
B, S, NB = 32, 16000 * 5, 40
rng = np.random.RandomState(0)
vocab = Vocab.synthetic(4232)
items = [((rng.randn(S) * 0.1).astype(np.float32), [int(t) for t in rng.randint(4, 4232, size=16)]) for _ in range(B * NB)]
if os.environ.get("FILES") == "1":      # the same utterances as 16-bit WAV files (decoded by the loader)
    import tempfile, wave
    d = tempfile.mkdtemp(prefix="asr_wav_")
    for i, (w, t) in enumerate(items):
        path = os.path.join(d, f"u{i}.wav")
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((np.clip(w, -1, 1) * 32767).astype("<i2").tobytes())
        items[i] = (path, t)
    print(f"{len(items)} WAV files under {d}")
ds = WaveDataset(items, vocab)
parser = AudioParser(n_mels=80, lfr_m=1, lfr_n=1, device="cuda")
M = Models.TransformerOffical if JOINT else Models.TransformerCTC
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3 if JOINT else 1.0))
model = M(cfg, vocab).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
loader = BucketedWaveLoader(ds, B, parser=parser, augment=True, shuffle=True, seed=1, dtype=torch.bfloat16)
def epoch():
    n = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for pack in loader:
        model.iterate(pack, optimizer=opt)
        n += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, pack
epoch()
for _ in range(3):
    ms, pack = epoch()
    print(f"from waveforms ({NB} batches of {B} x 5 s, SpecAugment on): {ms:.3f} ms/step", flush=True)
# host cost of preparing a batch alone
idx = list(range(B))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _k in range(20): loader._prepare(idx, _k)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"_prepare alone: {(t1 - t0) / 20 * 1e3:.3f} ms of host time per batch")
for _ in range(10): model.iterate(pack, optimizer=opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): model.iterate(pack, optimizer=opt)
torch.cuda.synchronize()
print(f"one resident batch: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/step")
