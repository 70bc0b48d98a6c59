"""The loader's staging slots under a SLOW consumer: every batch of an epoch is checksummed twice - once with the consumer waiting for
each batch before it asks for the next (nothing overlaps), once with the consumer's stream ~2 ms behind (a spin kernel in front of every
read), so that the helper thread runs PREFETCH batches ahead and reuses slots while earlier batches are still being read on the device.
The two lists must be bit-identical.  python tools/loader_stress.py [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_chinese_e2e_amd.data_handler import AudioParser, BucketedWaveLoader, Vocab, WaveDataset
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.RandomState(0)
vocab = Vocab.synthetic(30)
items = [((rng.randn(int(rng.randint(4000, 48000))) * 0.1).astype(np.float32), [int(t) for t in rng.randint(4, 30, size=rng.randint(2, 9))]) for _ in range(203)]
ds = WaveDataset(items, vocab)
parser = AudioParser(n_mels=80, lfr_m=1, lfr_n=1, device="cuda")
def checksum(p):
    w = p.wave.float()
    return torch.stack([w.sum(), w.abs().sum(), (w * w).sum(), p.wave_len.float().sum(), p.tgt_for_input.float().sum(), p.tgt_len.float().sum() if p.get("tgt_len") is not None else w.new_zeros(())])
def epoch(seed, slow):
    loader = BucketedWaveLoader(ds, 4, parser=parser, augment=False, shuffle=True, seed=seed, bucket_size=16, dtype=torch.bfloat16)
    out = []
    for p in loader:
        if slow:
            torch.cuda._sleep(4_000_000)      # ~2 ms: the reads of this batch run long after the helper has moved on
            out.append(checksum(p))
        else:
            torch.cuda.synchronize()
            out.append(checksum(p).clone())
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return torch.stack(out).cpu()
bad = 0
for e in range(epochs):
    a, b = epoch(e, False), epoch(e, True)
    same = torch.equal(a, b)
    bad += 0 if same else 1
    print(f"epoch {e}: {a.shape[0]} batches, slow consumer {'identical' if same else 'DIFFERENT at batches ' + str((a != b).any(1).nonzero().flatten().tolist()[:10])}", flush=True)
print("epochs with differences:", bad)
