"""Host enqueue time and GPU time of the phases of a joint training step (encoder fwd, CTC branch + decoder fwd + CE, decoder bwd,
encoder bwd, optimizer): wraps the engine's phase methods with perf_counter + events.  python tools/phase_times.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
M = Models.TransformerOffical
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
for _ in range(5): model.iterate(pack, optimizer=opt)
eng = model._engine if hasattr(model, "_engine") else model._ensure_engine("cuda")
host, gpu = collections.defaultdict(float), collections.defaultdict(list)
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t0 = time.perf_counter()
        r = fn(*a, **k)
        host[label] += time.perf_counter() - t0
        e1.record(); gpu[label].append((e0, e1))
        return r
    setattr(obj, name, w)
for name in ("encoder_fwd", "ctc_branch_async", "decoder_fwd", "decoder_bwd", "encoder_bwd"):
    wrap(eng, name, name)
wrap(opt, "fused_step", "optimizer")
N = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): model.iterate(pack, optimizer=opt)
th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"step: host {1e3 * th / N:.2f} ms, wall {1e3 * tw / N:.2f} ms (events add ~0.1 ms)")
for k in host:
    g = sum(a.elapsed_time(b) for a, b in gpu[k]) / N
    print(f"  {k:18s} host {1e3 * host[k] / N:6.2f} ms   GPU (main stream, start to end) {g:6.2f} ms")
