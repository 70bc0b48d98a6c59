"""One captured training step (no optimizer) of a small joint model replayed many times: every gradient tensor of every replay against the
first replay's.  The order of fp32 atomic adds moves a gradient by ~1e-7 of its largest element; a lost or doubled contribution (two
unordered writers, a block of the graph's pool reused before its last reader ran) by far more.  LAYERS=2 REPLAYS=3000 python tools/replay_stress.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
CTC_ONLY = os.environ.get("MODEL", "joint") == "ctc"
LAYERS, REPLAYS, TO = int(os.environ.get("LAYERS", "2")), int(os.environ.get("REPLAYS", "3000")), int(os.environ.get("TO", "9"))
B, T, V = int(os.environ.get("B", "3")), int(os.environ.get("T", "64")), int(os.environ.get("V", "60"))      # B=32 T=500 V=4232 LAYERS=6 TO=17: the headline shapes
EAGER = os.environ.get("EAGER", "0") == "1"      # the step launched eagerly every time (sequencer, auxiliary stream, armed hand-overs) instead of a replay
torch.manual_seed(5)
M = Models.TransformerCTC if CTC_ONLY else Models.TransformerOffical
cfg = M.get_default_config()()
cfg.fn_build(dict(n_mels=80, lfr_m=1, layer_num=LAYERS, dropout=0.0, ctc_weight=1.0 if CTC_ONLY else 0.3, dtype="bf16"))
m = M(cfg, Vocab.synthetic(V)).cuda()
pack = synthetic_pack(B, T, 80, V, seed=27, ragged=True, Lmin=TO, Lmax=TO, device="cuda", dtype=torch.bfloat16)
eng = m._ensure_engine(torch.device("cuda", 0))
def body():
    m.zero_flat_grads()
    return m.train_step(pack)[0]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
flat = m._flat
if EAGER:
    class g:      # same interface as the graph
        @staticmethod
        def replay():
            loss.copy_(body())
    loss = body().clone()
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = body()
g.replay()
ref, ref_loss = flat.g.clone(), loss.clone()
names = list(flat.index)
scale = torch.stack([flat.view(ref, n).abs().max() for n in names]).clamp_min(1e-30)
worst = torch.zeros(len(names), device="cuda")
worst_loss = torch.zeros((), device="cuda")
for i in range(REPLAYS):
    g.replay()
    d = (flat.g - ref).abs()
    worst = torch.maximum(worst, torch.stack([flat.view(d, n).max() for n in names]) / scale)
    worst_loss = torch.maximum(worst_loss, (loss - ref_loss).abs().max())
torch.cuda.synchronize()
print(f"{REPLAYS} {'eager steps' if EAGER else 'replays'}, {LAYERS} layers, B={B} T={T} To<={TO}: worst loss difference {float(worst_loss):.3e}")
for n, w in sorted(zip(names, worst.tolist()), key=lambda t: -t[1])[:8]:
    print(f"  {n:55s} worst |g - g0| / max|g0| = {w:.3e}")
