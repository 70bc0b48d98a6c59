"""The data-parallel step (one rank over RCCL, bf16 wire: every bucket is cast, all-reduced and cast BACK over the fp32 gradients while the
backward pass is still running) relaunched many times from one state: every gradient tensor against the first run's.  A bucket that left
before one of its gradients was complete loses that contribution when it is cast back - by far more than the bf16 rounding of the wire
(one bf16 ulp of an element = 4e-3 of it).  B=32 T=500 V=4232 LAYERS=6 TO=17 python tools/dp_replay_stress.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
import torch, torch.distributed as dist
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.dist import DataParallel
CTC_ONLY = os.environ.get("MODEL", "joint") == "ctc"
LAYERS, STEPS, TO = int(os.environ.get("LAYERS", "6")), int(os.environ.get("REPLAYS", "400")), int(os.environ.get("TO", "17"))
B, T, V = int(os.environ.get("B", "32")), int(os.environ.get("T", "500")), int(os.environ.get("V", "4232"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(5)
M = Models.TransformerCTC if CTC_ONLY else Models.TransformerOffical
cfg = M.get_default_config()()
cfg.fn_build(dict(n_mels=80, lfr_m=1, layer_num=LAYERS, dropout=0.0, ctc_weight=1.0 if CTC_ONLY else 0.3, dtype="bf16"))
m = M(cfg, Vocab.synthetic(V)).cuda()
pack = synthetic_pack(B, T, 80, V, seed=27, ragged=True, Lmin=TO, Lmax=TO, device="cuda", dtype=torch.bfloat16)
dp = DataParallel(m, torch.device("cuda", 0), wire_dtype=torch.bfloat16)
flat = m._flat
def step():
    m.zero_flat_grads()
    dp.bucketer.begin()
    loss, _ = m.train_step(pack, count_hook=dp._counts.start)
    dp.bucketer.finish()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
ref = flat.g.clone()
names = list(flat.index)
scale = torch.stack([flat.view(ref, n).abs().max() for n in names]).clamp_min(1e-30)
worst = torch.zeros(len(names), device="cuda")
for i in range(STEPS):
    step()
    d = (flat.g - ref).abs()
    worst = torch.maximum(worst, torch.stack([flat.view(d, n).max() for n in names]) / scale)
torch.cuda.synchronize()
print(f"{STEPS} data-parallel steps (1 rank, bf16 wire, {len(dp.bucketer.buckets)} buckets), {LAYERS} layers, B={B} T={T}:")
for n, w in sorted(zip(names, worst.tolist()), key=lambda t: -t[1])[:6]:
    print(f"  {n:55s} worst |g - g0| / max|g0| = {w:.3e}")
dist.destroy_process_group()
