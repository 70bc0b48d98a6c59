"""ms per training step of the CTC (default) or joint configuration in THIS process: python tools/step_time.py [joint]
(for settings that are fixed when the HIP runtime starts - environment variables of the runtime - run it once per setting:
tools/rt_env_sweep.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt

JOINT = len(sys.argv) > 1 and sys.argv[1] == "joint"
M = Models.TransformerOffical if JOINT else Models.TransformerCTC
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3 if JOINT else 1.0, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
for _ in range(15):
    model.iterate(pack, optimizer=opt)
torch.cuda.synchronize()
res = []
for _ in range(4):
    t0 = time.perf_counter()
    for _ in range(100):
        model.iterate(pack, optimizer=opt)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) * 10.0)
print("ms/step " + " ".join(f"{r:.3f}" for r in res) + f"  min {min(res):.3f}")
