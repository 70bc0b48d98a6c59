"""A/B of an environment switch read at engine construction, in ONE process (box-to-box noise is ~2 %):
python tools/ab_env.py NAME valueA valueB [joint]   -> ms/step of each, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
name, va, vb = sys.argv[1:4]
JOINT = len(sys.argv) > 4 and sys.argv[4] == "joint"
M = Models.TransformerOffical if JOINT else Models.TransformerCTC
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
from asr_chinese_e2e_amd import kernels as K
def apply(val):
    os.environ[name] = val
    if name == "ASR_DETERMINISTIC":      # library-level switch (the environment is only read when the library loads)
        K.set_deterministic(val == "1")
def build(val):
    apply(val)
    cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=float(os.environ.get("AB_DROPOUT", "0.0")), ctc_weight=0.3 if JOINT else 1.0, cer_in_iterate=False))
    model = M(cfg, Vocab.synthetic(4232)).cuda()
    opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    for _ in range(10): model.iterate(pack, optimizer=opt)
    torch.cuda.synchronize()
    return model, opt
def run(mo, n=100):
    model, opt = mo
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): model.iterate(pack, optimizer=opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a, b = build(va), build(vb)
ra, rb = [], []
for _ in range(4):
    apply(va); ra.append(run(a))
    apply(vb); rb.append(run(b))
print(f"{name}={va}: " + " ".join(f"{x:.3f}" for x in ra) + f"  min {min(ra):.3f} ms")
print(f"{name}={vb}: " + " ".join(f"{x:.3f}" for x in rb) + f"  min {min(rb):.3f} ms")
