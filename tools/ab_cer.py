"""A/B of the per-step CER placement (auxiliary stream beside the backward pass vs. main stream behind the optimizer), joint model WITH
cer_in_iterate, in one process: python tools/ab_cer.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
M = Models.TransformerOffical
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
def build():
    cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3, cer_in_iterate=True))
    model = M(cfg, Vocab.synthetic(4232)).cuda()
    opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    return model, opt
def run(mo, flag, n=100):
    model, opt = mo
    M.CER_BESIDE_BACKWARD = flag
    for _ in range(5): model.iterate(pack, optimizer=opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): model.iterate(pack, optimizer=opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a, b = build(), build()
for m in (a, b):
    for _ in range(10): m[0].iterate(pack, optimizer=m[1])
ra, rb = [], []
for _ in range(4):
    ra.append(run(a, True)); rb.append(run(b, False))
print("CER beside the backward pass: " + " ".join(f"{x:.3f}" for x in ra) + f"  min {min(ra):.3f} ms")
print("CER behind the optimizer:     " + " ".join(f"{x:.3f}" for x in rb) + f"  min {min(rb):.3f} ms")
