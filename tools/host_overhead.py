"""Diagnostic: host enqueue time vs GPU time per training step, plus a cProfile of the host side."""
import cProfile, pstats, sys, os, time
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
for _ in range(5):
    model.iterate(pack, optimizer=opt)
torch.cuda.synchronize()
N = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); t0 = time.perf_counter()
for _ in range(N):
    model.iterate(pack, optimizer=opt)
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"host enqueue {1e3*t_host/N:.2f} ms/step, gpu span {e0.elapsed_time(e1)/N:.2f} ms/step, wall {1e3*t_wall/N:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    model.iterate(pack, optimizer=opt)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
print("=== by cumulative time")
st.sort_stats("cumtime").print_stats(45)
