"""Diagnostic: host profile of the DataParallel wrapper with one rank (RCCL)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
import torch
from asr_chinese_e2e_amd import Models, dist as D
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
D.init("nccl")
M = Models.TransformerCTC
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=1.0, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
dp = D.DataParallel(model, "cuda", bucket_bytes=int(os.environ.get("BUCKET_MB", "16")) << 20)
if os.environ.get("NO_AR") == "1":
    import torch.distributed as tdist
    class _W:
        def wait(self): pass
    tdist.all_reduce = lambda *a, **k: _W()
for _ in range(5):
    dp.iterate(pack, optimizer=opt)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    dp.iterate(pack, optimizer=opt)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"DP host enqueue {1e3*t_host/N:.2f} ms/step, wall {1e3*t_wall/N:.2f} ms/step, buckets {len(dp.bucketer.buckets)}")
if os.environ.get("NOPROF") == "1":
    torch.distributed.destroy_process_group(); sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    dp.iterate(pack, optimizer=opt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
torch.distributed.destroy_process_group()
