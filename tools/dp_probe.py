"""Which part of the DataParallel wrapper costs what at ONE rank (the step takes 7.0 ms through it, 3.1 ms plain - round 4).
MODE=full   : RCCL process group, real collectives
MODE=stub   : process group initialised, every collective replaced by a no-op before the wrapper is built
MODE=nopg   : no process group at all (torch.distributed functions stubbed): the wrapper's own stream / event logic only
python tools/dp_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as tdist
from asr_chinese_e2e_amd import Models, dist as D
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt

mode = os.environ.get("MODE", "full")
from asr_chinese_e2e_amd import engine as E
if os.environ.get("PROBE", "1") == "0":
    E.QUEUE_PROBE = False
if os.environ.get("SIDE", "low") == "normal":      # experiment: the weight-gradient stream at normal priority (a torch pool stream)
    E._side_stream = lambda device: torch.cuda.Stream(device=device)
if os.environ.get("NOAUX", "0") == "1":            # experiment: no separate auxiliary stream (the weight-gradient stream serves as "aux" too)
    _orig_shared = E._shared_stream
    E._shared_stream = lambda device, kind: _orig_shared(device, "wgrad")
if os.environ.get("BURN", "0") != "0":             # experiment: take N streams from torch's pool first (shifts the pool index of every later stream)
    _burn = [torch.cuda.Stream() for _ in range(int(os.environ["BURN"]))]
late = os.environ.get("ORDER", "early") == "late"      # late: the process group is created AFTER the model, its streams and a few steps
if mode != "nopg" and not late:
    D.init("nccl")
if mode in ("stub", "nopg"):
    class _W:
        def wait(self): pass
    tdist.all_reduce = lambda *a, **k: _W()
    tdist.broadcast = lambda *a, **k: _W()
if mode == "nopg":
    tdist.get_rank = lambda group=None: 0
    tdist.get_world_size = lambda group=None: 1
JOINT = os.environ.get("CONFIG", "ctc") == "joint"
M = Models.TransformerOffical if JOINT else Models.TransformerCTC
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3 if JOINT else 1.0, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)


def timeit(step, n=40, warm=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * th / n, 1e3 * (time.perf_counter() - t0) / n


if os.environ.get("TOUCH", "0") != "0":            # experiment: launch one tiny kernel on N fresh pool streams BEFORE the engine / wrapper take theirs
    _t = [torch.cuda.Stream() for _ in range(int(os.environ["TOUCH"]))]
    _tb = torch.zeros(64, device="cuda")
    for _s in _t:
        with torch.cuda.stream(_s):
            _tb.add_(1.0)
    torch.cuda.synchronize()
print("mode", mode)
if os.environ.get("MAINSTREAM", "null") == "pool":      # experiment: the whole step on a non-blocking pool stream instead of the legacy null stream
    _main = torch.cuda.Stream()
    torch.cuda.set_stream(_main)
    print("main stream: torch pool stream (non-blocking)")
print("plain           host %.2f wall %.3f ms/step" % timeit(lambda: model.iterate(pack, optimizer=opt)), flush=True)
if mode != "nopg" and late:
    D.init("nccl")
    print("plain, pg up    host %.2f wall %.3f ms/step" % timeit(lambda: model.iterate(pack, optimizer=opt)), flush=True)
dp = D.DataParallel(model, "cuda", wire_dtype=os.environ.get("WIRE", "auto") if os.environ.get("WIRE", "auto") == "auto" else (torch.float32 if os.environ["WIRE"] == "fp32" else torch.bfloat16))
print("DataParallel    host %.2f wall %.3f ms/step" % timeit(lambda: dp.iterate(pack, optimizer=opt)), flush=True)
# without the asynchronous counts (loss normalisers): the hook is the only collective in front of the forward pass
orig = model.train_step
model.train_step = lambda input, loss_scale=1.0, count_hook=None: orig(input, loss_scale, None)
print("  - no counts   host %.2f wall %.3f ms/step" % timeit(lambda: dp.iterate(pack, optimizer=opt)), flush=True)
model.train_step = orig
# without the bucket launches (marks ignored until finish)
eng = model._engine
eng.grad_ready = None
print("  - no marks    host %.2f wall %.3f ms/step" % timeit(lambda: dp.iterate(pack, optimizer=opt)), flush=True)
if mode != "nopg":
    tdist.destroy_process_group()
