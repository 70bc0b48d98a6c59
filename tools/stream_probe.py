"""How the step time reacts to one more active stream (the data-parallel wrapper's communication stream): ms/step of the CTC config
(a) plain, (b) with a third / fourth stream that receives an event wait + one small kernel per 'bucket' and is joined at the end of the step,
(c) the DataParallel wrapper itself with its collective stubbed out.  python tools/stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt

M = Models.TransformerCTC
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=1.0, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda()
opt = NoamOpt(512, 1, 4000, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)


def timeit(step, n=60, warm=15):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"plain: {timeit(lambda: model.iterate(pack, optimizer=opt)):.3f} ms/step", flush=True)
eng = model._engine
buf = torch.zeros(1 << 20, device="cuda")
for label, mk in (("extra torch.cuda.Stream()", lambda: torch.cuda.Stream()), ("extra high-priority stream", lambda: torch.cuda.Stream(priority=-1))):
    extra = mk()
    marks = []

    def ready(off, streams, extra=extra):
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            extra.wait_event(ev)
        with torch.cuda.stream(extra):
            buf.add_(1.0)

    eng.grad_ready = ready

    def step(extra=extra):
        model.zero_flat_grads()
        model.train_step(pack)
        torch.cuda.current_stream().wait_stream(extra)
        opt.fused_step(model._flat, 5.0)

    print(f"{label}, one small kernel per mark: {timeit(step):.3f} ms/step", flush=True)
    eng.grad_ready = None
print(f"plain again: {timeit(lambda: model.iterate(pack, optimizer=opt)):.3f} ms/step", flush=True)
if os.environ.get("INIT_PG", "0") != "0":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
    dist.init_process_group("nccl", rank=0, world_size=1)
    print(f"process group initialised (no collective yet): {timeit(lambda: model.iterate(pack, optimizer=opt)):.3f} ms/step", flush=True)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"after one all_reduce (communicator exists): {timeit(lambda: model.iterate(pack, optimizer=opt)):.3f} ms/step", flush=True)
    if os.environ.get("INIT_PG") == "2":
        dist.destroy_process_group()
        print(f"after destroy_process_group: {timeit(lambda: model.iterate(pack, optimizer=opt)):.3f} ms/step", flush=True)
