"""Trainer loop, checkpoint round trip, evaluation path and the RCCL data-parallel wrapper (one
rank) on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from tests.helpers import ROOT  # noqa: E402


def small_model(ctc_weight=0.3, cls="TransformerOffical", seed=0):
    from asr_chinese_e2e_amd import Models
    from asr_chinese_e2e_amd.data_handler import Vocab
    torch.manual_seed(seed)
    M = getattr(Models, cls)
    cfg = M.get_default_config()()
    cfg.fn_build(dict(n_mels=16, lfr_m=1, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, dropout=0.0,
                      ctc_weight=ctc_weight, dtype="fp32", num_epoch=2, warm_up=10))
    return M(cfg, Vocab.synthetic(40)), cfg


def batches(n, seed=0):
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    return [synthetic_pack(4, 24, 16, 40, seed=seed + i, ragged=True, Lmin=2, Lmax=6, device="cuda") for i in range(n)]


def test_trainer_checkpoint_and_eval(tmp_path):
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt, Trainer11
    model, cfg = small_model()
    model = model.cuda()
    opt = NoamOpt(cfg.d_model, 1, cfg.warm_up, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    data = batches(3)
    tr = Trainer11(opt, model, data, dev_iter=data[:1], test_iter=data[:1], ckpt_root=str(tmp_path), exp_name="exp", log_every_iter=1,
                   eval_every_iter=2, save_every_iter=3)
    tr.train()                                   # 2 epochs x 3 steps
    assert tr.global_step == 6 and opt._step == 6
    tags = {h["tag"] for h in tr.history}
    assert {"lr", "train/loss", "dev/loss", "test/loss", "train/utt_per_s"} <= tags
    losses = [h["value"] for h in tr.history if h["tag"] == "train/loss"]
    assert all(np.isfinite(losses))
    # files named like the reference (trainer11.py:93-99)
    assert os.path.isfile(tmp_path / "exp" / "e1_s6.model") and os.path.isfile(tmp_path / "exp" / "e1_s6.opt")
    # resume: a fresh model + optimizer loaded from the checkpoint continues identically
    m2, _ = small_model(seed=1)
    m2 = m2.cuda()
    o2 = NoamOpt(cfg.d_model, 1, cfg.warm_up, FusedAdam(m2.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    m2._ensure_engine("cuda")
    t2 = Trainer11(o2, m2, data, ckpt_root=str(tmp_path), exp_name="exp2")
    t2.ckpt_root = str(tmp_path)
    t2.load_from_ckpt("exp", 1, 6)
    assert o2._step == 6 and abs(o2._rate - opt._rate) < 1e-15
    for (n, a), (_, b) in zip(model.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n
    ma, _ = model.iterate(data[0], optimizer=opt)
    mb, _ = m2.iterate(data[0], optimizer=o2)
    assert abs(float(ma.loss) - float(mb.loss)) < 1e-6 * abs(float(ma.loss))
    # float atomics (embedding scatter-add) make the last bits run-dependent, and Adam turns the
    # round-off of ~zero gradients into +-lr: bound by a few lr, and require the bulk to agree
    for (n, a), (_, b) in zip(model.named_parameters(), m2.named_parameters()):
        d = (a - b).abs()
        assert float(d.max()) <= 3 * opt._rate, n
        assert float((d > 1e-6 + 1e-5 * b.abs()).float().mean()) < 0.05, n
    # evaluation path returns loss and cer without touching the weights
    before = model._flat.p.clone()
    ev, _ = model.iterate(data[1], is_train=False)
    assert np.isfinite(float(ev.loss)) and 0 <= float(ev.cer)
    assert torch.equal(before, model._flat.p)


def test_any_torch_optimizer_still_works():
    """The model also drives a stock torch.optim.Adam under the reference's NoamOpt semantics."""
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    data = batches(1)[0]
    ref_model, cfg = small_model(seed=3)
    ref_model = ref_model.cuda()
    ref_model._ensure_engine("cuda")
    sd = {k: v.clone() for k, v in ref_model.state_dict().items()}
    o1 = NoamOpt(cfg.d_model, 1, 10, FusedAdam(ref_model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    ref_model.iterate(data, optimizer=o1)
    m2, _ = small_model(seed=4)
    m2 = m2.cuda()
    m2.load_state_dict(sd)
    o2 = NoamOpt(cfg.d_model, 1, 10, torch.optim.Adam(m2.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    m2.iterate(data, optimizer=o2)
    for (n, a), (_, b) in zip(ref_model.named_parameters(), m2.named_parameters()):
        if not n.endswith("w_ks.bias"):
            assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), n


DP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from asr_chinese_e2e_amd import Models, dist as D
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
rank, world = D.init("nccl")
def build():
    torch.manual_seed(0)
    M = Models.TransformerOffical
    cfg = M.get_default_config()()
    cfg.fn_build(dict(n_mels=16, lfr_m=1, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, dropout=0.0, ctc_weight=0.3, dtype="fp32"))
    m = M(cfg, Vocab.synthetic(40)).cuda()
    return m, NoamOpt(64, 1, 10, FusedAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
pack = synthetic_pack(4, 24, 16, 40, seed=5, ragged=True, Lmin=2, Lmax=6, device="cuda")
m1, o1 = build()
m2, o2 = build()
dp = D.DataParallel(m2, "cuda", bucket_bytes=64 << 10, reduce_loss=True)
assert len(dp.bucketer.buckets) > 3
for _ in range(3):
    a, _ = m1.iterate(pack, optimizer=o1)
    b, _ = dp.iterate(pack, optimizer=o2)
    assert abs(float(a.loss) - float(b.loss)) < 1e-5 * abs(float(a.loss)), (float(a.loss), float(b.loss))
for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
    assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), n
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("dp ok")
"""


def test_data_parallel_wrapper_one_rank_rccl(tmp_path):
    """world_size 1 over the nccl (= RCCL) backend: the bucketed all-reduce path, global loss
    normalisers and fused step give the same trajectory as the plain model."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "dp ok" in p.stdout


DP2_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from asr_chinese_e2e_amd import Models, dist as D
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
from asr_chinese_e2e_amd.Utils import Pack
rank, world = D.init("gloo")          # two ranks share the one GPU of the test box: gloo moves the CUDA buckets
torch.cuda.set_device(0)
mode = sys.argv[2]
def build():
    torch.manual_seed(0)
    M = Models.TransformerOffical if mode == "joint" else Models.TransformerCTC
    cfg = M.get_default_config()()
    cfg.fn_build(dict(n_mels=16, lfr_m=1, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, dropout=0.0,
                      ctc_weight=0.3 if mode == "joint" else 1.0, dtype="fp32"))
    m = M(cfg, Vocab.synthetic(40)).cuda()
    return m, NoamOpt(64, 1, 10, FusedAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
full = synthetic_pack(6, 24, 16, 40, seed=5, ragged=True, Lmin=2, Lmax=6, device="cuda")
lo, hi = (0, 3) if rank == 0 else (3, 6)
mine = Pack()
mine.add(**{k: full[k][lo:hi].contiguous() for k in ("wave", "wave_len", "tgt_for_input", "tgt_for_metric", "tgt_len")})
m1, o1 = build()                      # single process, whole batch
m2, o2 = build()                      # data parallel, half a batch per rank
dp = D.DataParallel(m2, "cuda", bucket_bytes=64 << 10, reduce_loss=True)
assert len(dp.bucketer.buckets) > 3
# (1) the reduced gradients of one backward equal the single-process gradients of the whole batch
m1._ensure_engine("cuda"); m1.zero_flat_grads(); m1.train_step(full)
m2.zero_flat_grads(); dp.bucketer.begin()
m2.train_step(mine, n_valid_override=dp._global_count if mode == "joint" else None, ctc_batch=6); dp.bucketer.finish()
torch.cuda.synchronize()
for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
    if n.endswith("w_ks.bias"):       # analytically zero gradient (softmax is shift-invariant): pure round-off
        continue
    sc = float(p.grad.abs().max()) + 1e-30
    assert float((p.grad - q.grad).abs().max()) <= 1e-4 * sc, (n, float((p.grad - q.grad).abs().max()), sc)
# (2) same loss trajectory through clip + Noam/Adam; parameters stay within a few learning rates
# (Adam, eps 1e-9, turns the round-off of near-zero gradient elements into +-lr)
for _ in range(3):
    a, _ = m1.iterate(full, optimizer=o1)
    b, _ = dp.iterate(mine, optimizer=o2)
    assert abs(float(a.loss) - float(b.loss)) < 2e-5 * abs(float(a.loss)), (float(a.loss), float(b.loss))
for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
    assert float((p.detach() - q.detach()).abs().max()) <= 3 * o1._rate, n
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("rank", rank, "dp2 ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ctc", "joint"])
def test_data_parallel_two_ranks_equals_single_process(tmp_path, mode):
    """two ranks with half the minibatch each (bucketed all-reduce overlapped with
    backward, global token count / global CTC batch normalisers, clip on the reduced gradients) follow
    the same trajectory as one process on the concatenated batch.  gloo carries the CUDA buckets here
    because the test box has one GPU; the RCCL path is the 1-rank test above and the driver's scaling run."""
    script = tmp_path / "dp2_worker.py"
    script.write_text(DP2_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551" if mode == "ctc" else "29552", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} dp2 ok" in o


def test_bucketed_wave_loader_feeds_the_model(tmp_path):
    """Waveforms -> bucketed loader (pinned copy on a side stream, log-mel / normalisation / SpecAugment /
    frame stacking on the GPU) -> the reference's batch contract -> a training step."""
    import random
    import wave
    from asr_chinese_e2e_amd import Models
    from asr_chinese_e2e_amd.data_handler import AudioParser, BucketedWaveLoader, Vocab, WaveDataset
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    from oracle import logmel_ref
    rng = np.random.RandomState(0)
    vocab = Vocab.synthetic(30)
    items = []
    for i in range(21):
        n = int(rng.randint(16000 // 2, 16000 * 2))
        w = (rng.randn(n) * 0.1).astype(np.float32)
        if i == 5:                                              # one utterance comes from a WAV file
            path = str(tmp_path / "u5.wav")
            with wave.open(path, "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
                f.writeframes((w * 32767).astype("<i2").tobytes())
            w = path
        items.append((w, [int(t) for t in rng.randint(4, 30, size=rng.randint(2, 7))]))
    ds = WaveDataset(items, vocab)
    parser = AudioParser(n_mels=40, lfr_m=4, lfr_n=3, device="cuda")
    loader = BucketedWaveLoader(ds, 4, parser=parser, augment=False, shuffle=True, seed=1, bucket_size=8, dtype=torch.float32)
    assert len(loader) == 6
    seen, packs = 0, []
    for pack in loader:
        B, T, F = pack.wave.shape
        assert F == 160 and pack.wave_len.max() == T and pack.tgt_for_input.shape[0] == B
        seen += B
        packs.append(pack)
    assert seen == 21
    # features of one utterance == the oracle front end on the same samples
    w0 = ds.wave(0)
    ref = logmel_ref.build_lfr(logmel_ref.utt_normalize(logmel_ref.log_mel(w0.astype(np.float64), 40)), 4, 3)
    single = BucketedWaveLoader(WaveDataset(items[:1], vocab), 1, parser=parser, shuffle=False, dtype=torch.float32)
    p0 = next(iter(single))
    assert int(p0.wave_len[0]) == ref.shape[0]
    assert np.allclose(p0.wave[0].cpu().numpy(), ref, rtol=2e-3, atol=2e-3)
    # and the batches train a model (SpecAugment on)
    M = Models.TransformerCTC
    cfg = M.get_default_config()()
    cfg.fn_build(dict(n_mels=40, lfr_m=4, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, dropout=0.0, ctc_weight=1.0, dtype="fp32"))
    model = M(cfg, vocab).cuda()
    opt = NoamOpt(64, 1, 10, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    aug = BucketedWaveLoader(ds, 4, parser=parser, augment=True, shuffle=True, seed=2, bucket_size=8, dtype=torch.float32)
    losses = [float(model.iterate(pack, optimizer=opt)[0].loss) for pack in aug]
    assert len(losses) == 6 and all(np.isfinite(losses))


def test_overfit_small_batch_then_decode_exactly():
    """End to end: a small joint model memorises four utterances (loss falls by > 10x), after which
    greedy CTC decoding and attention beam search both return the training transcripts."""
    from asr_chinese_e2e_amd import Models
    from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    torch.manual_seed(0)
    M = Models.TransformerOffical
    cfg = M.get_default_config()()
    cfg.fn_build(dict(n_mels=16, lfr_m=1, d_model=64, hidden_size=16, num_head=4, ff_size=128, layer_num=2, dropout=0.0, ctc_weight=0.3,
                      dtype="fp32", cross_mask="wave_len", cer_in_iterate=False))
    model = M(cfg, Vocab.synthetic(20)).cuda()
    opt = NoamOpt(64, 1, 60, FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
    pack = synthetic_pack(4, 40, 16, 20, seed=3, ragged=True, Lmin=3, Lmax=6, device="cuda")
    model.train()
    first = float(model.iterate(pack, optimizer=opt)[0].loss)
    labels = [[int(t) for t in row if int(t) != 0] for row in pack.tgt_for_input.cpu()]
    steps, last, ok = 1, first, False
    while steps < 3000 and not ok:                      # memorising four utterances takes a few hundred steps
        for _ in range(200):
            last = float(model.iterate(pack, optimizer=opt)[0].loss)
        steps += 200
        model.eval()
        ok = (last < 0.05 * first and model.ctc_greedy_search(pack) == labels and
              [h[0]["yseq"][1:-1] for h in model.beam_search(pack, beam_size=3, nbest=1, decode_max_len=10)] == labels)
        model.train()
    assert np.isfinite(last) and ok, (first, last, steps)
    model.eval()
    ev, _ = model.iterate(pack, is_train=False)
    assert float(ev.ctc_cer) < 1e-6            # same weights, same mode as the check above: exactly the transcripts
    assert float(ev.cer) <= 10.0               # teacher-forced argmax (a beam's best path need not be the greedy one)
