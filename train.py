#!/usr/bin/env python3
"""train.py - the reference's main.py (main.py:14-36, 38-41, 55-98) on the MI355X path.

    python train.py --model_name=TransformerOffical --batch_size=64 --warm_up=4000 --num_epoch=200 \
        --collector_path=data/collector --vocab_path=Predictor/vocab.t

Same flow: TrainConfig() <- fn_build(kwargs) ; ModelConfig merged over it (fn_combine) ; fn_build(kwargs) again ;
Vocab.load(vocab_path) ; build_dataloader(part = train / test / dev) ; Model(config, vocab).cuda() ;
Adam(lr=3e-4, betas=(0.9, 0.98), eps=1e-9) under NoamOpt(d_model, 1, warm_up) ; Trainer11(...).train().
Flags are `--key=value` pairs as fire would parse them (fire is not a dependency here); unknown keys are ADDED to the
config exactly as BaseConfig.fn_build does in the reference (base_config.py:7-15).

Differences that come with the MI355X path:
  * features are computed on the GPU per batch, so `predump` / `use_old` (cached .t feature files) are accepted and ignored;
  * multi-GPU: launch one process per GPU (python -m torch.distributed.run --nproc-per-node N train.py ...): the model is
    wrapped in dist.DataParallel (bucketed RCCL all-reduce overlapped with backward) and the loaders are sharded by rank -
    the reference's only multi-GPU hook is the commented-out `model.wrap()` (main.py:80);
  * `--synthetic=N` trains on N synthetic AISHELL-1-shaped utterances (no dataset ships with this repository);
  * `--trainer=BaseTrainer` selects the twin of Trainer/base_trainer.py instead of Trainer11.
"""
import ast
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from asr_chinese_e2e_amd import Models  # noqa: E402
from asr_chinese_e2e_amd.data_handler import DataConfigAiShell1, Vocab, build_dataloader, synthetic_pack  # noqa: E402
from asr_chinese_e2e_amd.Trainer import BaseTrainer, FusedAdam, NoamOpt, Trainer11  # noqa: E402


class TrainConfig(DataConfigAiShell1):      # main.py:14-36
    lr = 1e-3
    batch_size = 16
    eval_batch_size = 16
    num_epoch = 20
    warm_up = 1000
    device_id = [0, 1]
    exp_name = None
    drop_exp = True
    ckpt_root = "ckpt/"
    log_every_iter = 100
    eval_every_iter = 20000
    save_every_iter = 10000
    from_ckpt = None
    from_epoch = None
    from_step = None
    reference = "-loss"
    model_name = "TransformerOffical"      # the reference's class default is the stub 'ExampleModel'; its CLI default (main.py:103) is this
    predump = False
    use_old = False
    collector_path = "data/collector"       # data_config.py:18-19
    vocab_path = "Predictor/vocab.t"
    synthetic = 0                           # > 0: that many synthetic utterances instead of the manifests
    synthetic_frames = 500
    synthetic_vocab = 4232
    trainer = "Trainer11"


def get_model_class(model_name):            # main.py:38-41
    Model = getattr(Models, model_name)
    return Model, Model.get_default_config()


def parse_flags(argv):
    """--key=value / --key value / --flag  ->  dict with Python literals where they parse (as fire does)."""
    out, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected argument {a!r} (flags are --key=value)")
        if "=" in a:
            k, v = a[2:].split("=", 1)
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            k, v = a[2:], argv[i + 1]
            i += 1
        else:
            k, v = a[2:], "True"
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        out[k.replace("-", "_")] = v
        i += 1
    return out


class _SyntheticLoader:
    """len(data) synthetic batches with the reference's batch contract, sharded by rank like the real loader."""

    def __init__(self, n_utt, batch, frames, feat, vocab, seed, rank, world):
        self.args = (batch, frames, feat, vocab)
        n = n_utt // batch
        self.seeds = [seed + i for i in range(n // world * world)][rank::world] if world > 1 else [seed + i for i in range(n)]

    def __len__(self):
        return len(self.seeds)

    def __iter__(self):
        B, T, F, V = self.args
        for s in self.seeds:
            yield synthetic_pack(B, T, F, V, seed=s, ragged=True, device="cuda", dtype=torch.bfloat16)


def train(**kwargs):                        # main.py:55-98
    print("\nStart training\n")
    config = TrainConfig()
    config.fn_build(kwargs)
    assert config.model_name
    Model, ModelConfig = get_model_class(config.model_name)
    config.fn_combine(ModelConfig())
    config.fn_build(kwargs)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        config.fn_show()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the training path has no CPU fallback")
    if world > 1:
        from asr_chinese_e2e_amd import dist as D
        D.init(os.environ.get("ASR_DIST_BACKEND", "nccl"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if config.synthetic:
        vocab = Vocab.synthetic(config.synthetic_vocab)
        F = config.n_mels * config.lfr_m
        mk = lambda n, b, seed: _SyntheticLoader(n, b, config.synthetic_frames, F, vocab.vocab_size, seed, rank, world)
        train_iter = mk(config.synthetic, config.batch_size, 1000)
        test_iter = mk(max(config.synthetic // 8, config.eval_batch_size), config.eval_batch_size, 5000)
        dev_iter = mk(max(config.synthetic // 8, config.eval_batch_size), config.eval_batch_size, 7000)
    else:
        vocab = Vocab.load(config.vocab_path)
        common = dict(collector_path=config.collector_path, vocab=vocab, sample_rate=config.sample_rate, window_size=config.window_size,
                      n_mels=config.n_mels, predump=config.predump, use_old=config.use_old, lfr_m=config.lfr_m, lfr_n=config.lfr_n,
                      rank=rank, world=world)
        train_iter = build_dataloader(batch_size=config.batch_size, part="train", augment=config.augment, **common)
        test_iter = build_dataloader(batch_size=config.eval_batch_size, part="test", augment=False, **common)
        dev_iter = build_dataloader(batch_size=config.eval_batch_size, part="dev", augment=False, **common)

    model = Model(config, vocab).cuda()
    optimizer = FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-09)      # main.py:81
    assert config.hidden_size
    optimizer = NoamOpt(config.d_model, 1, config.warm_up, optimizer)                      # main.py:83
    runner = model
    if world > 1:
        runner = D.DataParallel(model, torch.device("cuda", torch.cuda.current_device()))
    Trainer = {"Trainer11": Trainer11, "BaseTrainer": BaseTrainer}[config.trainer]
    exp_name = config.exp_name if world == 1 or config.exp_name is None else f"{config.exp_name}"
    trainer = Trainer(model=runner, optimizer=optimizer, train_iter=train_iter, dev_iter=dev_iter, test_iter=test_iter, exp_name=exp_name,
                      ckpt_root=config.ckpt_root if rank == 0 else os.path.join(config.ckpt_root, f"rank{rank}"),
                      eval_every_iter=config.eval_every_iter, log_every_iter=config.log_every_iter, save_every_iter=config.save_every_iter,
                      drop_exp=config.drop_exp)
    print(f"start trainning at {trainer.get_time()}\n")
    if config.from_ckpt is not None:
        if config.trainer == "BaseTrainer":
            trainer.train((config.from_ckpt, config.from_epoch, config.from_step))
        else:
            trainer.train(config.from_ckpt, config.from_epoch, config.from_step)
    else:
        trainer.train()
    print(f"done at {trainer.get_time()}\n")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return trainer


if __name__ == "__main__":
    train(**parse_flags(sys.argv[1:]))
