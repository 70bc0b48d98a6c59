"""Epoch / step loop with the reference's shape (Trainer/trainer11.py:14-132): iterate() per
minibatch, checkpoints named e{epoch}_s{step}.model/.opt, periodic evaluation.  Scalars go where the reference's
`SummaryWriter(self.exp_root).add_scalar(tag, value, global_step)` puts them - a TensorBoard event file under
exp_root (`self.summary_writer`, written by Utils/tfevents.py: the tensorboard package is not a dependency) -
and, for scripts, to scalars.jsonl beside it."""
import ctypes
import gc
import datetime
import json
import os
import shutil
import time

import torch

from ..Utils.tfevents import EventFileWriter


class Trainer11:
    reference = "-loss"

    def __init__(self, optimizer, model, train_iter, dev_iter=None, test_iter=None, ckpt_root="ckpt/", exp_name="base_exp",
                 log_every_iter=100, eval_every_iter=1000, save_every_iter=5000, drop_exp=True, log_path=None):
        self.optimizer, self.model = optimizer, model
        self.train_iter, self.dev_iter, self.test_iter = train_iter, dev_iter, test_iter
        self.ckpt_root = ckpt_root
        self.exp_name = exp_name if exp_name is not None else self.get_time()
        self.log_every_iter, self.eval_every_iter, self.save_every_iter = log_every_iter, eval_every_iter, save_every_iter
        self.exp_root = os.path.join(self.ckpt_root, self.exp_name)
        self.global_step = 0
        self.global_epoch = 0
        if drop_exp and os.path.exists(self.exp_root):
            shutil.rmtree(self.exp_root)
        os.makedirs(self.exp_root, exist_ok=True)
        self.config = self.model.config
        self.log_path = log_path or os.path.join(self.exp_root, "scalars.jsonl")
        self.history = []
        self.summary_writer = EventFileWriter(self.exp_root)      # trainer11.py:38
        self._log_file = open(self.log_path, "a")

    def add_scalar(self, tag, value, step):
        rec = {"tag": tag, "value": float(value), "step": int(step)}
        self.history.append(rec)
        self.summary_writer.add_scalar(tag, rec["value"], rec["step"])
        self._log_file.write(json.dumps(rec) + "\n")

    def flush_logs(self):
        self.summary_writer.flush()
        self._log_file.flush()

    def train(self, from_ckpt=None, from_epoch=None, from_step=None):
        if from_ckpt is not None and from_epoch is not None and from_step is not None:
            self.load_from_ckpt(from_ckpt, from_epoch, from_step)
        # the model, the engine's plans and the loaders exist by now.  (1) Their construction left a few hundred MB of freed host memory at the
        # top of the C heap, which glibc hands back to the kernel at some later free() - a ~30-ms pause of this thread, ten training steps, at a
        # random step (found in bench.py's timed regions, round 4: DESIGN.md section 5): hand it back now.  (2) ~270 k collector-tracked objects
        # stay for the whole run; a full pass over them takes ~80 ms and the steps leave almost nothing to collect: collect once, then move what
        # exists out of the collector's way.
        gc.collect()
        try:
            ctypes.CDLL("libc.so.6").malloc_trim(0)
        except (OSError, AttributeError):
            pass
        gc.freeze()
        for _ in range(self.config.num_epoch):
            self.train_epoch()
            self.global_epoch += 1

    def train_epoch(self):
        self.model.train()
        t0, frames, utts = time.time(), 0, 0
        for i, data in enumerate(self.train_iter):
            metrics, _ = self.model.iterate(data, optimizer=self.optimizer, is_train=True)
            self.add_scalar("lr", self.optimizer.rate(), self.global_step)
            if self.global_step % self.log_every_iter == 0 and self.global_step != 0:
                self.summarize(metrics, "train/")
            self.global_step += 1
            frames += int(data.wave.size(0) * data.wave.size(1))
            utts += int(data.wave.size(0))
            if self.dev_iter is not None and self.global_step % self.eval_every_iter == 0:
                self.evaluate(self.dev_iter, "dev/")
            if self.global_step % self.save_every_iter == 0:
                self.save_ckpt()
        torch.cuda.synchronize()
        dt = time.time() - t0
        self.add_scalar("train/utt_per_s", utts / max(dt, 1e-9), self.global_step)
        self.add_scalar("train/frames_per_s", frames / max(dt, 1e-9), self.global_step)
        self.flush_logs()
        self.save_ckpt()
        if self.test_iter is not None:
            self.evaluate(self.test_iter, "test/")

    def load_from_ckpt(self, exp_name, epoch, step):
        prefix = f"e{epoch}_s{step}"
        self.global_step, self.global_epoch = step, epoch
        self.model.load(os.path.join(self.ckpt_root, exp_name, prefix + ".model"))
        # a fresh model allocates its flat HBM buffers at the first step: allocate them now when it already sits on
        # the GPU, so that Adam's moments land in them (otherwise NoamOpt keeps the state until the first fused step)
        ensure = getattr(self.model, "_ensure_engine", None)
        dev = next(self.model.parameters()).device
        if ensure is not None and dev.type == "cuda":
            ensure(dev)
        self.optimizer.load(os.path.join(self.ckpt_root, exp_name, prefix + ".opt"), getattr(self.model, "_flat", None))
        self.config = self.model.config

    def save_ckpt(self):
        prefix = f"e{self.global_epoch}_s{self.global_step}"
        self.model.save(os.path.join(self.exp_root, prefix + ".model"))
        self.optimizer.save(os.path.join(self.exp_root, prefix + ".opt"))

    def summarize(self, pack, prefix="train/"):
        for k in pack:
            self.add_scalar(prefix + k, pack[k].detach().float().cpu().reshape(-1)[0], self.global_step)

    def evaluate(self, dev_iter, prefix="dev/"):
        self.model.eval()
        sums, n = {}, 0
        with torch.no_grad():
            for data in dev_iter:
                metrics, _ = self.model.iterate(data, is_train=False)
                for k, v in metrics.items():
                    sums[k] = sums.get(k, 0.0) + float(v.detach().float().cpu().reshape(-1)[0])
                n += 1
        for k, v in sums.items():
            self.add_scalar(prefix + k, v / max(n, 1), self.global_step)
        self.model.train()
        return {k: v / max(n, 1) for k, v in sums.items()}

    def get_time(self):
        return (datetime.datetime.now() + datetime.timedelta(hours=8)).strftime("%Y%m%d%H%M")
