"""BaseTrainer: the older twin of Trainer11 in the reference (Trainer/base_trainer.py:14-123), named by
BASELINE.json's north_star ("behind the existing Trainer/base_trainer.py + Predictor.Models API surface").

Same constructor fields, same loop shape (iterate() per minibatch, log / eval / save cadences, checkpoints
e{epoch}_s{step}.model/.opt), same differences from Trainer11 as in the reference:
  * train(from_ckpt=None) takes no epoch / step (base_trainer.py:42-49: the resume branch is commented out there;
    here from_ckpt = (exp_name, epoch, step) resumes, None trains from scratch);
  * save_ckpt(reference_score) receives the tracked metric (`reference` = '-loss': sign = direction, name = key of the
    metrics pack, base_trainer.py:27, 41, 63-67); the best score is tracked in `self.best`;
  * evaluation always logs under 'dev/' (base_trainer.py:117).
TensorBoard is replaced by JSON lines, MetricsManager's string round trip by plain means (SURVEY.md section 2 row 4:
out of scope)."""
import ctypes
import gc

from .trainer11 import Trainer11


class BaseTrainer(Trainer11):
    reference = "-loss"

    def __init__(self, optimizer, model, train_iter, dev_iter, test_iter, ckpt_root="ckpt/", exp_name="base_exp", log_every_iter=100,
                 eval_every_iter=1000, save_every_iter=5000, drop_exp=True, log_path=None):
        super().__init__(optimizer, model, train_iter, dev_iter=dev_iter, test_iter=test_iter, ckpt_root=ckpt_root, exp_name=exp_name,
                         log_every_iter=log_every_iter, eval_every_iter=eval_every_iter, save_every_iter=save_every_iter, drop_exp=drop_exp,
                         log_path=log_path)
        assert self.reference[0] in ["-", "+"]                     # base_trainer.py:39
        self.best = 1e10 if self.reference[0] == "-" else 0

    def train(self, from_ckpt=None):
        self.best = 1e10 if self.reference[0] == "-" else 0       # base_trainer.py:42
        if from_ckpt is not None:
            self.load_from_ckpt(*from_ckpt)
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
        metrics = None
        for data in self.train_iter:
            metrics, _ = self.model.iterate(data, optimizer=self.optimizer, is_train=True)
            if self.global_step % self.log_every_iter == 0 and self.global_step != 0:
                self.summarize(metrics, "train/")
            self.global_step += 1
            if self.dev_iter is not None and self.global_step % self.eval_every_iter == 0 and self.global_step != 0:
                self.evaluate(self.dev_iter, "dev/")
            if self.global_step % self.save_every_iter == 0 and self.global_step != 0:
                self.save_ckpt(metrics[self.reference[1:]])
        if metrics is not None:
            self.save_ckpt(metrics[self.reference[1:]])
        if self.test_iter is not None:
            self.evaluate(self.test_iter, "test/")

    def save_ckpt(self, reference_score=None):
        super().save_ckpt()
        if reference_score is not None:
            score = float(reference_score.detach().float().cpu().reshape(-1)[0])
            if (self.reference[0] == "-" and score < self.best) or (self.reference[0] == "+" and score > self.best):
                self.best = score

    def evaluate(self, dev_iter, prefix="dev/"):
        return super().evaluate(dev_iter, "dev/")                 # base_trainer.py:117 always writes under 'dev/'
