"""Noam-scheduled Adam with the reference's interface (Trainer/optimizer.py:4-46, main.py:81-83)
on the fused HIP optimizer kernels.

    adam = FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9)
    opt  = NoamOpt(config.d_model, 1, config.warm_up, adam)        # same call as main.py:83

NoamOpt keeps the reference's host-side attributes (_step, _rate, rate(), zero_grad(), step(),
save(), load()) because the trainer reads them (trainer11.py:58, 74).  When the wrapped optimizer
is a FusedAdam and the model stores its parameters flat (engine.FlatParams), `fused_step` runs
grad-norm + clip + Noam LR + Adam + bf16 shadow refresh as three kernel launches over the flat
buffers, with the step counter and learning rate computed on the device (graph-capturable).
Any other torch.optim optimizer still works through step() (per-tensor path).
"""
import torch

from .. import kernels as K


class FusedAdam(torch.optim.Optimizer):
    """Adam with torch.optim.Adam's constructor; state lives in the model's flat buffers when
    driven through NoamOpt.fused_step, else per-parameter tensors."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._t = 0
        self._ws = None

    @torch.no_grad()
    def step(self, closure=None):
        """Per-tensor path (no clipping): one asr_adam_step launch per parameter."""
        self._t += 1
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["hyper"] = torch.zeros(4, device=p.device)
                st["hyper"].copy_(torch.tensor([group["lr"], 1 - b1 ** self._t, (1 - b2 ** self._t) ** 0.5, self._t]))
                K.adam_step(p.data.view(-1), p.grad.view(-1), st["m"].view(-1), st["v"].view(-1), None, st["hyper"], None,
                            0.0, b1, b2, group["eps"], write_clipped=False)


class NoamOpt:
    "Optim wrapper that implements rate (Trainer/optimizer.py:4-31)."

    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer = optimizer
        self._step = 0
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self._rate = 0
        self._dev = None   # (step int32[1], hyper f32[4], sumsq f32[1]) on the flat buffers' device
        self._flat = None

    def rate(self, step=None):
        if step is None:
            step = self._step
        return self.factor * ((self.model_size ** -0.5) * min(step ** -0.5, step * (self.warmup ** -1.5)))

    def step(self):
        self._step += 1
        rate = self.rate()
        for g in self.optimizer.param_groups:
            g["lr"] = rate
        self._rate = rate
        self.optimizer.step()

    def zero_grad(self):
        # gradients are views of one flat buffer that the model zeroes with a single memset;
        # never set them to None (that would detach the views)
        if self.optimizer is not None and not isinstance(self.optimizer, FusedAdam):
            self.optimizer.zero_grad(set_to_none=False)

    # ---- fused path -------------------------------------------------------------------------
    def _device_state(self, flat):
        if self._dev is None or self._flat is not flat or self._dev[0].device != flat.p.device:
            dev = flat.p.device
            self._dev = (torch.full((1,), self._step, dtype=torch.int32, device=dev), torch.zeros(4, device=dev),
                         torch.zeros(1, device=dev), K.Workspace(dev))
            self._flat = flat
        return self._dev

    def fused_step(self, flat, max_norm, after_norm=None):
        """clip_grad_norm_(max_norm) + NoamOpt.step() + Adam.step() over the flat buffers."""
        if not isinstance(self.optimizer, FusedAdam):
            params = [p for g in self.optimizer.param_groups for p in g["params"]]
            torch.nn.utils.clip_grad_norm_(params, max_norm)
            self.step()
            flat.refresh_lowp()
            return
        step, hyper, sumsq, ws = self._device_state(flat)
        g = self.optimizer.param_groups[0]
        b1, b2 = g["betas"]
        K.grad_sumsq(flat.g, sumsq, ws)
        K.noam_hyper(step, hyper, self.model_size, self.warmup, self.factor, 0.0, b1, b2)
        K.adam_step(flat.p, flat.g, flat.m, flat.v, flat.lp, hyper, sumsq, max_norm, b1, b2, g["eps"], write_clipped=True)
        flat.version += 1               # the bf16 weights changed: transposed copies (engine.refresh_transposes) are stale
        self._step += 1                 # host mirror of the device counter (no sync)
        self._rate = self.rate()
        for gr in self.optimizer.param_groups:
            gr["lr"] = self._rate
        self.last_grad_sumsq = sumsq

    # ---- checkpoint (Trainer/optimizer.py:33-46; file layout: dict with the same keys) -----------
    def save(self, path):
        state = {"step": self._step, "factor": self.factor, "model_size": self.model_size, "rate": self._rate}
        if self._flat is not None and isinstance(self.optimizer, FusedAdam):
            state["opt_state"] = {"flat_m": self._flat.m.cpu(), "flat_v": self._flat.v.cpu(),
                                  "index": {k: list(v[1]) + [v[0]] for k, v in self._flat.index.items()}}
        else:
            state["opt_state"] = self.optimizer.state_dict()
        torch.save(state, path)
        print(f"opt saved to {path}")

    def load(self, path, flat=None):
        state = torch.load(path, map_location="cpu", weights_only=True)
        self._step, self.factor = state["step"], state["factor"]
        self.model_size, self._rate = state["model_size"], state["rate"]
        os_ = state["opt_state"]
        flat = flat if flat is not None else self._flat
        if isinstance(os_, dict) and "flat_m" in os_:
            if flat is None:
                raise RuntimeError("pass the model's flat buffers (model._flat) to load a fused optimizer state")
            flat.m.copy_(os_["flat_m"])
            flat.v.copy_(os_["flat_v"])
        else:
            self.optimizer.load_state_dict(os_)
        if self._dev is not None:
            self._dev[0].fill_(self._step)
        print(f"opt loaded from {path}")
