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
    keeps_grad_views = True    # zero_grad() never sets .grad to None (the model re-checks the views otherwise)

    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer = optimizer
        self._step = 0
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self._rate = 0
        self._dev = None   # (step int32[1], hyper f32[4], sumsq f32[1]) on the flat buffers' device
        self._flat = None
        self._pending_state = None   # opt_state loaded before the model allocated its flat buffers

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
        if self._pending_state is not None:          # checkpoint loaded before the flat buffers existed
            self._apply_adam_state(flat, self._pending_state)
            self._pending_state = None
        g = self.optimizer.param_groups[0]
        b1, b2 = g["betas"]
        K.grad_sumsq_noam(flat.g, sumsq, ws, step, hyper, self.model_size, self.warmup, self.factor, 0.0, b1, b2)      # squared norm + Noam rate: two launches
        K.adam_step(flat.p, flat.g, flat.m, flat.v, flat.lp, hyper, sumsq, max_norm, b1, b2, g["eps"], write_clipped=True)
        flat.version += 1               # the bf16 weights changed: transposed copies (engine.refresh_transposes) are stale
        self._step += 1                 # host mirror of the device counter (no sync)
        self._rate = self.rate()
        for gr in self.optimizer.param_groups:
            gr["lr"] = self._rate
        self.last_grad_sumsq = sumsq

    # ---- checkpoint (Trainer/optimizer.py:33-46): the SAME file layout as the reference --------------------
    # {'opt_state': torch.optim.Adam.state_dict(), 'step', 'factor', 'model_size', 'rate'}: a reference-trained .opt
    # loads here and a file written here loads into the reference's NoamOpt(torch.optim.Adam).  The fused path keeps
    # Adam's moments in the model's flat buffers; they are sliced into / filled from the per-parameter
    # exp_avg / exp_avg_sq entries, keyed by parameter order (= model.parameters() order, as torch does).
    def _flat_slices(self, flat):
        """[(offset, numel, shape)] of the optimizer's parameters inside the flat buffers, in param_groups order."""
        out = []
        base = flat.p.data_ptr()
        for g in self.optimizer.param_groups:
            for p in g["params"]:
                off = (p.data_ptr() - base) // 4
                if p.device != flat.p.device or off < 0 or off + p.numel() > flat.p.numel() or (p.data_ptr() - base) % 4:
                    raise RuntimeError("the optimizer's parameters are not views of the model's flat buffer "
                                       "(build the optimizer from model.parameters() of the model it steps)")
                out.append((off, p.numel(), tuple(p.shape)))
        return out

    def _adam_state_dict(self, flat):
        """torch.optim.Adam.state_dict() of the fused state (moments on the CPU)."""
        template = torch.optim.Adam([torch.zeros(1)], lr=self._rate or self.optimizer.param_groups[0]["lr"],
                                    betas=self.optimizer.param_groups[0]["betas"], eps=self.optimizer.param_groups[0]["eps"])
        groups, state, k = [], {}, 0
        m, v = flat.m.cpu(), flat.v.cpu()
        slices = self._flat_slices(flat)
        for g in self.optimizer.param_groups:
            d = dict(template.state_dict()["param_groups"][0])
            d.update({key: val for key, val in g.items() if key != "params"})
            d["params"] = list(range(k, k + len(g["params"])))
            groups.append(d)
            k += len(g["params"])
        if self._step > 0:
            for i, (off, n, shape) in enumerate(slices):
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": m[off:off + n].view(shape).clone(),
                            "exp_avg_sq": v[off:off + n].view(shape).clone()}
        return {"state": state, "param_groups": groups}

    def _apply_adam_state(self, flat, os_):
        """Fill the flat moments from a torch.optim.Adam.state_dict() (or this package's round-1 flat layout)."""
        if "flat_m" in os_:                      # round-1 files: whole flat buffers + their index
            idx = os_.get("index")
            if idx is not None and {k: list(v[1]) + [v[0]] for k, v in flat.index.items()} != {k: list(v) for k, v in idx.items()}:
                raise RuntimeError("optimizer checkpoint was written for a different parameter layout")
            flat.m.copy_(os_["flat_m"])
            flat.v.copy_(os_["flat_v"])
            return
        slices = self._flat_slices(flat)
        state = os_["state"]
        n_saved = sum(len(g["params"]) for g in os_["param_groups"])
        if n_saved != len(slices):
            raise RuntimeError(f"optimizer checkpoint holds {n_saved} parameters, the model has {len(slices)}")
        flat.m.zero_()
        flat.v.zero_()
        for i, (off, n, shape) in enumerate(slices):
            st = state.get(i, state.get(str(i)))
            if st is None:                       # no step taken yet for this parameter (torch omits the entry)
                continue
            if tuple(st["exp_avg"].shape) != shape:
                raise RuntimeError(f"optimizer checkpoint: parameter {i} has shape {tuple(st['exp_avg'].shape)}, the model {shape}")
            flat.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            flat.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))

    def save(self, path):
        state = {"step": self._step, "factor": self.factor, "model_size": self.model_size, "rate": self._rate}
        if isinstance(self.optimizer, FusedAdam) and self._pending_state is not None:
            state["opt_state"] = self._pending_state         # loaded but never stepped: pass it through
        elif self._flat is not None and isinstance(self.optimizer, FusedAdam):
            state["opt_state"] = self._adam_state_dict(self._flat)
        elif isinstance(self.optimizer, FusedAdam):          # never stepped, nothing loaded
            state["opt_state"] = torch.optim.Adam([torch.zeros(1)], lr=self.optimizer.param_groups[0]["lr"], betas=self.optimizer.param_groups[0]["betas"],
                                                  eps=self.optimizer.param_groups[0]["eps"]).state_dict()
            state["opt_state"]["param_groups"][0]["params"] = list(range(sum(len(g["params"]) for g in self.optimizer.param_groups)))
        else:
            state["opt_state"] = self.optimizer.state_dict()
        torch.save(state, path)
        print(f"opt saved to {path}")

    def load(self, path, flat=None):
        state = torch.load(path, map_location="cpu", weights_only=True)
        self._step, self.factor = state["step"], state["factor"]
        self.model_size, self._rate = state["model_size"], state["rate"]
        os_ = state["opt_state"]
        if isinstance(self.optimizer, FusedAdam):
            flat = flat if flat is not None else self._flat
            if flat is not None and flat.m is not None:
                self._apply_adam_state(flat, os_)
                self._pending_state = None
            else:
                # the model has not allocated its flat buffers yet (fresh Model(...).cuda() before the first step):
                # the moments are applied by the first fused_step, which receives the buffers
                self._pending_state = os_
        else:
            self.optimizer.load_state_dict(os_)
        if self._dev is not None:
            self._dev[0].fill_(self._step)
        print(f"opt loaded from {path}")
