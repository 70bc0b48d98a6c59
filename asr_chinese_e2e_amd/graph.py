"""hipGraph capture of the whole training step (forward, losses, backward, clip, Noam/Adam).

The step is a fixed sequence of ~200-500 kernel launches on two streams; for the small-matrix
decoder part the host cannot enqueue them as fast as the GPU runs them (joint config: 8.6 ms of
Python per 6 ms of GPU work).  Every C-ABI entry point only enqueues work on the stream it is
given (no allocation, no sync - include/asr_hip.h), so the whole step is capturable: one graph
per batch shape, replayed with the batch copied into static input buffers.

Restrictions: dropout must be 0 (the dropout seed is a launch-time scalar, it would be frozen
into the graph); single-process only (the RCCL path stays eager - it cannot be validated on a
one-GPU box).  Capture has no side effects on the weights / optimizer state: the warm-up steps
needed to size workspaces run on a snapshot that is restored afterwards.
"""
import torch

from .Utils import Pack


class GraphedStep:
    def __init__(self, model, optimizer, example, warmup=2):
        from .Models.transformer_official import CLIP_NORM
        if not hasattr(optimizer, "fused_step"):
            raise TypeError("graph capture needs the fused optimizer path (Trainer.NoamOpt over FusedAdam)")
        eng = model._ensure_engine(example.wave.device)
        if eng.drop_p > 0 and model.training:
            raise ValueError("graph capture with dropout > 0 is not supported (seed is frozen at capture)")
        self.model, self.opt = model, optimizer
        self.keys = [k for k in ("wave", "wave_len", "tgt_for_input", "tgt_len", "tgt_for_metric") if example.get(k) is not None]
        self.static = Pack({k: example[k].clone() for k in self.keys})
        flat = model._flat
        optimizer._device_state(flat)
        step_dev = optimizer._dev[0]
        snap = [t.clone() for t in (flat.p, flat.m, flat.v)] + [step_dev.clone()]
        host_step, host_rate = optimizer._step, optimizer._rate
        seed = model._step_seed

        def body():
            model.zero_flat_grads()
            loss, pg = model.train_step(self.static)
            optimizer.fused_step(flat, CLIP_NORM)
            cer = model._cer_of(pg) if pg is not None else None      # the step's CER (iterate's `cer` key), scored inside the graph
            return loss, cer

        # A loader's helper thread (data_handler.loader) allocates pinned memory, grows the caching allocator and synchronises events while
        # it prepares the next batches: under torch.cuda.graph's default capture mode any such call on another thread invalidates the
        # capture.  loader.paused() holds every loader's gate for the warm-up and the capture (round-4 ADVICE).
        from .data_handler import loader as _loader
        with _loader.paused():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss, self.cer = body()
        # restore: warm-up / capture leave no trace
        flat.p.copy_(snap[0]); flat.m.copy_(snap[1]); flat.v.copy_(snap[2]); step_dev.copy_(snap[3])
        flat.refresh_lowp()
        optimizer._step, optimizer._rate = host_step, host_rate
        model._step_seed = seed
        self.shapes = {k: tuple(example[k].shape) for k in self.keys}

    def matches(self, pack):
        return all(tuple(pack[k].shape) == self.shapes[k] for k in self.keys)

    def __call__(self, pack):
        for k in self.keys:
            if pack[k] is not self.static[k]:
                self.static[k].copy_(pack[k], non_blocking=True)
        self.graph.replay()
        self.opt._step += 1                    # host mirror of the device-side counter
        self.opt._rate = self.opt.rate()
        m = Pack()
        m.add(loss=self.loss[0])
        if self.model.use_decoder and self.model.use_ctc:
            m.add(ce=self.loss[1], ctc=self.loss[2])
        if self.cer is not None:
            m.add(cer=self.cer)
        return m, None


class GraphedModel:
    """iterate() with the reference's signature; one captured graph per batch shape."""

    def __init__(self, model, max_graphs=16):
        self.model, self.graphs, self.max_graphs = model, {}, max_graphs

    def iterate(self, input, optimizer=None, is_train=True):
        if optimizer is None or not is_train:
            return self.model.iterate(input, optimizer, is_train)
        key = tuple(tuple(input[k].shape) for k in ("wave", "tgt_for_input"))
        g = self.graphs.get(key)
        if g is None:
            if len(self.graphs) >= self.max_graphs:
                return self.model.iterate(input, optimizer, is_train)
            g = self.graphs[key] = GraphedStep(self.model, optimizer, input)
        return g(input)
