"""Data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI.

The reference's only multi-GPU mechanism is an unused single-process torch.nn.DataParallel
wrapper (Predictor/Bases/base_model.py:9-21, call site commented out at main.py:80).  Utterances
are independent (every mask is per utterance, LayerNorm is per frame), so the path shards by
minibatch with ONE exchange step per iteration: a sum all-reduce of the gradients.

MI355X-first design:
  * gradients already live in one flat fp32 buffer laid out in forward order; backward finishes
    them from the END of the buffer towards the start, so buckets are contiguous slices - no
    gather/scatter copies, no per-parameter hooks;
  * the engine reports progress ("gradients at offsets >= o are final"); every bucket that lies
    wholly above the mark is all-reduced on a separate communication stream while backward continues;
  * xGMI is point-to-point (7 links per GPU): a few large buckets (default 16 MB) keep each
    transfer bandwidth-bound rather than latency-bound;
  * the loss normalisers are made global BEFORE backward, so the summed gradients equal a single-process run on
    the concatenated batch - no post-scaling pass over the gradients: the all-reduce of [non-pad token count,
    local batch size] is started on the communication stream at the START of the step and the compute stream waits
    for it only where the loss kernels need it (after the forward pass): neither the host nor the GPU ever idles on
    it, and ranks may hold different local batch sizes (both kernels take their divisor from device memory);
  * the wire format of the buckets is bf16 for bf16 models (67.6 MB per step at 33.8 M parameters instead of
    135 MB; xGMI rings are per-link bound): cast on the communication stream, summed by RCCL, cast back into the
    fp32 gradient buffer; fp32 models (parity mode) and ASR_DP_WIRE=fp32 keep fp32 on the wire.  The bf16 wire ROUNDS:
    each rank's gradient is rounded to 8 significant bits before the sum and the ring adds in bf16, so an element of the
    reduced gradient differs from the fp32 sum by up to ~2^-8 * (number of ranks) of the largest addend (bound checked
    by tests/test_train_loop_gpu.py::test_data_parallel_bf16_wire_two_ranks); with fp32 on the wire the summed
    gradients equal a single-process run on the concatenated batch up to summation order;
  * the clip norm is computed on the reduced gradients, identical on every rank.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:      # a default would collide with whatever else runs on the box; the launcher (torchrun, bench.py) sets it
            raise RuntimeError("MASTER_PORT is not set: start the ranks through torch.distributed.run / bench.py --gpus N, or export it")
        os.environ["MASTER_PORT"] = str(free_port())
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def free_port():
    """A TCP port that was free a moment ago (bound to port 0, then released) - for single-node rendezvous."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_buckets(block_ranges, numel, bucket_elems, force_cuts=()):
    """Contiguous (start, end) slices of a flat buffer, built from the END backwards (the order in
    which backward completes gradients), cut only at block starts.  Covers [0, numel) exactly.
    force_cuts: offsets (block starts) where a bucket must begin whatever its size - used to keep
    the LAST bucket, whose all-reduce nothing overlaps, small."""
    starts = sorted({s for s, _ in block_ranges} | {0})
    force = {int(o) for o in force_cuts}
    assert force <= set(starts), "forced cuts must be block starts"
    buckets, end = [], numel
    for s in reversed(starts):
        if end - s >= bucket_elems or s == 0 or (s in force and s < end):
            buckets.append((s, end))
            end = s
    assert buckets and buckets[-1][0] == 0 and sum(e - s for s, e in buckets) == numel
    return buckets  # first bucket = highest offsets = ready first


class GradBucketer:
    """All-reduces `flat_g` bucket by bucket as `ready(offset)` marks move down.
    wire_dtype torch.bfloat16: each bucket travels as bf16 (cast -> all-reduce -> cast back, all on the
    communication stream); None / torch.float32: the fp32 slice itself is reduced in place."""

    def __init__(self, flat_g, block_ranges, bucket_bytes=32 << 20, group=None, force_cuts=(), wire_dtype=None, avoid_streams=()):
        self.g = flat_g
        self.group = group
        self.buckets = make_buckets(block_ranges, flat_g.numel(), max(1, bucket_bytes // flat_g.element_size()), force_cuts)
        self.cuda = flat_g.is_cuda
        # NORMAL priority (a high-priority HIP stream next to the two compute streams doubled the step time on MI355X / ROCm 7: 11.0 vs
        # 5.2 ms, one rank), and on a hardware queue of its own: sharing one with the compute or weight-gradient stream blocks that
        # stream behind every bucket's event waits (engine.pick_stream; round 4: 7.0 instead of 3.1 ms per step at one rank)
        self.comm_stream = None
        if self.cuda:
            from . import engine as E
            avoid = [torch.cuda.current_stream()] + list(avoid_streams)
            avoid += [s for s in E.registered_streams(flat_g.device) if all(s is not a for a in avoid)]      # e.g. a loader's copy stream
            self.comm_stream = E.pick_stream(flat_g.device, avoid)
        self.wire = None
        if wire_dtype is not None and wire_dtype != flat_g.dtype:
            if not self.cuda:
                raise ValueError("a reduced-precision wire format needs the gradients on the GPU (the casts are HIP kernels)")
            self.wire = torch.empty(flat_g.numel(), dtype=wire_dtype, device=flat_g.device)
        self.next = 0
        self.works = []
        # events are pooled and reused round-robin (a wait captures the record it follows), as the engine does: a fresh
        # torch.cuda.Event per bucket and producer stream cost a hipEventCreate/Destroy pair each, every step
        self._events = [torch.cuda.Event() for _ in range(64)] if self.cuda else []
        self._ev_next = 0
        # communication-exposure probe (bench.py, N > 1): timing events around the compute stream's final wait for the
        # communication stream = the part of the all-reduce that backward did not cover
        self.measure_exposed = False
        self._exposed = []
        self.bytes_on_wire = sum((e - s) for s, e in self.buckets) * (self.wire.element_size() if self.wire is not None else flat_g.element_size())

    def begin(self):
        self.next = 0
        self.works = []

    def _launch(self, i, streams=None):
        s, e = self.buckets[i]
        view = self.g[s:e]
        if not self.cuda:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        for st in (streams or [torch.cuda.current_stream()]):   # producers of these gradients
            ev = self._event()
            ev.record(st)
            self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            # blocking-style calls: with the RCCL backend "wait" orders the communication stream behind the collective,
            # the host does not block (gloo, used by the one-GPU tests, does block the host: correctness only)
            if self.wire is None:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            else:
                from . import kernels as K
                w = self.wire[s:e]
                K.cast(view, w)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                K.cast(w, view)

    def _event(self):
        ev = self._events[self._ev_next]
        self._ev_next = (self._ev_next + 1) & 63
        return ev

    def ready(self, offset, streams=None):
        """Gradients at flat offsets >= offset are final (after the work already queued on
        `streams`): launch every bucket that lies fully above the mark."""
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= offset:
            self._launch(self.next, streams)
            self.next += 1

    def finish(self):
        """Launch what is left and make the current stream wait for every bucket."""
        self.ready(0)
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)                      # runs when backward's own work on the compute stream is done
                cur.wait_stream(self.comm_stream)
                e1.record(cur)                      # runs when the last bucket has arrived
                self._exposed.append((e0, e1))
            else:
                cur.wait_stream(self.comm_stream)

    def exposed_ms(self, reset=True):
        """Mean time per step the compute stream spent waiting for the communication stream after backward had finished
        (measure_exposed = True; call after a device synchronisation)."""
        if not self._exposed:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        if reset:
            self._exposed = []
        return ms


class _Counts:
    """The step's global normalisers: [non-pad token count, batch size] summed over the ranks, all-reduced on the
    communication stream while the forward pass runs."""

    def __init__(self, buf, comm_stream, group):
        self.buf, self.comm, self.group = buf, comm_stream, group
        self.event = None

    def start(self, n_valid, B):
        """n_valid: 1-element f32 device tensor of this rank (written on the current stream); B: local batch size."""
        cur = torch.cuda.current_stream()
        self.buf[0:1].copy_(n_valid)
        self.buf[1].fill_(float(B))
        if self.event is None:      # two events, reused every step (a wait captures the record it follows)
            self._fork_ev, self.event = torch.cuda.Event(), torch.cuda.Event()
        self._fork_ev.record(cur)
        self.comm.wait_event(self._fork_ev)
        with torch.cuda.stream(self.comm):
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
            self.event.record(self.comm)
        return self

    def wait(self):
        """(global token count, global batch size) as 1-element device tensors; the current stream waits for them."""
        torch.cuda.current_stream().wait_event(self.event)
        return self.buf[0:1], self.buf[1:2]


class DataParallel:
    """Wraps a model built on engine.FlatParams; `iterate` has the reference's signature, every other attribute
    (config, train(), eval(), save(), load(), parameters(), ...) is the wrapped model's, as with the reference's
    Wrapper (Predictor/Bases/base_model.py:9-21)."""

    def __init__(self, model, device, bucket_bytes=16 << 20, reduce_loss=False, wire_dtype="auto", group=None):
        self.model = model
        self.reduce_loss = reduce_loss
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        eng = model._ensure_engine(device)
        flat = model._flat
        if wire_dtype == "auto":      # bf16 models send bf16 gradients; ASR_DP_WIRE=fp32|bf16 overrides
            env = os.environ.get("ASR_DP_WIRE", "")
            wire_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(env, torch.bfloat16 if flat.lp is not None else torch.float32)
        # The all-reduce of the last bucket (lowest offsets = first encoder layer + input projection) starts when
        # backward ends and is fully exposed: cut it at the feed-forward block of encoder layer 0, whose gradients
        # are final one block earlier (the engine raises a mark there), so only ~4 MB remain for the very end.
        tail = eng.tail_mark_name()
        compute_streams = (eng.side, eng.ctc_stream) if flat.g.is_cuda else ()
        self.bucketer = GradBucketer(flat.g, flat.block_range, bucket_bytes, group=group, force_cuts=[flat.index[tail][0]] if tail else (),
                                     wire_dtype=wire_dtype, avoid_streams=compute_streams)
        # BEFORE the process group's first collective (it takes its internal stream from torch's pool then): leave the pool's counter in
        # front of a stream that runs beside the compute streams (engine.steer_stream_pool) - RCCL's kernels will run on it
        self.pool_draws = 0
        if flat.g.is_cuda:
            from . import engine as E
            avoid = [torch.cuda.current_stream()] + list(compute_streams)
            avoid += [s for s in E.registered_streams(flat.g.device) if all(s is not a for a in avoid)]
            self.pool_draws = E.steer_stream_pool(flat.g.device, avoid)
        dist.broadcast(flat.p, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)          # identical replicas
        flat.refresh_lowp()
        self._counts = _Counts(torch.zeros(2, dtype=torch.float32, device=flat.g.device), self.bucketer.comm_stream, group)
        eng.grad_ready = self.bucketer.ready

    def __getattr__(self, name):           # only called when normal lookup fails: forward to the wrapped model
        return getattr(self.__dict__["model"], name)

    def iterate(self, input, optimizer=None, is_train=True):
        model = self.model
        if optimizer is None or not is_train:
            return model.iterate(input, optimizer, is_train)
        from .Models.transformer_official import CLIP_NORM
        from .Utils import Pack
        model.zero_flat_grads()
        self.bucketer.begin()
        B = input.wave.shape[0]
        loss, pg = model.train_step(input, count_hook=self._counts.start)
        self.bucketer.finish()
        optimizer.fused_step(model._flat, CLIP_NORM)
        cer = model._cer_of(pg) if pg is not None else None      # of this rank's utterances, computed on the device (beside the backward pass)
        if not self.reduce_loss:      # rank-local loss estimate (no extra collective per step)
            metrics = Pack()
            metrics.add(loss=loss[0])
            if cer is not None:
                metrics.add(cer=cer)
            return metrics, None
        # CE was normalised by the GLOBAL token count (sum over ranks = global CE); the CTC term the kernel reports is
        # the mean over the LOCAL batch: weight it by local / global batch before the sum over ranks
        loss = loss.clone()
        loss[2] = loss[2] * (float(B) / self._counts.buf[1])
        dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        lam = model.ctc_weight
        ce, ctc = loss[1], loss[2]
        total = ((1.0 - lam) * ce if model.use_ctc else ce) if model.use_decoder else 0.0
        if model.use_ctc:
            total = total + lam * ctc
        metrics = Pack()
        metrics.add(loss=total)
        if cer is not None:
            metrics.add(cer=cer)
        return metrics, None
