"""Explicit forward / backward engine of the Speech-Transformer over flat HBM buffers.

No autograd and no tracing compiler: the step is a fixed sequence of C-ABI kernel launches
(asr_chinese_e2e_amd.kernels) plus dense projections, written out by hand so that
  * every parameter lives in ONE flat fp32 buffer (+ one flat fp32 gradient buffer, + Adam m/v,
    + a bf16 shadow copy of the parameters that the MFMA GEMMs read); nn.Parameters are views,
  * the flat order is the reverse of the order in which backward finishes gradients, so
    data-parallel buckets are contiguous slices that can be all-reduced while backward is still
    running (dist.py),
  * fused projections (Q|K|V, cross-attention K|V) are plain views of adjacent parameters.

Reference call stack this replaces: TransformerOffical.forward -> Encoder/Decoder ->
MultiHeadAttention / PositionwiseFeedForwardUseConv -> cal_performance -> loss.backward()
(Predictor/Models/transformer_official.py:68-104 and what it calls).
"""
import math
import os

import torch

import collections
import ctypes

from . import _lib
from . import kernels as K
from ._lib import ACT_NONE, ACT_RELU, ACT_RELU_MASK

ALIGN = 64  # elements; keeps every block 256-byte (fp32) / 128-byte (bf16) aligned


def positional_encoding(max_len, d_model):
    """PositionalEncoding buffer, Predictor/Models/module.py:16-24."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    w = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * w)
    pe[:, 1::2] = torch.cos(pos * w)
    return pe.unsqueeze(0)


class FlatParams:
    """Flat fp32 parameter / gradient / Adam-state buffers with named views.

    `blocks` is a list of lists of (name, shape): the tensors of one block are laid out
    back-to-back (no padding inside a block, so e.g. w_qs|w_ks|w_vs form one (3*H*dk, d) matrix);
    every block starts at a multiple of ALIGN elements."""

    def __init__(self, blocks):
        self.blocks = blocks
        self.index = {}       # name -> (offset, shape)
        self.block_range = []  # (start, end) per block
        off = 0
        for blk in blocks:
            start = off
            for name, shape in blk:
                n = 1
                for s in shape:
                    n *= s
                self.index[name] = (off, tuple(shape))
                off += n
            self.block_range.append((start, off))
            off = (off + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.device = None
        self.p = self.g = self.m = self.v = self.lp = None

    def allocate(self, device, lowp):
        self.device = torch.device(device)
        z = lambda dt: torch.zeros(self.numel, dtype=dt, device=self.device)
        self.p, self.g, self.m, self.v = z(torch.float32), z(torch.float32), z(torch.float32), z(torch.float32)
        self.lp = z(torch.bfloat16) if lowp else None
        self.lpT = None      # transposed copies of the projection weights (engine.refresh_transposes), same offsets as lp
        self.version = 0     # bumped whenever lp is rewritten (refresh_lowp, fused optimizer step)
        self.lpT_version = -1   # value of `version` the transposed copies were made from

    def view(self, buf, name):
        off, shape = self.index[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def span(self, buf, first, last, shape):
        """One view over the adjacent tensors first..last (must be consecutive in a block)."""
        o0, _ = self.index[first]
        o1, s1 = self.index[last]
        n1 = 1
        for s in s1:
            n1 *= s
        n = 1
        for s in shape:
            n *= s
        assert o1 + n1 - o0 == n, f"{first}..{last} is not contiguous ({o1 + n1 - o0} != {n})"
        return buf[o0:o0 + n].view(shape)

    def refresh_lowp(self):
        if self.lp is not None:
            K.cast(self.p, self.lp)
            self.version += 1


class Linear:
    """y = x W^T + b with W (N,K) a view of the flat buffers.  bf16: hand-written MFMA kernels (asr_gemm_nt_bf16 forward and,
    with transposed weight copies, input gradient; asr_gemm_small_bf16 for the decoder's few rows; asr_gemm_tn_bf16 weight
    gradient).  fp32 (parity mode) and the bf16 shapes those kernels do not take (K or leading dimensions not multiples of 8):
    asr_gemm_f32, fp32 on the matrix cores.  No library GEMM anywhere (round 3), and never a CPU fallback."""

    def __init__(self, flat, w_names, b_names, N, Kdim):
        self.flat, self.N, self.K = flat, N, Kdim
        last = w_names[0].split(".")[-2]
        self.tag = {"w_qs": "qkv", "w_1": "w1", "w_2": "w2"}.get(last, last)      # short role name (qkv, fc, w1, w2, ...)
        self.w32 = flat.span(flat.p, w_names[0], w_names[-1], (N, Kdim))
        self.gw = flat.span(flat.g, w_names[0], w_names[-1], (N, Kdim))
        self.wlp = flat.span(flat.lp, w_names[0], w_names[-1], (N, Kdim)) if flat.lp is not None else None
        self.w_off = flat.index[w_names[0]][0]
        self.wlpT = None     # (K, N) bf16 view of flat.lpT when the engine keeps transposed copies
        if b_names:
            self.b32 = flat.span(flat.p, b_names[0], b_names[-1], (N,))
            self.gb = flat.span(flat.g, b_names[0], b_names[-1], (N,))
        else:
            self.b32 = self.gb = None

    def rows(self, lo, hi):
        """A Linear over output rows [lo, hi) of this one (e.g. the K|V part of Q|K|V)."""
        sub = object.__new__(Linear)
        sub.flat, sub.N, sub.K = self.flat, hi - lo, self.K
        sub.tag = self.tag + "_rows"
        sub.w32, sub.gw = self.w32[lo:hi], self.gw[lo:hi]
        sub.wlp = self.wlp[lo:hi] if self.wlp is not None else None
        sub.b32 = self.b32[lo:hi] if self.b32 is not None else None
        sub.gb = self.gb[lo:hi] if self.gb is not None else None
        sub.wlpT, sub.w_off = None, self.w_off + lo * self.K
        return sub

    # ---- forward
    SMALL_M = 1024      # rows up to which the 64 x 64-tile kernel serves a projection (the decoder's B*To)
    SMALL_REDUCE = 8192      # longest reduction it takes (the tied output projection's input gradient reduces over V = 4232)

    def small(self, x, reduce_len):
        """True when the small-M kernel (asr_gemm_small_bf16) takes this operand: few rows, bf16, 8-element alignment."""
        return (x.dtype == torch.bfloat16 and x.shape[0] <= Linear.SMALL_M and self.N % 8 == 0 and self.K % 8 == 0 and x.stride(0) % 8 == 0
                and x.stride(1) == 1 and x.data_ptr() % 16 == 0 and reduce_len <= Linear.SMALL_REDUCE)

    def fwd(self, x, act=ACT_NONE, out=None):
        M = x.shape[0]
        out = torch.empty(M, self.N, dtype=x.dtype, device=x.device) if out is None else out
        if self.small(x, self.K) and out.stride(0) % 4 == 0:
            return K.gemm_small(x, self.wlp, self.b32, out, trans_b=False, act=act)
        if x.dtype == torch.bfloat16 and K.gemm_nt_supported(M, self.N, self.K, x.stride(0), self.wlp.stride(0), out.stride(0)):
            K.gemm_nt(x, self.wlp, self.b32, out, act)
            return out
        if x.dtype == torch.float32:      # parity mode: fp32 on the matrix cores (asr_gemm_f32), ReLU in the store tail
            return K.gemm_f32(x, self.w32, out, bias=self.b32, trans_b=True, act=act)
        # bf16 shapes the bf16 kernels refuse (K or a leading dimension not a multiple of 8): through the fp32 kernel
        # (the bf16 shadow of the weight, as every bf16 kernel reads it - not the fp32 master)
        tmp = K.gemm_f32(x.float(), self.wlp.float(), torch.empty(M, self.N, dtype=torch.float32, device=x.device), bias=self.b32, trans_b=True, act=act)
        out.copy_(tmp)
        return out

    # ---- backward pieces
    def own_dgrad(self, dy, accumulate=False):
        """True when dgrad takes the persistent NT kernel (transposed weight copy) for this dy."""
        return (self.wlpT is not None and not accumulate and dy.shape[0] >= 4096 and dy.dtype == torch.bfloat16 and self.N % 8 == 0 and self.N >= 128
                and self.K % 8 == 0 and dy.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0)

    def _fresh_transpose(self):
        if self.flat.lpT_version != self.flat.version:
            # the copies are made at the start of every training step (Engine.refresh_transposes); a backward pass without
            # that call would multiply by last step's weights - refuse instead of silently taking another path
            raise RuntimeError("transposed weight copies are stale: call Engine.refresh_transposes() before the backward pass")

    def dgrad(self, dy, out=None, accumulate=False, relu_mask=None):
        """dx = dy W, or out += dy W when accumulate.  relu_mask: the activations of the ReLU in front of this projection -
        their backward mask is applied in the GEMM's store tail."""
        if dy.dtype == torch.float32:      # parity mode: fp32 on the matrix cores, W as stored
            out = torch.empty(dy.shape[0], self.K, dtype=dy.dtype, device=dy.device) if out is None else out
            return K.gemm_f32(dy, self.w32, out, act=ACT_RELU_MASK if relu_mask is not None else ACT_NONE, mask=relu_mask, accumulate=accumulate)
        if not accumulate and self.small(dy, self.N) and (out is None or out.stride(0) % 4 == 0):
            # few rows (decoder): dX = dY W straight from the weight as stored, on the 64 x 64-tile kernel (no transposed copy)
            out = torch.empty(dy.shape[0], self.K, dtype=dy.dtype, device=dy.device) if out is None else out
            return K.gemm_small(dy, self.wlp, None, out, trans_b=True, act=ACT_RELU_MASK if relu_mask is not None else ACT_NONE, mask=relu_mask)
        if accumulate and out is not None and relu_mask is None and self.own_dgrad(dy) and out.dtype == torch.bfloat16 and out.stride(0) % 8 == 0 \
                and out.data_ptr() % 16 == 0:
            self._fresh_transpose()
            return K.gemm_nt(dy, self.wlpT, None, out, res=out)      # out += dY W: residual add in the store tail, in place
        if self.own_dgrad(dy, accumulate):
            # dX = dY W as an NT product with the transposed weight copy: own MFMA kernel
            self._fresh_transpose()
            out = torch.empty(dy.shape[0], self.K, dtype=dy.dtype, device=dy.device) if out is None else out
            if relu_mask is not None:
                return K.gemm_nt(dy, self.wlpT, None, out, act=ACT_RELU_MASK, res=relu_mask)
            return K.gemm_nt(dy, self.wlpT, None, out)
        # what is left: bf16 shapes none of the bf16 kernels takes (odd vocabulary sizes, misaligned views, a many-row operand
        # without a transposed copy): through the fp32 kernel, W as stored - correct for every shape, never the bench's path
        res = K.gemm_f32(dy.float(), self.wlp.float(), torch.empty(dy.shape[0], self.K, dtype=torch.float32, device=dy.device),
                         act=ACT_RELU_MASK if relu_mask is not None else ACT_NONE, mask=relu_mask.float() if relu_mask is not None else None)
        if out is None:
            return res.to(dy.dtype)
        if accumulate:
            out.add_(res.to(out.dtype))
        else:
            out.copy_(res)
        return out

    def fused_bias_wgrad(self, dy, x):
        """True when wgrad can also produce the bias gradient (MFMA weight-gradient kernel path)."""
        return (self.gb is not None and dy.dtype == torch.bfloat16 and self.N % 8 == 0 and self.K % 8 == 0 and
                dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0)

    def wgrad(self, dy, x, with_bias=False, ws=None):
        """gw += dy^T x  (fp32 accumulation into the flat gradient buffer); with_bias (only when
        fused_bias_wgrad): also gb += column sums of dy, inside the same kernel.  ws: scratch for the deterministic mode."""
        M = x.shape[0]
        if dy.dtype == torch.bfloat16 and self.N % 8 == 0 and self.K % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0:
            K.gemm_tn(dy, x, self.gw, accumulate=True, dbias=self.gb if with_bias else None, ws=ws)
        elif dy.dtype == torch.bfloat16:      # odd shapes: through the fp32 kernel
            K.gemm_f32(dy.float(), x.float(), self.gw, trans_a=True, accumulate=True)
        else:                                 # parity mode: gw += dy^T x on the fp32 matrix-core kernel (one fixed-order sum per element)
            K.gemm_f32(dy, x, self.gw, trans_a=True, accumulate=True)

    def bgrad(self, dy, ws):
        if self.gb is not None:
            K.colsum(dy, self.gb, ws, accumulate=True)


class LayerNormP:
    def __init__(self, flat, prefix):
        self.g = flat.view(flat.p, prefix + ".weight")
        self.b = flat.view(flat.p, prefix + ".bias")
        self.gg = flat.view(flat.g, prefix + ".weight")
        self.gb = flat.view(flat.g, prefix + ".bias")


class MHA:
    """MultiHeadAttention of the reference (attention.py:6-62) over fused projections.  cross = False: Q | K | V adjacent (one (3 H dk, d)
    projection of the block's input).  cross = True (the decoder's encoder-decoder attention): Q on its own, K | V adjacent inside the
    matrix that holds the K | V projections of every decoder layer (cross_kv_param_blocks)."""

    def __init__(self, flat, pre, H, dk, d, cross=False):
        self.H, self.dk, self.d = H, dk, d
        hd = H * dk
        if cross:
            self.qkv = None
            self.q = Linear(flat, [pre + "w_qs.weight"], [pre + "w_qs.bias"], hd, d)
            self.kv = Linear(flat, [pre + "w_ks.weight", pre + "w_vs.weight"], [pre + "w_ks.bias", pre + "w_vs.bias"], 2 * hd, d)
            self.q.tag, self.kv.tag = "q_c", "kv_c"
        else:
            self.qkv = Linear(flat, [pre + "w_qs.weight", pre + "w_ks.weight", pre + "w_vs.weight"],
                              [pre + "w_qs.bias", pre + "w_ks.bias", pre + "w_vs.bias"], 3 * hd, d)
            self.q = self.qkv.rows(0, hd)
            self.kv = self.qkv.rows(hd, 3 * hd)
        self.fc = Linear(flat, [pre + "fc.weight"], [pre + "fc.bias"], d, hd)
        self.ln = LayerNormP(flat, pre + "layer_norm")


class FFN:
    def __init__(self, flat, pre, d, ff):
        self.w1 = Linear(flat, [pre + "w_1.weight"], [pre + "w_1.bias"], ff, d)
        self.w2 = Linear(flat, [pre + "w_2.weight"], [pre + "w_2.bias"], d, ff)
        self.ln = LayerNormP(flat, pre + "layer_norm")


def mha_param_block(pre, H, dk, d):
    hd = H * dk
    return [[(pre + "w_qs.weight", (hd, d)), (pre + "w_ks.weight", (hd, d)), (pre + "w_vs.weight", (hd, d))],
            [(pre + "w_qs.bias", (hd,)), (pre + "w_ks.bias", (hd,)), (pre + "w_vs.bias", (hd,))],
            [(pre + "layer_norm.weight", (d,))], [(pre + "layer_norm.bias", (d,))],
            [(pre + "fc.weight", (d, hd))], [(pre + "fc.bias", (d,))]]


def cross_kv_param_blocks(pres, H, dk, d):
    """w_ks | w_vs of the encoder-decoder attention of every decoder layer as ONE (L 2 H dk, d) matrix (+ one bias vector)."""
    hd = H * dk
    return [[(pre + n + ".weight", (hd, d)) for pre in pres for n in ("w_ks", "w_vs")],
            [(pre + n + ".bias", (hd,)) for pre in pres for n in ("w_ks", "w_vs")]]


def cross_q_param_block(pre, H, dk, d):
    """What is left of an encoder-decoder attention block beside its K | V projection (cross_kv_param_blocks)."""
    hd = H * dk
    return [[(pre + "w_qs.weight", (hd, d))], [(pre + "w_qs.bias", (hd,))],
            [(pre + "layer_norm.weight", (d,))], [(pre + "layer_norm.bias", (d,))],
            [(pre + "fc.weight", (d, hd))], [(pre + "fc.bias", (d,))]]


def ffn_param_block(pre, d, ff):
    return [[(pre + "w_1.weight", (ff, d, 1))], [(pre + "w_1.bias", (ff,))],
            [(pre + "w_2.weight", (d, ff, 1))], [(pre + "w_2.bias", (d,))],
            [(pre + "layer_norm.weight", (d,))], [(pre + "layer_norm.bias", (d,))]]


_STREAMS = {}
_HELD = []          # candidate streams that were rejected by pick_stream: kept alive, so that their hardware queue stays "in use" for later streams
QUEUE_PROBE = True  # False: take streams as they come (tests of the probe itself)


_CHAIN_BASE = {}


def _chain_ms(streams, n=40, cycles=50000, gate_cycles=8_000_000):
    """GPU time (ms) of n short spin kernels (one workgroup, ~20 us each) queued on EACH of `streams` and started TOGETHER: every stream first
    waits behind a gate - a long spin (~3 ms) on the first stream - while the host queues the chains, and the time is taken by events on the
    device, from the gate's end to the last chain's end.  (Round 4 timed the host's wall clock around the enqueue loop: on a slow or busy
    host - eight ranks starting together, a profiler - the launches of two chains take twice as long to ISSUE as one chain's, which read
    as a conflict for streams that run side by side.)  If the host did not finish queueing before the gate opened, the gate is lengthened."""
    for _ in range(4):
        torch.cuda.synchronize()
        gate = torch.cuda.Event()
        with torch.cuda.stream(streams[0]):
            torch.cuda._sleep(gate_cycles)
            gate.record()
        starts, ends = [], []
        for s in streams:
            with torch.cuda.stream(s):
                if s is not streams[0]:
                    s.wait_event(gate)
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                starts.append(e0)
        for _ in range(n):
            for s in streams:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
        for s in streams:
            with torch.cuda.stream(s):
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                ends.append(e1)
        queued_in_time = not gate.query()      # the gate was still closed when the last launch was queued
        torch.cuda.synchronize()
        if queued_in_time:
            return max(starts[0].elapsed_time(e) for e in ends)
        gate_cycles *= 4
    return max(starts[0].elapsed_time(e) for e in ends)      # a host this slow: the last (longest-gate) measurement, for what it is worth


def streams_conflict(a, b):
    """True when two streams cannot run side by side.  Measured, because no API tells: a chain of 40 short spin kernels goes to each stream
    at the same time.  tools/queue_probe.py on MI355X / ROCm 7 (100-kernel chains, 2.2 ms alone) shows three cases: 2.2 ms - the streams
    sit on different hardware queues served by different command-processor pipes: concurrent; 4.4 ms - the HIP runtime has mapped both
    onto ONE in-order hardware queue (it multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues per priority, in the order of first use);
    5.5 ms - two hardware queues on the SAME pipe, which alternates between them at a cost: worse than one queue."""
    dev = a.device.index
    if dev not in _CHAIN_BASE:
        _CHAIN_BASE[dev] = min(_chain_ms([a]) for _ in range(3))
    t = _chain_ms([a, b])
    if t > 1.5 * _CHAIN_BASE[dev]:      # a second look before a stream is rejected (a one-off stall of the box reads as a conflict)
        t = min(t, _chain_ms([a, b]))
    return t > 1.5 * _CHAIN_BASE[dev]


def pick_stream(device, avoid, factory=None, tries=12):
    """A stream that can run beside every stream in `avoid` (streams_conflict).
    Which hardware queue - and so which command-processor pipe - a stream gets depends on how many other streams the process has USED
    before it: a data loader's copy stream, the process group's internal stream, a second model.  Round 4 (tools/dp_probe.py): with the
    weight-gradient or auxiliary stream on the main stream's queue or pipe the CTC step takes 6.3 - 7.1 ms instead of 3.0 and the joint
    step 11.4 - 12.4 instead of 4.8 (every kernel of the trace ~40 us longer) - through the data-parallel wrapper at one rank in the CTC
    configuration, or after two or three unrelated streams had been used first.  So the streams are CHOSEN: candidates come from `factory`
    (default: torch's stream pool), each is tested against every stream in `avoid`; rejected candidates stay referenced (their queue then
    counts as used for whatever stream the process creates next)."""
    factory = factory or (lambda: torch.cuda.Stream(device=device))
    first, held = None, []
    for _ in range(tries):
        cand = factory()
        first = first if first is not None else cand
        if not QUEUE_PROBE or torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep"):      # (_sleep: the spin kernel of the test)
            return cand
        if not any(streams_conflict(a, cand) for a in avoid if a is not None):
            _HELD.extend(held)      # the rejected candidates stay referenced: their queues count as used for later streams
            return cand
        held.append(cand)
    # EVERY candidate conflicted: more likely the probe failing on this box than twelve unlucky draws - take the first candidate and hold none
    import warnings
    warnings.warn("asr_chinese_e2e_amd: no stream that runs beside the main stream after %d candidates (GPU_MAX_HW_QUEUES too small for the streams "
                  "of this process, or the probe cannot tell on this box): taking the first candidate; the multi-stream step may run serialised" % tries)
    return first


def steer_stream_pool(device, avoid, ring_max=64):
    """Advance torch's stream-pool counter until the NEXT stream the pool hands out runs beside every stream in `avoid`.
    The process group takes its internal stream - the one RCCL's kernels run on - from that pool (at::cuda::getStreamFromPool, a ring of 32
    streams per priority) at its first collective, and nothing lets the caller choose it; but the ring is deterministic: drawing streams
    until one repeats walks it once (every stream tested against `avoid` on the way), after which the position of the counter and the
    verdict for the stream behind it are known.  Returns how many draws it took, or -1 when no stream of the ring qualifies."""
    if not QUEUE_PROBE or torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep"):
        return 0
    order, verdict = [], {}
    with torch.cuda.device(device):
        s = torch.cuda.Stream(device=device)
        while s.cuda_stream not in verdict and len(order) < ring_max:
            verdict[s.cuda_stream] = not any(streams_conflict(a, s) for a in avoid if a is not None)
            order.append(s.cuda_stream)
            _HELD.append(s)
            s = torch.cuda.Stream(device=device)
        if s.cuda_stream not in verdict:
            return -1      # not a ring we understand: leave the counter where it is
        pos, n, draws = order.index(s.cuda_stream), len(order), len(order) + 1
        if not any(verdict.values()):
            return -1
        while not verdict[order[(pos + 1) % n]]:      # the stream after the last one drawn is what the next caller gets
            torch.cuda.Stream(device=device)
            pos, draws = pos + 1, draws + 1
        return draws


def shared_stream(device, kind):
    """One auxiliary stream of each kind per device and PROCESS, shared by every engine (engines of one process run their steps one
    after the other, so sharing costs nothing; a second set of streams would only compete for the few hardware queues), each on a
    hardware queue of its own (pick_stream): "wgrad" - the weight-gradient stream, lowest priority; "aux" (the CTC branch /
    cross-attention K|V work beside the decoder; normal priority: at the lowest 5.49 vs 5.48 ms, nothing); "comm" (dist.GradBucketer's
    all-reduces); "loader" (data_handler.loader's copies and front-end kernels).  A stream is tested against the current (main) stream
    and against EVERY stream registered before it for the device, whatever the order in which a process builds its loader, model and
    data-parallel wrapper (round-4 ADVICE: the loader's stream used to be picked against the main stream only)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, kind)
    if key not in _STREAMS:
        with torch.cuda.device(dev):
            avoid = [torch.cuda.current_stream()] + [s for (i, _), s in _STREAMS.items() if i == dev.index]
            _STREAMS[key] = pick_stream(dev, avoid, factory=(lambda: _side_stream(dev)) if kind.startswith("wgrad") else None)
    return _STREAMS[key]


_shared_stream = shared_stream


def registered_streams(device):
    """Every stream chosen so far for this device (weight-gradient, auxiliary, loader, ...): what a stream chosen later must run beside."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return [s for (i, _), s in _STREAMS.items() if i == idx]


def _side_stream(device):
    """The weight-gradient stream, at the LOWEST HIP stream priority: its
    GEMMs are off the critical path, and at equal priority they take CUs from the dgrad / attention / LayerNorm chain
    whenever both have workgroups pending (the LayerNorm backward ran 30 us beside them against 14 us alone).
    torch.cuda.Stream only offers normal / high, so the stream comes from asr_stream_create (the HIP runtime the library and
    torch share) and is wrapped as an ExternalStream (host plumbing; it lives as long as the process)."""
    import ctypes
    from . import _lib
    with torch.cuda.device(device):
        h = ctypes.c_void_p()
        # created inside libasr_hip.so, i.e. by the HIP runtime the kernels (and torch) are bound to: a second dlopen of
        # libamdhip64 by bare name could map another runtime whose stream handles mean nothing to this one
        _lib.check(_lib.lib.asr_stream_create(-1, ctypes.byref(h)), "asr_stream_create")
    return torch.cuda.ExternalStream(h.value, device=device)


class Engine:
    """Forward + backward of encoder (+ decoder) (+ CTC head) for one minibatch on one GPU."""
    PAD_HEAD_ROWS = True      # rows of the CTC head's logits / gradient on whole 128-byte lines (see self.ld_v)

    def __init__(self, flat, cfg, vocab_size, use_decoder, use_ctc, pe):
        self.flat, self.cfg = flat, cfg
        self.V = vocab_size
        self.use_decoder, self.use_ctc = use_decoder, use_ctc
        d, H, dk, ff = cfg.d_model, cfg.num_head, cfg.hidden_size, cfg.ff_size
        self.d, self.H, self.dk, self.ff = d, H, dk, ff
        self.d_in = cfg.n_mels * cfg.lfr_m
        self.L = cfg.layer_num
        self.dtype = torch.bfloat16 if flat.lp is not None else torch.float32
        self.pe = pe  # (>=maxlen, d) f32 on device
        self.ws = K.Workspace(flat.device)
        self.ws_side = K.Workspace(flat.device)   # scratch of the kernels issued on the side stream
        self.lin_in = Linear(flat, ["encoder.linear_in.weight"], ["encoder.linear_in.bias"], d, self.d_in)
        self.ln_in = LayerNormP(flat, "encoder.layer_norm_in")
        self.enc = [(MHA(flat, f"encoder.layer_stack.{i}.slf_attn.", H, dk, d), FFN(flat, f"encoder.layer_stack.{i}.pos_ffn.", d, ff))
                    for i in range(self.L)]
        if use_decoder:
            self.emb32 = flat.view(flat.p, "decoder.tgt_word_emb.weight")
            self.gemb = flat.view(flat.g, "decoder.tgt_word_emb.weight")
            self.prj = Linear(flat, ["decoder.tgt_word_emb.weight"], None, vocab_size, d)
            self.dec = [(MHA(flat, f"decoder.layer_stack.{i}.slf_attn.", H, dk, d), MHA(flat, f"decoder.layer_stack.{i}.enc_attn.", H, dk, d, cross=True),
                         FFN(flat, f"decoder.layer_stack.{i}.pos_ffn.", d, ff)) for i in range(self.L)]
            # the K | V projections of all decoder layers' encoder-decoder attention as ONE projection of the encoder output
            kvn = [f"decoder.layer_stack.{i}.enc_attn.{n}" for i in range(self.L) for n in ("w_ks", "w_vs")]
            self.kv_all = Linear(flat, [n + ".weight" for n in kvn], [n + ".bias" for n in kvn], self.L * 2 * H * dk, d)
            self.kv_all.tag = "kv_all"
        if use_ctc:
            self.ctc_lo = Linear(flat, ["ctc_lo.weight"], ["ctc_lo.bias"], vocab_size, d)
        # row stride of the CTC head's logits / gradient rows in the training step: whole 128-byte lines (bf16: multiples of 64
        # elements; V = 4232 -> 4288).  With dense rows of 8464 B every row of a GEMM tile straddles one line more and shares
        # its first and last line with the neighbouring tiles: head GEMM 110 -> 94 us, its input gradient 87 -> 76 us
        # (tools/gemm_bench.py pad).  Engine.PAD_HEAD_ROWS = False: dense rows (tests).
        self.ld_v = vocab_size
        if self.dtype == torch.bfloat16 and vocab_size % 8 == 0 and self.PAD_HEAD_ROWS:
            self.ld_v = (vocab_size + 63) // 64 * 64
        self.grad_ready = None  # callback(offset): gradients at flat offsets >= offset are final
        # Input gradients dX = dY W of the encoder projections run on the own persistent NT kernel, as NT products with
        # transposed bf16 weight copies (one batched transpose launch per step on the side stream, idle during the forward
        # pass), instead of the library GEMM: with the streaming store tail the own kernel is the faster one
        # (step 3.505 -> 3.428 ms; before that change the two were at parity).
        self._tr_tiles = None
        if flat.lp is not None:
            # encoder projections only: the decoder's B*To ~ 500 rows run on the 64 x 64-tile kernel with W as stored
            lins = [l for mha, ffn in self.enc for l in (mha.qkv, mha.fc, ffn.w1, ffn.w2)]
            # ... and the decoder's cross-attention K|V weights (one matrix for all layers): they multiply all B*T encoder frames in the
            # accumulating input gradient d_enc += dK|dV W_kv (the residual-add store tail of the own kernel, no library call)
            cross_qkv = [self.kv_all] if getattr(self, "dec", None) else []
            # ... and the CTC head (reduction over V = 4232 columns: the kernel's last k-step is ragged)
            head = [self.ctc_lo] if use_ctc else []
            lins = [l for l in lins + cross_qkv + head if l.N % 8 == 0 and l.K % 8 == 0]
            # copies sit at the offsets of their matrices, rows N elements apart; the CTC head's (rows of V = 4232 elements = 8464 B:
            # every row starts 16 B further into a 128-byte line) goes behind them with rows padded to whole lines (self.ld_v)
            n_flat = (flat.lp.numel() + 63) // 64 * 64
            pad_head = use_ctc and self.ld_v != self.V and self.ctc_lo in lins
            flat.lpT = torch.zeros(n_flat + (self.ctc_lo.K * self.ld_v if pad_head else 0), dtype=flat.lp.dtype, device=flat.device)
            tiles = []
            for l in lins:
                if pad_head and l is self.ctc_lo:
                    dst_off, ldd = n_flat, self.ld_v
                    l.wlpT = flat.lpT[n_flat:].view(l.K, ldd)[:, :l.N]
                else:
                    dst_off, ldd = l.w_off, l.N
                    l.wlpT = flat.lpT[l.w_off:l.w_off + l.N * l.K].view(l.K, l.N)
                tiles += [[l.w_off, l.N, l.K, (r << 16) | c, dst_off, ldd] for r in range((l.N + 63) // 64) for c in range((l.K + 63) // 64)]
            for i, (_, cross, _) in enumerate(self.dec if getattr(self, "dec", None) else []):
                if self.kv_all.wlpT is not None:
                    cross.kv.wlpT = self.kv_all.wlpT[:, i * cross.kv.N:(i + 1) * cross.kv.N]      # (d, 2 H dk) view, row stride L 2 H dk
            self._tr_tiles = torch.tensor(tiles, dtype=torch.int32, device=flat.device)
            self._tr_event = torch.cuda.Event()
            self._tr_pending = False
        # Weight/bias gradients are off the critical path (only the optimizer reads them): they run
        # on a side stream, concurrently with the dgrad chain on the main stream, so the short,
        # latency-bound kernels of both chains fill each other's idle CUs.
        # the CTC branch of the joint model runs on its own stream beside the decoder's forward pass (ctc_branch_async);
        # off in deterministic mode (the weight gradients then share one scratch buffer on whatever stream is current)
        self.side = _shared_stream(flat.device, "wgrad")      # first: the stream the step depends on most gets the first pick
        self.ctc_stream = _shared_stream(flat.device, "aux")
        self.ws_ctc = K.Workspace(flat.device)
        self._side_handle = self.side.cuda_stream
        self._events, self._ev_next = [torch.cuda.Event() for _ in range(64)], 0     # reused round-robin (a wait captures the record it follows)
        # ---- environment switches of the engine (all read here, when the engine is built; the complete list is in README.md):
        # ASR_WGRAD_OVERLAP=0: ONE stream - weight gradients, the CTC branch and the cross-attention K|V work stay on the main stream
        self.overlap_wgrad = os.environ.get("ASR_WGRAD_OVERLAP", "1") == "1"
        # Deterministic mode (kernels.set_deterministic / ASR_DETERMINISTIC=1, read when the engine is built): every
        # gradient reduction runs in a fixed order.  The weight gradients then stay on the main stream (the tied
        # embedding / projection weight is updated by plain read-modify-writes of two kernels, which must not overlap),
        # one launch per projection (the grouped kernel has only the atomic form).
        self.deterministic = K.deterministic()
        if self.deterministic:
            self.overlap_wgrad = False
        # the CTC branch of the joint model beside the decoder's forward pass (ctc_branch_async), and the cross-attention K|V projections /
        # their input gradients, on the auxiliary stream: with the weight-gradient overlap (off in deterministic mode: the weight gradients
        # then share one scratch buffer on whatever stream is current)
        self.overlap_ctc = self.aux_overlap = self.overlap_wgrad
        # ASR_WGRAD_GROUP: "decoder" (default): the seven weight gradients of a DECODER layer (self-attention, cross
        # attention incl. the K|V projection of all encoder frames, feed-forward) go out as ONE grouped GEMM
        # (asr_gemm_tn_grouped_bf16) - the decoder's main stream is a chain of small launches (B*To rows) that
        # leaves the GPU to the side stream: joint config 6.19 -> 6.02 ms; the ENCODER keeps one launch per
        # projection, as soon as its dY exists.  "0": never group; "layer" / "block": group everywhere,
        # per layer / per attention or feed-forward block.  The grouped kernel is 1.6x faster alone (98 vs 160 us
        # per config-2 layer) but one 100-us launch filling every CU overlaps worse with the main stream than
        # four short ones spread over the layer: step 3.84 (layer) / 3.96 (block) vs 3.79 ms, joint 6.27 vs 6.16.
        # "pair" (round 5; measured SLOWER in the step, see below) = "decoder" + the ENCODER's weight gradients as one launch per block: (w_2, w_1) behind the w_2 input
        # gradient, (out-projection, Q|K|V) behind the attention backward.  Not the 256 x 128-tile grouped kernel that "block" measured
        # slower with (its tiles double the bytes every workgroup adds to memory with atomics): the 128 x 128-tile code of the single
        # launches on 64 tiles x 4 M-splits (asr_gemm_tn_grouped_bf16 picks it for problems over the same >= 4096 rows) - two launches'
        # 2 x 16 MB of atomics, kernel boundaries and ring prologues become one launch's.  Stand-alone a layer's four gradients take 119 us as two
        # pairs against 143 us as four launches (132 with the 256 x 128-tile kernel) - and the step is SLOWER: CTC 3.182 vs 3.143 ms, joint 4.912
        # vs 4.894; only the feed-forward pair 3.212 vs 3.136; only the attention pair 3.144 vs 3.133 (A/B in one process, profiles/round5_a_*):
        # the pair starts when the block's SECOND dY exists and then holds every CU for 60 us, the single launches start earlier and their short
        # workgroups give the CUs back to the main stream's kernels sooner.  Opt-in.
        self.group_wgrad = os.environ.get("ASR_WGRAD_GROUP", "decoder")
        self.group_wgrad = None if (self.group_wgrad == "0" or self.deterministic) else self.group_wgrad
        # "pair_ffn" / "pair_attn": only the feed-forward / only the attention block's pair (the other block keeps one launch per projection)
        self._pair_tags = {"pair": ("w2", "w1", "fc", "qkv"), "pair_ffn": ("w2", "w1"), "pair_attn": ("fc", "qkv")}.get(self.group_wgrad, ())
        # The decoder of the joint model is a chain of ~25 small kernels per layer and direction on B*To ~ 550 rows whose workgroups need whole
        # CUs (the small-M GEMM holds 132 - 141 KiB of LDS, attention K / V images 128 KiB); the large launches that run BESIDE the chain on the
        # auxiliary / weight-gradient streams (CTC branch, cross-attention K|V projections and their input gradients, the layer's grouped
        # weight gradients) are one-workgroup-per-CU kernels that hold every CU for 25 - 110 us: a chain kernel launched meanwhile waits
        # for one of them to END (kernel trace, round 4: the 10-us w_1 input gradient took 42 - 50 us beside the grouped weight gradients).
        # ASR_DEC_CU_LIMIT = n > 0 (default 192) sizes those large launches for n CUs (tuning option "cu_limit"), which leaves 256 - n CUs to
        # the chain: joint step 5.01 -> 4.94 ms, and 4.82 with the K|V projections queued ahead of the CTC branch (A/B in one process; 144 .. 224
        # within 1 % of each other; 0 = the whole device).  The head's weight gradient queued BEHIND its input gradient on the auxiliary stream
        # instead of beside it on the weight-gradient stream: 4.880 vs 4.853 ms - not kept.
        self.dec_cu_limit = int(os.environ.get("ASR_DEC_CU_LIMIT", "192"))
        # ASR_KV_GROUPS = g (default 3): the decoder layers' cross-attention K | V projections share one weight matrix, one activation buffer and
        # one gradient buffer (all layers read the same encoder output): forward = layer 0's columns, then ONE GEMM for the rest; backward = the
        # encoder-output gradient d_enc += G W_kv and the K | V weight gradient once per group of L / g adjacent layers (g = L: per layer, as
        # before round 5; g = 1: one K = L 2 H dk GEMM behind the whole decoder backward pass).  Measured (joint step, one box, tools/ab_env.py and
        # the previous commit's tree beside this one): g = 6 / 3 / 2 / 1: 5.01 / 5.01 / 5.04 / 5.09 ms, and 4.89 (previous commit) / 4.89 (g = 6) / 4.89 -
        # 4.90 (g = 3) on another box: these GEMMs run on the auxiliary stream beside the decoder's chain of small kernels, off the critical path -
        # fewer, larger launches there buy nothing, and one launch behind the whole backward pass (g = 1) is exposed.
        g = max(1, min(self.L, int(os.environ.get("ASR_KV_GROUPS", "3"))))
        cuts = [round(j * self.L / g) for j in range(g + 1)]
        self.kv_groups = [(cuts[j], cuts[j + 1]) for j in range(g) if cuts[j + 1] > cuts[j]]
        self._kv_ahead = None
        self._pending = []
        self._deferred = []
        # Operands of kernels issued on the weight-gradient / auxiliary streams are kept alive HERE until the step has joined those
        # streams (encoder_bwd's end), instead of being marked with Tensor.record_stream: a marked block is not reusable when its
        # tensor dies but only once an event recorded at that moment has completed, which the caching allocator polls at later
        # allocations - early steps then find the pool empty and call hipMalloc (a device synchronisation) until the pool has
        # grown past what the step needs (round 3: the extra bench configurations ran 12 - 38 % slower under 5 warm-up steps
        # than under 30).  Holding references costs nothing (the blocks are the step's own) and makes the allocation sequence
        # of every step identical from the second step on.
        self._keep = []
        # ASR_WGRAD_DEFER (default "fc"): projections whose weight gradient is held back until the layer's attention backward is launched (_wgrad)
        self.defer_wgrad = tuple(t for t in os.environ.get("ASR_WGRAD_DEFER", "fc").split(",") if t)
        self.armed_fork = os.environ.get("ASR_ARMED_FORK", "1") == "1"      # hand-overs by the producer kernel's own completion event (_arm)
        self._armed = False
        self._arm_covers_pending = False      # set by _dec_exec_bwd while it collects a layer's weight gradients behind its armed last kernel
        self._ln_part, self._ln_pending = {}, []
        self.dec_exec = os.environ.get("ASR_DEC_EXEC", "1") == "1"      # decoder layers through the native launch sequencer (_dec_exec_ok)
        self._dec_cache = collections.OrderedDict()      # (B, To, T, dropout) -> persistent buffers + plans, least recently used first
        self._block_flush = self.group_wgrad == "block"
        self._in_decoder = False       # "decoder": only the decoder's weight gradients are grouped (one launch per decoder layer)
        # dropout (reference default 0.1; sites: transformer_official.py:175, 306; attention.py:59, 83;
        # module.py:73): masks are regenerated in backward from (step seed, site id), never stored
        self.drop_p = float(getattr(cfg, "dropout", 0.0))
        self.training = True
        self.step_seed = 0

    # ------------------------------------------------------------------ helpers
    def _drop(self, site):
        """(p, seed) of a dropout site for the current step; p = 0 in eval mode."""
        if not self.training or self.drop_p <= 0.0:
            return 0.0, 0
        return self.drop_p, (self.step_seed * 0x9E3779B1 + site * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF

    def _ready(self, name):
        """Gradients at flat offsets >= this tensor's offset are final once the work queued so far
        on BOTH streams has run: the consumer (dist.GradBucketer) waits on events of the two.  Weight gradients held back by
        ASR_WGRAD_DEFER are released first: a mark raised over a gradient that has not been launched yet would let the
        bucketer all-reduce the bucket before that gradient is added (rank-divergent gradients under DataParallel)."""
        if self.grad_ready is not None:
            self._release_deferred()
        self.flush_wgrads()
        if self.grad_ready is not None:
            self.flush_ln_reduce()
            streams = [torch.cuda.current_stream()] + ([self.side] if self.overlap_wgrad else [])
            self.grad_ready(self.flat.index[name][0], streams)

    def _ln_bwd(self, ln, dbias, dy, dy2, xhat, rstd, lens, B, T, **kw):
        """Backward of residual + LayerNorm.  The reduction of the per-workgroup partial parameter gradients is not
        launched per site: the partial sums go to a buffer owned by the site and ONE batched launch reduces all
        pending sites (flush_ln_reduce: per layer when gradients are all-reduced as they become final, otherwise once
        at the end of backward) - each of the 13 small launches cost ~5 us of dispatch and drain."""
        d = xhat.shape[-1]
        need = K.add_ln_bwd_workspace_bytes(B * T, d)
        part = self._ln_part.get(id(ln))
        if part is None or part.numel() < need:
            part = self._ln_part[id(ln)] = torch.empty(need, dtype=torch.uint8, device=xhat.device)
        out = K.add_ln_bwd(dy, dy2, xhat, rstd, ln.g, lens, ln.gg, ln.gb, dbias, B, T, self.ws, partials=part, **kw)
        self._ln_pending.append((part, ln.gg, ln.gb, dbias, B * T))
        return out

    def flush_ln_reduce(self):
        if self._ln_pending:
            items, self._ln_pending = self._ln_pending, []
            K.add_ln_bwd_reduce_batched(items, self.d)

    def tail_mark_name(self):
        """First tensor of the block that encoder_bwd marks final in the middle of layer 0 (see dist.DataParallel)."""
        return "encoder.layer_stack.0.pos_ffn.w_1.weight"

    def _event(self):
        if torch.cuda.is_current_stream_capturing():      # events recorded in a capture belong to that graph
            return torch.cuda.Event()
        ev = self._events[self._ev_next]
        self._ev_next = (self._ev_next + 1) & 63
        return ev

    def _arm(self):
        """The NEXT library entry point called on the current stream hands its output over to the weight-gradient stream by itself: its
        last kernel carries the completion event (asr_stream_arm), and the next _fork(self.side) is then a no-op.  Without it the
        hand-over is an event record behind the producer - a barrier packet that delays the next main-stream kernel by ~3.5 us, ~26
        times per step.  Only for producers that are ONE armed-capable call (see include/asr_hip.h); anything else falls back."""
        if self.armed_fork and self.overlap_wgrad and not torch.cuda.is_current_stream_capturing() and not self._in_decoder:
            K.stream_arm(self._side_handle)
            self._armed = True

    def _disarm(self):
        """Drop an arm that no _fork(self.side) consumed right behind its producer.  If the producer's launch took it, the side stream
        merely waits for that kernel (harmless); what must not happen is that a LATER _fork finds `_armed` set and skips its event
        although other main-stream kernels ran since (round-3 ADVICE: ASR_WGRAD_DEFER=w2 at B*T < 4096, ASR_WGRAD_GROUP=block with
        ASR_FUSE_RELU_BWD=0 - the side stream then read a dY ordered only behind the armed kernel)."""
        if self._armed:
            self._armed = False
            K.stream_arm_pending()

    def _fork(self, stream):
        """`stream` continues after everything queued on the current stream so far."""
        if self._armed:
            self._armed = False
            taken = not K.stream_arm_pending()      # clears the arm either way
            if taken and stream is self.side:
                return                               # the producer's own completion event is already queued on the side stream
        if torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            stream.wait_event(ev)
        else:
            K.stream_fork(stream.cuda_stream)

    def _release_kept(self, joined=False):
        """Drop the references of self._keep.  joined: the caller has just made the current stream wait for the side streams (the end
        of a training step).  Otherwise (an evaluation forward pass, a step that raised) the current stream first waits for both, so
        that whatever reuses the blocks is ordered behind their last reader."""
        if self._keep:
            if not joined and not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream()
                cur.wait_stream(self.side)
                cur.wait_stream(self.ctc_stream)
            self._keep.clear()

    def refresh_transposes(self, zero=None):
        """W^T copies for the own-kernel input gradients: one launch on the side stream, which is idle during the
        forward pass; the backward pass waits for it (wait_transposes) before its first input-gradient GEMM.
        Called by the model at the start of every step that will run a backward pass (train_step), whatever the
        module's train / eval flag says."""
        if self._tr_tiles is None:
            if zero is not None:
                zero.zero_()
            return
        self._disarm()      # an arm left by a step that raised must not skip this hand-over (the copies must follow the optimizer update)
        self._fork(self.side)
        if zero is not None:      # the step's gradient buffer, zeroed beside the forward pass (Models.zero_flat_grads); ordered by _tr_event below
            with torch.cuda.stream(self.side):
                zero.zero_()
        K.STREAM_OVERRIDE = self._side_handle
        try:
            K.transpose_batched(self.flat.lp, self.flat.lpT, self._tr_tiles)
        finally:
            K.STREAM_OVERRIDE = None
        self._tr_event.record(self.side)
        self._tr_pending = True
        self.flat.lpT_version = self.flat.version

    def wait_transposes(self):
        """The current stream waits for this step's transposed weight copies (every stream that multiplies by them calls it: the main
        stream in front of the backward pass, the auxiliary stream in front of the CTC head's input gradient)."""
        if self._tr_tiles is not None and self._tr_pending:
            torch.cuda.current_stream().wait_event(self._tr_event)

    def join_side(self):
        """Main stream waits for every weight-gradient kernel issued so far."""
        self._release_deferred()
        self.flush_wgrads()
        if self.overlap_wgrad:
            torch.cuda.current_stream().wait_stream(self.side)

    def flush_wgrads(self):
        """Launch the weight gradients collected since the last flush as ONE grouped GEMM
        (asr_gemm_tn_grouped_bf16) - on the side stream when the overlap is on."""
        if not self._pending:
            return
        probs, self._pending = self._pending, []
        if not self.overlap_wgrad:
            K.gemm_tn_grouped(probs, accumulate=True)
            return
        self._fork(self.side)
        K.STREAM_OVERRIDE = self._side_handle
        lim = self.dec_cu_limit if self._in_decoder else 0
        try:
            if lim:
                K.set_cu_limit(lim)
            K.gemm_tn_grouped(probs, accumulate=True)
        finally:
            K.STREAM_OVERRIDE = None
            if lim:
                K.set_cu_limit(0)
        # also while capturing: a graph's private pool DOES reuse a block freed earlier in the same capture, and without the mark a
        # replay overwrote dY / X tensors that the side stream's weight-gradient kernel had not read yet (round 3: gradients of the
        # replayed step differed from the eager step; tests/test_train_loop_gpu.py::test_graphed_step_matches_eager)
        for dy, x, _, _ in probs:
            self._keep += (dy, x)

    def _release_deferred(self):
        """Launch the weight gradients held back by ASR_WGRAD_DEFER (see _wgrad)."""
        if self._deferred:
            items, self._deferred = self._deferred, []
            hold, self.defer_wgrad = self.defer_wgrad, ""
            try:
                for lin, dy, x, bias_from in items:
                    self._wgrad(lin, dy, x, bias_from)
            finally:
                self.defer_wgrad = hold

    def _wgrad(self, lin, dy, x, bias_from=None):
        """lin.gw += dy^T x (and lin.gb += colsum(bias_from)) on the side stream."""
        pair = self._pair_tags and not self._in_decoder and lin.tag in self._pair_tags
        if self.defer_wgrad and self.overlap_wgrad and not self._in_decoder and lin.tag in self.defer_wgrad and not pair:
            # ASR_WGRAD_DEFER (default "fc"): hold this projection's weight gradient until the layer's attention backward is
            # about to be launched.  For the out-projection that takes it from beside its own (short) dgrad GEMM to beside the
            # 83-us attention kernel: step 3.53 -> 3.50 ms.  Holding back any other projection (w1, w2, qkv) or releasing after
            # the attention launch is slower (3.56 .. 3.73 ms, tools/env_sweep.sh): the side stream has no slack to give.
            self._deferred.append((lin, dy, x, bias_from))
            self._disarm()      # no fork here: the arm must not outlive its producer (a later _fork would skip its event)
            return
        fused = bias_from is dy and lin.fused_bias_wgrad(dy, x)     # bias gradient inside the weight-gradient GEMM
        grouped_here = self.group_wgrad and (self._in_decoder or pair or self.group_wgrad in ("layer", "block"))      # ("decoder", "pair*": the decoder's layers)
        if grouped_here and (bias_from is None or fused) and dy.dtype == torch.bfloat16 and lin.N % 8 == 0 and lin.K % 8 == 0 \
                and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0:
            self._pending.append((dy, x, lin.gw, lin.gb if fused else None))     # launched by flush_wgrads (end of the layer)
            if pair and lin.tag in ("w1", "qkv"):
                # the block's second weight gradient: its dY was written by the kernel launched last (the w_2 input gradient, the attention
                # backward) - that kernel's arm, if it took one, is this launch's hand-over
                self.flush_wgrads()
            elif not self._arm_covers_pending:      # only the decoder sequencer's arm (its layer's LAST kernel) orders everything collected here
                self._disarm()
            return
        if not self.overlap_wgrad:
            if bias_from is not None and not fused:
                lin.bgrad(bias_from, self.ws_side)
            lin.wgrad(dy, x, with_bias=fused, ws=self.ws_side)
            return
        self._fork(self.side)
        own = dy.dtype == torch.bfloat16 and lin.N % 8 == 0 and lin.K % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
        if own:     # only own kernels run: hand them the side stream directly instead of switching torch's current stream
            K.STREAM_OVERRIDE = self._side_handle
            try:
                if bias_from is not None and not fused:
                    lin.bgrad(bias_from, self.ws_side)
                lin.wgrad(dy, x, with_bias=fused, ws=self.ws_side)
            finally:
                K.STREAM_OVERRIDE = None
        else:
            with torch.cuda.stream(self.side):
                if bias_from is not None and not fused:
                    lin.bgrad(bias_from, self.ws_side)
                lin.wgrad(dy, x, with_bias=fused)
        self._keep += (dy, x)      # also while capturing (see flush_wgrads)

    def _attn_block_fwd(self, m, x, kv_src, B, Tq, Tk, k_len, q_lens, causal, window, cross, site, kv_pre=None):
        """x: (B*Tq, d) queries + residual; kv_src: (B*Tk, d).  Returns output and cache."""
        H, dk, hd = self.H, self.dk, self.H * self.dk
        c = {}
        if not cross:
            qkv = m.qkv.fwd(x)
            q, k, v = qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:]
            c["qkv"] = qkv
        else:
            q = m.q.fwd(x)
            if kv_pre is not None:      # projected ahead of time on the auxiliary stream (decoder_fwd)
                kv, ev = kv_pre
                torch.cuda.current_stream().wait_event(ev)
            else:
                kv = m.kv.fwd(kv_src)
            k, v = kv[:, :hd], kv[:, hd:]
            c["q"], c["kv"] = q, kv
        pa, sa = self._drop(site)          # attention probabilities (attention.py:83)
        pf, sf = self._drop(site + 1)      # after fc, before residual + LN (attention.py:59)
        # Heads of more than 512 keys (the long-form configuration): the forward pass of a training step (grad mode) also stores the low-order piece of
        # its bf16 output for the backward pass's delta (asr_hip.h: asr_sdpa_fwd's o_lo) - the band form of the backward kernel and the two-kernel path
        # hold no head-wide mean for the remedies of the one-workgroup form (centred keys, dK's mean removed: sdpa.hip)
        ctx_lo = torch.empty(B * Tq, hd, dtype=q.dtype, device=q.device) if (Tk > 512 and q.dtype == torch.bfloat16 and torch.is_grad_enabled()) else None
        ctx, lse = K.sdpa_fwd(q, k, v, k_len, B, H, Tq, Tk, dk, causal, window, drop_p=pa, drop_seed=sa, o_lo=ctx_lo)
        a = m.fc.fwd(ctx)
        y, xhat, rstd = K.add_ln_fwd(a, x, m.ln.g, m.ln.b, None, q_lens, B, Tq, xhat=a, drop_p=pf, drop_seed=sf, drop_mode=1)
        c.update(x=x, kv_src=kv_src, ctx=ctx, ctx_lo=ctx_lo, lse=lse, xhat=xhat, rstd=rstd, k_len=k_len, q_lens=q_lens, dims=(B, Tq, Tk), causal=causal,
                 window=window, cross=cross, drop=(pa, sa, pf, sf))
        return y, c

    def _attn_block_bwd(self, m, c, dy, dy2, d_kv_src=None):
        """Returns (dx_proj, dz): gradient wrt the block input through the projections, and the
        residual-path gradient dz (to be added by the consumer).  For cross attention the
        key/value-source gradient is accumulated into d_kv_src in place."""
        H, dk, hd = self.H, self.dk, self.H * self.dk
        B, Tq, Tk = c["dims"]
        pa, sa, pf, sf = c["drop"]
        fc_deferred = bool(self.defer_wgrad) and self.overlap_wgrad and not self._in_decoder and m.fc.tag in self.defer_wgrad and "fc" not in self._pair_tags
        if not fc_deferred:
            self._arm()
        dz, dxg = self._ln_bwd(m.ln, m.fc.gb, dy, dy2, c["xhat"], c["rstd"], c["q_lens"], B, Tq, drop_p=pf, drop_seed=sf, drop_mode=1)
        self._wgrad(m.fc, dxg, c["ctx"])
        if fc_deferred and not c["cross"] and m.fc.own_dgrad(dxg):
            self._arm()      # the held-back weight gradient is released right behind this input-gradient GEMM: its completion is the hand-over
        dctx = m.fc.dgrad(dxg)
        if not c["cross"]:
            qkv = c["qkv"]
            dqkv = torch.empty_like(qkv)
            self._release_deferred()      # before the attention launch (after it: 3.59 vs 3.50 ms)
            self._disarm()      # nothing was released: the arm must not leak into the attention kernel's launch
            self._arm()
            K.sdpa_bwd(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], c["ctx"], dctx, c["lse"], c["k_len"], B, H, Tq, Tk, dk,
                       dqkv[:, :hd], dqkv[:, hd:2 * hd], dqkv[:, 2 * hd:], c["causal"], c["window"], drop_p=pa, drop_seed=sa, o_lo=c["ctx_lo"])
            self._wgrad(m.qkv, dqkv, c["x"], bias_from=dqkv)
            dx = m.qkv.dgrad(dqkv)
        else:
            q, kv = c["q"], c["kv"]
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            K.sdpa_bwd(q, kv[:, :hd], kv[:, hd:], c["ctx"], dctx, c["lse"], c["k_len"], B, H, Tq, Tk, dk, dq, dkv[:, :hd], dkv[:, hd:],
                       c["causal"], c["window"], drop_p=pa, drop_seed=sa, o_lo=c["ctx_lo"])
            self._wgrad(m.q, dq, c["x"], bias_from=dq)
            self._wgrad(m.kv, dkv, c["kv_src"], bias_from=dkv)
            dx = m.q.dgrad(dq)
            if self.aux_overlap and not torch.cuda.is_current_stream_capturing():
                # d_enc += dK|dV W_kv (a 16000-row GEMM) is off the decoder's dependent chain: only the encoder's backward pass
                # needs d_enc.  It runs on the auxiliary stream (in order behind the CTC branch and the previous layers' adds);
                # decoder_bwd joins that stream at its end.
                self._fork(self.ctc_stream)
                with torch.cuda.stream(self.ctc_stream):
                    m.kv.dgrad(dkv, out=d_kv_src, accumulate=True)
                self._keep.append(dkv)
            else:
                m.kv.dgrad(dkv, out=d_kv_src, accumulate=True)
        if self._block_flush:
            self.flush_wgrads()
        return dx, dz

    def _ffn_block_fwd(self, f, x, B, T, lens, site):
        h = f.w1.fwd(x, act=ACT_RELU)
        pf, sf = self._drop(site)          # after w_2, before residual + LN (module.py:73)
        o = f.w2.fwd(h)
        y, xhat, rstd = K.add_ln_fwd(o, x, f.ln.g, f.ln.b, None, lens, B, T, xhat=o, drop_p=pf, drop_seed=sf, drop_mode=1)
        return y, dict(x=x, h=h, xhat=xhat, rstd=rstd, lens=lens, dims=(B, T), drop=(pf, sf))

    def _ffn_block_bwd(self, f, c, dy, dy2):
        B, T = c["dims"]
        pf, sf = c["drop"]
        self._arm()
        dz, dxg = self._ln_bwd(f.ln, f.w2.gb, dy, dy2, c["xhat"], c["rstd"], c["lens"], B, T, drop_p=pf, drop_seed=sf, drop_mode=1)
        self._wgrad(f.w2, dxg, c["h"])
        # ReLU backward in the store tail of the input-gradient GEMM (every kernel that takes the shape has that tail; else a masking pass)
        fused_relu = (f.w2.own_dgrad(dxg) or f.w2.small(dxg, f.w2.N) or dxg.dtype == torch.float32) and c["h"].is_contiguous() and c["h"].data_ptr() % 16 == 0
        if fused_relu and f.w2.own_dgrad(dxg):      # dh then comes out of ONE launch of the NT kernel
            self._arm()
        dh = f.w2.dgrad(dxg, relu_mask=c["h"] if fused_relu else None)
        fused = f.w1.fused_bias_wgrad(dh, c["x"])      # then the w_1 bias gradient comes out of its weight-gradient GEMM
        if not fused_relu:
            self._disarm()      # dh is finished by the masking kernel below, not by the (possibly armed) GEMM above
            K.relu_bwd_(dh, c["h"], None if fused else f.w1.gb, self.ws)
        elif not fused:
            f.w1.bgrad(dh, self.ws)
        self._wgrad(f.w1, dh, c["x"], bias_from=dh if fused else None)
        dx = f.w1.dgrad(dh)
        if self._block_flush:
            self.flush_wgrads()
        return dx, dz

    # ------------------------------------------------------------------ encoder
    def encoder_fwd(self, wave, wave_len, window=-1):
        """wave (B,T,F) in the compute dtype, wave_len (B) int32.  transformer_official.py:158-189."""
        B, T, F = wave.shape
        self._deferred.clear()
        self._pending.clear()        # work queued by a step that did not finish (an exception between backward and
        self._ln_pending.clear()     # its flush) must not be launched into this step's gradients
        self._disarm()               # ... nor may its arm skip this step's first hand-over (refresh_transposes clears it too)
        self._release_kept()
        x_in = wave.reshape(B * T, F)
        e0 = self.lin_in.fwd(x_in)
        p0, s0 = self._drop(1)             # dropout(LN(linear_in(x)) + PE)  (transformer_official.py:175-177)
        h, xhat, rstd = K.add_ln_fwd(e0, None, self.ln_in.g, self.ln_in.b, self.pe, None, B, T, xhat=e0, drop_p=p0, drop_seed=s0, drop_mode=2)
        cache = dict(x_in=x_in, xhat_in=xhat, rstd_in=rstd, B=B, T=T, layers=[], drop=(p0, s0))
        for i, (mha, ffn) in enumerate(self.enc):
            h1, c1 = self._attn_block_fwd(mha, h, h, B, T, T, wave_len, wave_len, False, window, False, site=10 + 4 * i)
            h, c2 = self._ffn_block_fwd(ffn, h1, B, T, wave_len, site=12 + 4 * i)
            cache["layers"].append((c1, c2))
        return h, cache

    def encoder_bwd(self, cache, d_enc):
        B, T = cache["B"], cache["T"]
        dy, dy2 = d_enc, None
        self.wait_transposes()
        for i in reversed(range(self.L)):
            mha, ffn = self.enc[i]
            c1, c2 = cache["layers"][i]
            dx, dz = self._ffn_block_bwd(ffn, c2, dy, dy2)
            if i == 0:      # finer mark inside the last layer: its feed-forward gradients can leave while attention runs
                self._ready(self.tail_mark_name())
            dx, dz = self._attn_block_bwd(mha, c1, dx, dz)
            dy, dy2 = dx, dz
            self._ready(f"encoder.layer_stack.{i}.slf_attn.w_qs.weight")
        p0, s0 = cache["drop"]
        dz, _ = self._ln_bwd(self.ln_in, self.lin_in.gb, dy, dy2, cache["xhat_in"], cache["rstd_in"], None, B, T, drop_p=p0, drop_seed=s0, drop_mode=2)
        # the last weight gradient runs on the MAIN stream: the side stream is still busy with layer 0,
        # and a cross-stream hand-over costs ~20 us of latency that nothing would hide at this point
        self.lin_in.wgrad(dz, cache["x_in"], ws=self.ws)      # A/B: 8.50 vs 8.45 k utt/s
        self._block_flush = self.group_wgrad == "block"
        self.flush_ln_reduce()
        self.join_side()
        # every stream has been joined (the auxiliary stream at the end of decoder_bwd; without a decoder nothing runs on it): the
        # operands held for them can go back to the pool - what reuses them is ordered behind this point on the current stream
        self._release_kept(joined=True)
        self._ready("encoder.linear_in.weight")

    # ------------------------------------------------------------------ CTC head
    def ctc_branch_async(self, enc, wave_len, labels32, lab_len, B, T, grad_scale, grad_scale_div=None):
        """The whole CTC branch (head projection, forward-backward, its input gradient and weight gradient) on its own stream,
        beside the decoder's forward pass: the decoder is a chain of ~70 launches on B*To ~ 550 rows that leaves most of the
        GPU idle, the CTC branch is four large kernels (~0.3 ms at config 3) that do not depend on it.  Returns (nll, d_enc,
        event): consumers on the main stream wait for the event (decoder_bwd does, before its first write into d_enc)."""
        main = torch.cuda.current_stream()
        self._fork(self.ctc_stream)
        with torch.cuda.stream(self.ctc_stream):
            if self.dec_cu_limit:      # beside the decoder's forward chain: leave it some CUs (see self.dec_cu_limit)
                K.set_cu_limit(self.dec_cu_limit)
            try:
                nll, d_enc = self.ctc_fwd_bwd(enc, wave_len, labels32, lab_len, B, T, grad_scale, grad_scale_div=grad_scale_div, ws=self.ws_ctc)
            finally:
                if self.dec_cu_limit:
                    K.set_cu_limit(0)
            done = torch.cuda.Event()
            done.record(self.ctc_stream)
        # inputs read on the auxiliary stream, and the two results (blocks of that stream's pool, read on the main stream): alive until
        # the step has joined the stream (decoder_bwd), see self._keep
        self._keep += (enc, wave_len, labels32, lab_len, nll, d_enc)
        return nll, d_enc, done

    def ctc_fwd_bwd(self, enc, wave_len, labels32, lab_len, B, T, grad_scale, want_grad=True, grad_scale_div=None, ws=None, best_path=None):
        """Returns (nll (B,), d_enc contribution or None).  best_path: optional (B, T) int32 tensor for the frame-wise argmax of the
        logits (the greedy path of the step's CER; the gradient overwrites the logits in place, so it is taken inside the loss kernels)."""
        buf = torch.empty(B * T, self.ld_v, dtype=enc.dtype, device=enc.device)      # rows padded to whole lines (see self.ld_v)
        ws = ws if ws is not None else self.ws
        logits = self.ctc_lo.fwd(enc, out=buf[:, :self.V])
        frames = buf.view(B, T, self.ld_v)[:, :, :self.V]
        if want_grad:
            self._arm()      # the head's weight gradient follows the loss kernels directly
        nll, dl = K.ctc_fwd_bwd(frames, wave_len, labels32, lab_len, ws, blank=0, grad_scale=grad_scale,
                                dlogits=frames if want_grad else None, want_grad=want_grad, grad_scale_div=grad_scale_div, best_path=best_path)
        if not want_grad:
            return nll, None
        dl = logits      # the gradient was written in place
        self._wgrad(self.ctc_lo, dl, enc, bias_from=dl)
        self.wait_transposes()      # this step's transposed copy of the head (made on the weight-gradient stream at the start of the step)
        d_enc = self.ctc_lo.dgrad(dl)
        # "ready(o)" means every gradient at flat offsets >= o is final.  ctc_lo sits BELOW the decoder
        # block in the flat buffer and the joint model runs the CTC backward BEFORE the decoder
        # backward, so with a decoder the mark is raised at the end of decoder_bwd instead (raising it
        # here all-reduced the still-empty decoder gradients: wrong with two or more ranks)
        if not self.use_decoder:
            self._ready("ctc_lo.weight")
        return nll, d_enc

    # ------------------------------------------------------------------ decoder layers through the native launch sequencer
    DEC_CACHE_SHAPES = 6

    def _dec_exec_ok(self, B, To, T):
        """The C++ sequencer (csrc/decoder_exec.hip: asr_decoder_layer_fwd / _bwd) takes the decoder layers when every projection
        of the layer runs on the small-M kernel: bf16, B*To rows within its limit, widths multiples of 8.  ASR_DEC_EXEC=0 keeps
        the per-kernel Python path (also used while bench.py times individual kernels)."""
        hd = self.H * self.dk
        return (self.dec_exec and self.dtype == torch.bfloat16 and K.TIMER is None and B * To <= Linear.SMALL_M and self.d % 8 == 0 and hd % 8 == 0
                and self.ff % 8 == 0 and max(3 * hd, self.ff, self.d) <= Linear.SMALL_REDUCE)

    def _dec_bufs(self, B, To, T, drop):
        """Persistent activation / gradient buffers and launch plans of the decoder layers for one batch shape (the layer sequence
        is fixed, so nothing is allocated per step; steps are sequential and the optimizer joins every side stream before the
        next forward pass touches these buffers)."""
        key = (B, To, T, bool(drop))
        hit = self._dec_cache.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if hit is not None:
            self._dec_cache.move_to_end(key)
            # a captured hipGraph bakes these buffers' addresses into its kernel arguments: the shape stays for the life of the engine
            hit["pinned"] = hit["pinned"] or capturing
            return hit
        # real batches come in many shapes: keep the most recent few, free the rest - never one a graph replays into
        evictable = [k for k, v in self._dec_cache.items() if not v["pinned"]]
        while len(evictable) >= self.DEC_CACHE_SHAPES:
            del self._dec_cache[evictable.pop(0)]
        dev, M, d, hd, ff, H = self.flat.device, B * To, self.d, self.H * self.dk, self.ff, self.H
        bf = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        part_bytes = K.add_ln_bwd_workspace_bytes(M, d)
        need = max(_lib.lib.asr_sdpa_bwd_workspace_bytes(B, H, To, T, self.dk, 0, -1, _lib.ASR_BF16), _lib.lib.asr_sdpa_bwd_workspace_bytes(B, H, To, To, self.dk, 1, -1, _lib.ASR_BF16))
        delta = f32((need + 3) // 4)
        layers = []
        # K | V of the encoder frames for ALL layers in one buffer (layer i = columns [i 2hd, (i+1) 2hd): the attention kernels take a row
        # stride), and the gradient wrt it likewise: the six projections are one GEMM forward (two launches: layer 0's columns first), the
        # encoder-output gradient d_enc += G W_kv one GEMM per layer GROUP with the reduction over the group's columns, and the weight
        # gradient one problem per group (self.kv_groups)
        kv_all, g_kv_all = bf(B * T, self.L * 2 * hd), bf(B * T, self.L * 2 * hd)
        for i, (slf, cross, ffn) in enumerate(self.dec):
            t = dict(qkv_s=bf(M, 3 * hd), ctx_s=bf(M, hd), a_s=bf(M, d), y_s=bf(M, d), lse_s=f32(B, H, To), rstd_s=f32(M),
                     q_c=bf(M, hd), kv_c=kv_all[:, i * 2 * hd:(i + 1) * 2 * hd], ctx_c=bf(M, hd), a_c=bf(M, d), y_c=bf(M, d), lse_c=f32(B, H, To), rstd_c=f32(M),
                     h=bf(M, ff), o=bf(M, d), y_f=bf(M, d), rstd_f=f32(M),
                     dz_f=bf(M, d), g_h=bf(M, ff), dx_f=bf(M, d), dz_c=bf(M, d), g_qc=bf(M, hd), g_kvc=g_kv_all[:, i * 2 * hd:(i + 1) * 2 * hd], dx_c=bf(M, d),
                     dz_s=bf(M, d), g_qkv=bf(M, 3 * hd), dx_s=bf(M, d), dctx=bf(M, hd),
                     part_f=torch.empty(part_bytes, dtype=torch.uint8, device=dev), part_c=torch.empty(part_bytes, dtype=torch.uint8, device=dev),
                     part_s=torch.empty(part_bytes, dtype=torch.uint8, device=dev))
            if drop:      # pre-residual dropout: the gradient wrt a projection's output is its own tensor
                t.update(g_o=bf(M, d), g_ac=bf(M, d), g_as=bf(M, d))
            pl = _lib.DecLayerPlan()
            pl.B, pl.To, pl.T, pl.d, pl.H, pl.dk, pl.ff = B, To, T, d, H, self.dk, ff
            pl.ld_kv_c = self.L * 2 * hd
            for name, ten in t.items():
                setattr(pl, name, ten.data_ptr())
            for name, lin in (("qkv_s", slf.qkv), ("fc_s", slf.fc), ("q_c", cross.q), ("fc_c", cross.fc), ("1", ffn.w1), ("2", ffn.w2)):
                setattr(pl, "w_" + name, lin.wlp.data_ptr())
                setattr(pl, "b_" + name, lin.b32.data_ptr())
            for name, ln in (("s", slf.ln), ("c", cross.ln), ("f", ffn.ln)):
                setattr(pl, "g_" + name, ln.g.data_ptr())
                setattr(pl, "be_" + name, ln.b.data_ptr())
            pl.gb_2, pl.gb_fc_c, pl.gb_fc_s = ffn.w2.gb.data_ptr(), cross.fc.gb.data_ptr(), slf.fc.gb.data_ptr()
            if cross.kv.wlpT is not None:
                pl.w_kv_c_T, pl.ld_kv_c_T = cross.kv.wlpT.data_ptr(), cross.kv.wlpT.stride(0)
            pl.delta, pl.delta_bytes = delta.data_ptr(), delta.numel() * 4
            t["kv_event"] = torch.cuda.Event()
            layers.append((pl, t))
        hit = self._dec_cache[key] = dict(layers=layers, delta=delta, pinned=capturing, kv_all=kv_all, g_kv_all=g_kv_all, kv_event_rest=torch.cuda.Event())
        return hit

    def _kv_exec_async(self, bufs, enc):
        """The six cross-attention K|V projections of the encoder output (16000-row GEMMs, ~24 us each) on the auxiliary stream, one event
        each: they do not depend on the decoder state, and the main stream works through the decoder's small kernels meanwhile."""
        self._fork(self.ctc_stream)
        with torch.cuda.stream(self.ctc_stream):
            if self.dec_cu_limit:
                K.set_cu_limit(self.dec_cu_limit)
            try:
                # layer 0's columns first (the chain's first cross-attention, ~80 us in, waits for them), then layers 1 .. L-1 as ONE GEMM
                n0 = self.dec[0][1].kv.N
                pl0, t0 = bufs["layers"][0]
                self.dec[0][1].kv.fwd(enc, out=t0["kv_c"])
                t0["kv_event"].record(self.ctc_stream)
                pl0.kv_ready_event = t0["kv_event"].cuda_event
                if self.L > 1:
                    self.kv_all.rows(n0, self.kv_all.N).fwd(enc, out=bufs["kv_all"][:, n0:])
                    bufs["kv_event_rest"].record(self.ctc_stream)
                    for pl, t in bufs["layers"][1:]:
                        pl.kv_ready_event = bufs["kv_event_rest"].cuda_event
            finally:
                if self.dec_cu_limit:
                    K.set_cu_limit(0)
        self._keep.append(enc)

    def decoder_kv_async(self, prep, enc, B, T):
        """Called by the training step BEFORE ctc_branch_async: queues the K|V projections on the auxiliary stream AHEAD of the CTC branch.
        Issued from decoder_fwd they sat behind the whole branch (head GEMM, loss kernels, its input gradient: ~0.3 ms), and the
        decoder's first cross-attention - ~80 us into the chain - waited for layer 0's projection at the end of that queue (kernel
        trace, round 4: the chain stood still for ~240 us).  The CTC results are needed only when the decoder's backward pass starts."""
        self._kv_ahead = None
        To = prep[0].shape[1]
        if not (self.aux_overlap and not torch.cuda.is_current_stream_capturing() and self._dec_exec_ok(B, To, T)):
            return
        bufs = self._dec_bufs(B, To, T, self.training and self.drop_p > 0.0)
        self._kv_exec_async(bufs, enc)
        self._kv_ahead = (bufs, enc)

    def _dec_exec_fwd(self, x, enc, dec_len, cross_len, B, To, T):
        drop = self.training and self.drop_p > 0.0
        bufs = self._dec_bufs(B, To, T, drop)
        main = K._stream()
        overlap = self.aux_overlap and not torch.cuda.is_current_stream_capturing()
        ahead, self._kv_ahead = self._kv_ahead, None
        if overlap and not (ahead is not None and ahead[0] is bufs and ahead[1] is enc):      # not issued ahead of the CTC branch (decoder_kv_async)
            self._kv_exec_async(bufs, enc)
        for i, ((pl, t), (_, cross, _)) in enumerate(zip(bufs["layers"], self.dec)):
            if not overlap:
                cross.kv.fwd(enc, out=t["kv_c"])
                pl.kv_ready_event = None
            pl.x_in = x.data_ptr()
            pl.dec_len, pl.cross_len = dec_len.data_ptr(), cross_len.data_ptr()
            pl.drop_p = self.drop_p if drop else 0.0
            for j, site in enumerate((100 + 8 * i, 101 + 8 * i, 102 + 8 * i, 103 + 8 * i, 104 + 8 * i)):
                pl.seed[j] = self._drop(site)[1]
            _lib.check(_lib.fast.asr_decoder_layer_fwd(ctypes.addressof(pl), main), "asr_decoder_layer_fwd")
            t["x_in"] = x
            x = t["y_f"]
        return x, bufs

    def _dec_exec_bwd(self, bufs, dy, enc, d_enc, d_enc_ready):
        drop = self.training and self.drop_p > 0.0
        main = K._stream()
        aux = self.ctc_stream.cuda_stream if (self.aux_overlap and not torch.cuda.is_current_stream_capturing()) else None
        dy2 = None
        M = dy.shape[0]
        nkv = self.dec[0][1].kv.N
        groups = {lo: (lo, hi) for lo, hi in self.kv_groups}      # keyed by the layer that closes the group
        for i in reversed(range(self.L)):
            pl, t = bufs["layers"][i]
            slf, cross, ffn = self.dec[i]
            if cross.kv.wlpT is not None:
                cross.kv._fresh_transpose()
            if d_enc_ready is not None:      # d_enc must hold the CTC branch's contribution before the first cross-attention add
                torch.cuda.current_stream().wait_event(d_enc_ready)
                d_enc_ready = None
            # the encoder-output gradient d_enc += G W_kv runs once per GROUP of layers (self.kv_groups), issued by the group's last layer
            # (the lowest index: the backward pass walks downwards) over the group's columns of the shared gradient buffer
            grp = groups.get(i) if cross.kv.wlpT is not None else None
            pl.d_enc, pl.kv_dgrad_cols, pl.g_kv_group = None, 0, None
            if grp is not None:
                c0, c1 = grp[0] * nkv, grp[1] * nkv
                pl.d_enc, pl.kv_dgrad_cols = d_enc.data_ptr(), c1 - c0
                pl.g_kv_group = bufs["g_kv_all"][:, c0:].data_ptr()
                pl.w_kv_c_T, pl.ld_kv_c_T = self.kv_all.wlpT[:, c0:].data_ptr(), self.kv_all.wlpT.stride(0)
            hand_over = self.armed_fork and self.overlap_wgrad and not torch.cuda.is_current_stream_capturing()
            pl.wgrad_stream = self._side_handle if hand_over else None      # the layer's last kernel signals the weight-gradient stream itself
            pl.aux_cus = self.dec_cu_limit if aux is not None else 0
            _lib.check(_lib.fast.asr_decoder_layer_bwd(ctypes.addressof(pl), dy.data_ptr(), None if dy2 is None else dy2.data_ptr(), main, aux),
                       "asr_decoder_layer_bwd")
            self._armed = self._arm_covers_pending = hand_over      # consumed (or found pending) by the next _fork(self.side): flush_wgrads in _ready
            if cross.kv.wlpT is None:      # no transposed copy (odd widths): the accumulating input gradient through the generic path
                cross.kv.dgrad(t["g_kvc"], out=d_enc, accumulate=True)
            # weight gradients of the layer (one grouped launch on the side stream) and the LayerNorm parameter-gradient partial sums
            self._wgrad(ffn.w2, t["g_o"] if drop else t["dz_f"], t["h"])
            self._wgrad(ffn.w1, t["g_h"], t["y_c"], bias_from=t["g_h"])
            self._wgrad(cross.fc, t["g_ac"] if drop else t["dz_c"], t["ctx_c"])
            self._wgrad(cross.q, t["g_qc"], t["y_s"], bias_from=t["g_qc"])
            if cross.kv.wlpT is None:
                self._wgrad(cross.kv, t["g_kvc"], enc, bias_from=t["g_kvc"])
            elif grp is not None:      # the K | V weight gradients of the group's layers as ONE problem (their rows of the shared matrix are adjacent)
                c0, c1 = grp[0] * nkv, grp[1] * nkv
                g = bufs["g_kv_all"][:, c0:c1]
                self._wgrad(self.kv_all.rows(c0, c1), g, enc, bias_from=g)
            self._wgrad(slf.fc, t["g_as"] if drop else t["dz_s"], t["ctx_s"])
            self._wgrad(slf.qkv, t["g_qkv"], t["x_in"], bias_from=t["g_qkv"])
            self._ln_pending += [(t["part_f"], ffn.ln.gg, ffn.ln.gb, ffn.w2.gb, M), (t["part_c"], cross.ln.gg, cross.ln.gb, cross.fc.gb, M),
                                 (t["part_s"], slf.ln.gg, slf.ln.gb, slf.fc.gb, M)]
            dy, dy2 = t["dx_s"], t["dz_s"]
            self._ready(f"decoder.layer_stack.{i}.slf_attn.w_qs.weight")
            self._arm_covers_pending = False
        return dy, dy2

    # ------------------------------------------------------------------ decoder
    def decoder_fwd(self, prep, enc, cross_len, B, T):
        """prep = kernels.dec_preprocess(tgt).  transformer_official.py:277-328."""
        ys_in, ys_out, labels32, dec_len, lab_len, n_valid = prep
        To = ys_in.shape[1]
        pe_, se_ = self._drop(2)           # dropout(emb * scale + PE)  (transformer_official.py:306-307)
        x = K.embed_pe_fwd(ys_in.reshape(-1), self.emb32, self.pe, self.d ** -0.5, B, To, self.dtype, drop_p=pe_, drop_seed=se_)
        cache = dict(B=B, T=T, To=To, ys_in=ys_in, layers=[], drop=(pe_, se_))
        if self._dec_exec_ok(B, To, T):
            x, bufs = self._dec_exec_fwd(x, enc, dec_len, cross_len, B, To, T)
            cache.update(exec_bufs=bufs, enc=enc, keep=(dec_len, cross_len))      # the plans hold raw pointers to the length vectors
            pred = self.prj.fwd(x)
            cache["x_last"] = x
            return pred, cache
        kv_pre = [None] * self.L
        if self.aux_overlap and not torch.cuda.is_current_stream_capturing():
            # the six cross-attention K|V projections of the encoder output (16000-row GEMMs, 26 us each) do not depend on the
            # decoder state: they run on the auxiliary stream while the main stream works through the decoder's small kernels
            self._fork(self.ctc_stream)
            with torch.cuda.stream(self.ctc_stream):
                for i, (slf, cross, ffn) in enumerate(self.dec):
                    kv = cross.kv.fwd(enc)
                    e = torch.cuda.Event()
                    e.record(self.ctc_stream)
                    kv_pre[i] = (kv, e)
            self._keep.append(enc)
            self._keep += [kv for kv, _ in kv_pre]      # blocks of the auxiliary stream's pool, read on the main stream
        for i, (slf, cross, ffn) in enumerate(self.dec):
            x1, c1 = self._attn_block_fwd(slf, x, x, B, To, To, dec_len, dec_len, True, -1, False, site=100 + 8 * i)
            x2, c2 = self._attn_block_fwd(cross, x1, enc, B, To, T, cross_len, dec_len, False, -1, True, site=102 + 8 * i, kv_pre=kv_pre[i])
            x, c3 = self._ffn_block_fwd(ffn, x2, B, To, dec_len, site=104 + 8 * i)
            cache["layers"].append((c1, c2, c3))
        pred = self.prj.fwd(x)
        cache["x_last"] = x
        return pred, cache

    def decoder_bwd(self, cache, dpred, d_enc, d_enc_ready=None):
        """dpred (B*To, V); accumulates the encoder-output gradient into d_enc (B*T, d) in place.
        d_enc_ready: event after which d_enc holds the CTC branch's contribution (ctc_branch_async); waited for before the
        first cross-attention block adds to it."""
        self._in_decoder = True
        self.wait_transposes()
        self._wgrad(self.prj, dpred, cache["x_last"])
        dy, dy2 = self.prj.dgrad(dpred), None
        if "exec_bufs" in cache:
            dy, dy2 = self._dec_exec_bwd(cache["exec_bufs"], dy, cache["enc"], d_enc, d_enc_ready)
        for i in (reversed(range(self.L)) if "exec_bufs" not in cache else ()):
            slf, cross, ffn = self.dec[i]
            c1, c2, c3 = cache["layers"][i]
            dx, dz = self._ffn_block_bwd(ffn, c3, dy, dy2)
            if d_enc_ready is not None:
                torch.cuda.current_stream().wait_event(d_enc_ready)
                d_enc_ready = None
            dx, dz = self._attn_block_bwd(cross, c2, dx, dz, d_kv_src=d_enc)
            dx, dz = self._attn_block_bwd(slf, c1, dx, dz)
            dy, dy2 = dx, dz
            self._ready(f"decoder.layer_stack.{i}.slf_attn.w_qs.weight")
        # gradient wrt the embedding output = projection path + residual path, added inside the scatter kernel.  The embedding's gradient IS
        # the output projection's (tied weights, transformer_official.py:253-256): the scatter's atomic adds and the projection's weight-gradient
        # GEMM (a plain read-modify-write when it has one M-split) must not overlap - the scatter goes to the weight-gradient stream, behind
        # the launch that holds the projection's problem.  (Round 5, tools/race_stress.py: with both on their own streams a one-layer decoder
        # replayed from a graph lost an update about once in 100 - 200 steps; in the six-layer step the two are 1.8 ms apart.)
        ids_flat, dyc, dy2c = cache["ys_in"].reshape(-1), dy.contiguous(), (dy2.contiguous() if dy2 is not None else None)
        self.flush_wgrads()
        if self.overlap_wgrad:
            self._disarm()
            self._fork(self.side)
            K.STREAM_OVERRIDE = self._side_handle
            try:
                K.embed_bwd(ids_flat, dyc, self.gemb, self.d ** -0.5, drop_p=cache["drop"][0], drop_seed=cache["drop"][1], dy2=dy2c)
            finally:
                K.STREAM_OVERRIDE = None
            self._keep += (ids_flat, dyc) + ((dy2c,) if dy2c is not None else ())
        else:
            K.embed_bwd(ids_flat, dyc, self.gemb, self.d ** -0.5, drop_p=cache["drop"][0], drop_seed=cache["drop"][1], dy2=dy2c)
        if self.aux_overlap and not torch.cuda.is_current_stream_capturing():      # nothing is forked to that stream while capturing
            torch.cuda.current_stream().wait_stream(self.ctc_stream)      # d_enc is complete (CTC branch + every cross-attention add)
        self._in_decoder = False
        self._ready("decoder.tgt_word_emb.weight")
        if self.use_ctc:
            self._ready("ctc_lo.weight")      # final since ctc_fwd_bwd, which ran before this function
