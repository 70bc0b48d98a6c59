"""Tensor-level wrappers over the C ABI: each takes torch CUDA tensors, checks what the kernel
assumes (device, dtype, contiguity, shapes) ON THE HOST before launching, and passes raw device
pointers + the current HIP stream.  PyTorch is only the allocator / stream provider here.
"""
import threading
import ctypes

import torch

from . import _lib
from ._lib import ASR_BF16, ASR_F32, ACT_NONE, ACT_RELU, LN_REDUCE_MAX, TN_GROUP_MAX, LnReduceItem, TnProblem, check
from ._lib import fast as lib      # vectorcall trampolines into the C ABI (_lib.py); pointers and stream handles are plain ints

_DT = {torch.float32: ASR_F32, torch.bfloat16: ASR_BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("the HIP path needs CUDA (ROCm) tensors; there is no CPU fallback")
    return t.data_ptr()


_DEV_INDEX = None
_TLS = threading.local()   # .own_stream = True on a helper thread (the loader's): its launches never follow STREAM_OVERRIDE, which the
                           # training thread sets around ITS side-stream launches
STREAM_OVERRIDE = None     # raw stream handle (int) the next launches go to instead of torch's current stream:
                           # the engine sets it around its side-stream launches of own kernels, which saves the
                           # `with torch.cuda.stream(...)` context switch (~6 us of host time, 25 times per step)


def _stream():
    """Raw handle of torch's current stream.  torch.cuda.current_stream() builds a Stream object and
    resolves the device index through several Python layers (~8 us, ~160 calls per step); the
    private raw getter is a single C call."""
    global _DEV_INDEX
    if STREAM_OVERRIDE is not None and not getattr(_TLS, "own_stream", False):
        return STREAM_OVERRIDE
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()   # one process per GPU: bind_device() (engine construction) or the first launch sets it
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def stream_fork(to_stream_handle, from_stream_handle=None):
    """Work queued on `to` from now on runs after what is queued on `from` (default: torch's current stream) so far.
    Raw stream handles (ints); ~2 us of host time against ~12 us for torch.cuda.Event record + wait."""
    frm = _stream() if from_stream_handle is None else from_stream_handle
    check(lib.asr_stream_fork(frm, to_stream_handle), "asr_stream_fork")


def stream_arm(to_stream_handle, from_stream_handle=None):
    """The next armed-capable entry point (include/asr_hip.h: asr_stream_arm) signals `to` from its last kernel on `from`."""
    frm = _stream() if from_stream_handle is None else from_stream_handle
    check(lib.asr_stream_arm(frm, to_stream_handle), "asr_stream_arm")


def stream_arm_pending():
    """True when the arm was NOT taken by a launch (cleared either way): fall back to stream_fork."""
    return bool(lib.asr_stream_arm_pending())


def bind_device(device):
    """One process drives ONE GPU: the launch stream is looked up on this device from now on.  Called when a model
    builds its engine; a model on a device other than torch's current one is refused (its kernels would be issued
    on the current device's stream against the other device's memory)."""
    global _DEV_INDEX
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        raise RuntimeError(f"the model lives on cuda:{idx} but torch's current device is cuda:{torch.cuda.current_device()}: call "
                           f"torch.cuda.set_device({idx}) first (one process per GPU)")
    if _DEV_INDEX is not None and _DEV_INDEX != idx:
        raise RuntimeError(f"this process already launches on cuda:{_DEV_INDEX}; a second GPU (cuda:{idx}) needs its own process")
    _DEV_INDEX = idx


def _chk_f32(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("expected a contiguous float32 tensor")


def _chk_i32(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise ValueError("expected a contiguous int32 tensor")


class LaunchTimer:
    """HIP-event timing of selected launches on the stream they are issued on (bench.py's roofline / kernels
    legs).  `work` is the algorithmic FLOP count of the launch, `nbytes` its algorithmic HBM bytes (0 = not stated)."""

    def __init__(self, names):
        self.names = set(names)
        self.rec = {n: [] for n in names}

    def run(self, name, work, fn, nbytes=0.0):
        if name not in self.names:
            return fn()
        st = torch.cuda.ExternalStream(STREAM_OVERRIDE) if STREAM_OVERRIDE is not None else None   # side-stream launches: record there
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st) if st is not None else e0.record()
        out = fn()
        e1.record(st) if st is not None else e1.record()
        self.rec[name].append((e0, e1, work, nbytes))
        return out

    def summary(self):
        """name -> dict(launches, total_ms, avg_us, work_per_s, work_per_launch, bytes_per_s, bytes_per_launch)
        (call after a device sync)."""
        out = {}
        for n, r in self.rec.items():
            if not r:
                continue
            ms = [a.elapsed_time(b) for a, b, _, _ in r]
            tot, work, nb = sum(ms), sum(x[2] for x in r), sum(x[3] for x in r)
            sec = tot * 1e-3
            out[n] = dict(launches=len(r), total_ms=tot, avg_us=1e3 * tot / len(r), work_per_s=work / sec if tot > 0 else 0.0,
                          work_per_launch=work / len(r), bytes_per_s=nb / sec if tot > 0 else 0.0, bytes_per_launch=nb / len(r))
        return out


TIMER = None   # set to a LaunchTimer by bench.py


def timed(name, work, fn, nbytes=0.0):
    if TIMER is None:
        return fn()
    return TIMER.run(name, work, fn, nbytes)


class Workspace:
    """Grow-only scratch buffer (never shrinks, so pointers stay valid under graph replay once
    the high-water mark has been reached during warm-up)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = max(int(nbytes), 16)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf


# --------------------------------------------------------------------------------- LayerNorm
def add_ln_fwd(x, res, gamma, beta, pe, lens, B, T, y=None, xhat=None, rstd=None, drop_p=0.0, drop_seed=0, drop_mode=0):
    """y = LN(x + res) * gamma + beta (+ pe[t]), rows t >= lens[b] zeroed.  xhat may alias x.
    drop_mode 1: dropout on x before the residual add; 2: dropout on the output."""
    d = x.shape[-1]
    assert x.is_contiguous() and x.numel() == B * T * d
    if res is not None:
        assert res.is_contiguous() and res.shape == x.shape and res.dtype == x.dtype
    _chk_f32(gamma, beta, pe)
    _chk_i32(lens)
    assert gamma.numel() == d and beta.numel() == d
    if pe is not None:
        assert pe.shape[-1] == d and pe.shape[-2] >= T
    if lens is not None:
        assert lens.numel() == B
    y = torch.empty_like(x) if y is None else y
    xhat = torch.empty_like(x) if xhat is None else xhat
    rstd = torch.empty(B * T, dtype=torch.float32, device=x.device) if rstd is None else rstd
    nb = (3 + (res is not None)) * x.numel() * x.element_size()      # x, (res), y, xhat: SURVEY 8(d) "LN/add/mask"
    timed("add_ln_fwd", 0.0, lambda: check(
        lib.asr_add_ln_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(pe), _p(lens), _p(y), _p(xhat), _p(rstd),
                           B, T, d, float(drop_p), int(drop_seed) & 0xFFFFFFFF, int(drop_mode), _dt(x), _stream()), "asr_add_ln_fwd"), nb)
    return y, xhat, rstd


def add_ln_bwd(dy, dy2, xhat, rstd, gamma, lens, dgamma, dbeta, dbias, B, T, ws, dz=None, drop_p=0.0, drop_seed=0, drop_mode=0,
               partials=None):
    """Returns (dz, dx): dz = gradient wrt the residual input; dx = gradient wrt x (the same tensor
    unless pre-residual dropout is active, then dz * keep / (1-p)).
    partials: a uint8 buffer of add_ln_bwd_workspace_bytes(B * T, d) that receives the per-workgroup partial sums of
    the parameter gradients INSTEAD of their reduction into dgamma / dbeta / dbias; the caller reduces several
    sites at once with add_ln_bwd_reduce_batched."""
    d = dy.shape[-1]
    assert dy.is_contiguous() and xhat.is_contiguous() and dy.numel() == B * T * d == xhat.numel()
    assert xhat.dtype == dy.dtype and (dy2 is None or (dy2.dtype == dy.dtype and dy2.is_contiguous() and dy2.numel() == dy.numel()))
    _chk_f32(rstd, gamma, dgamma, dbeta, dbias)
    _chk_i32(lens)
    assert rstd.numel() == B * T and dgamma.numel() == d and dbeta.numel() == d and (dbias is None or dbias.numel() == d)
    dz = torch.empty_like(dy) if dz is None else dz
    dx = torch.empty_like(dy) if (drop_p > 0 and drop_mode == 1) else None
    nbytes = lib.asr_add_ln_bwd_workspace_bytes(B * T, d)
    if partials is not None:
        assert partials.dtype == torch.uint8 and partials.numel() >= nbytes
        w, dgamma, dbeta = partials, None, None
    else:
        w = ws.get(nbytes)
    nb = (3 + (dy2 is not None) + (dx is not None)) * dy.numel() * dy.element_size()     # dy, (dy2), xhat, dz, (dx)
    timed("add_ln_bwd", 0.0, lambda: check(
        lib.asr_add_ln_bwd(_p(dy), _p(dy2), _p(xhat), _p(rstd), _p(gamma), _p(lens), _p(dz), _p(dx), _p(dgamma), _p(dbeta),
                           _p(dbias), _p(w), w.numel(), B, T, d, float(drop_p), int(drop_seed) & 0xFFFFFFFF, int(drop_mode),
                           _dt(dy), _stream()), "asr_add_ln_bwd"), nb)
    return dz, (dx if dx is not None else dz)


def add_ln_bwd_workspace_bytes(rows, d):
    return lib.asr_add_ln_bwd_workspace_bytes(rows, d)


def add_ln_bwd_reduce_batched(items, d):
    """items: list of (partials, dgamma, dbeta, dbias or None, rows) left by add_ln_bwd(partials=...)."""
    for i in range(0, len(items), LN_REDUCE_MAX):
        chunk = items[i:i + LN_REDUCE_MAX]
        arr = (LnReduceItem * len(chunk))()
        for q, (part, dg, db, dbias, rows) in zip(arr, chunk):
            _chk_f32(dg, db, dbias)
            assert dg.numel() == d and db.numel() == d and (dbias is None or dbias.numel() == d)
            assert part.numel() >= lib.asr_add_ln_bwd_workspace_bytes(rows, d)
            q.ws, q.dgamma, q.dbeta, q.dbias, q.rows = _p(part), _p(dg), _p(db), _p(dbias), rows
        check(lib.asr_add_ln_bwd_reduce_batched(ctypes.addressof(arr), len(chunk), d, _stream()), "asr_add_ln_bwd_reduce_batched")


# --------------------------------------------------------------------------------- attention
def _strided_rows(t, H, dk):
    """t is a (rows, >=H*dk) view whose last dim is contiguous; returns row stride in elements."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.shape[1] == H * dk
    return t.stride(0)


def sdpa_fwd(q, k, v, k_len, B, H, Tq, Tk, dk, causal=False, window=-1, scale=None, o=None, lse=None, drop_p=0.0, drop_seed=0, o_lo=None):
    """q: (B*Tq, H*dk) view, k/v: (B*Tk, H*dk) views (may be column slices of a fused buffer).
    o_lo: a tensor of o's shape and strides for the low-order piece of the bf16 output (asr_hip.h; hand it to sdpa_bwd), or None."""
    ldq, ldk, ldv = _strided_rows(q, H, dk), _strided_rows(k, H, dk), _strided_rows(v, H, dk)
    assert q.shape[0] == B * Tq and k.shape[0] == B * Tk and v.shape[0] == B * Tk
    assert q.dtype == k.dtype == v.dtype
    _chk_i32(k_len)
    assert k_len is None or k_len.numel() == B
    o = torch.empty(B * Tq, H * dk, dtype=q.dtype, device=q.device) if o is None else o
    ldo = _strided_rows(o, H, dk)
    lse = torch.empty(B, H, Tq, dtype=torch.float32, device=q.device) if lse is None else lse
    assert o_lo is None or (o_lo.dtype == o.dtype == torch.bfloat16 and o_lo.shape == o.shape and o_lo.stride() == o.stride())
    scale = float(dk) ** -0.5 if scale is None else float(scale)
    e = q.element_size()
    timed("sdpa_fwd", 4.0 * B * H * Tq * Tk * dk, lambda: check(
        lib.asr_sdpa_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(k_len), B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo,
                         int(causal), int(window), scale, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(o_lo), _dt(q), _stream()), "asr_sdpa_fwd"),
          2.0 * B * H * (Tq + Tk) * dk * e)          # Q, O + K, V (4 B H T dk e at Tq = Tk)
    return o, lse


def sdpa_bwd(q, k, v, o, do, lse, k_len, B, H, Tq, Tk, dk, dq, dk_, dv, causal=False, window=-1, scale=None, delta=None,
             drop_p=0.0, drop_seed=0, o_lo=None):
    ldq, ldk, ldv, ldo = (_strided_rows(t, H, dk) for t in (q, k, v, o))
    assert _strided_rows(do, H, dk) == ldo and _strided_rows(dq, H, dk) == ldq
    assert _strided_rows(dk_, H, dk) == ldk and _strided_rows(dv, H, dk) == ldv
    assert q.dtype == k.dtype == v.dtype == o.dtype == do.dtype == dq.dtype == dk_.dtype == dv.dtype
    _chk_i32(k_len)
    _chk_f32(lse)
    assert o_lo is None or (o_lo.dtype == o.dtype == torch.bfloat16 and o_lo.shape == o.shape and o_lo.stride() == o.stride())
    if delta is None:      # scratch: row sums of dO o O, or the band kernel's dQ partials of the tiles on a key-block boundary
        need = lib.asr_sdpa_bwd_workspace_bytes(B, H, Tq, Tk, dk, int(causal), int(window), _dt(q))
        delta = torch.empty((need + 3) // 4, dtype=torch.float32, device=q.device)
    scale = float(dk) ** -0.5 if scale is None else float(scale)
    e = q.element_size()
    timed("sdpa_bwd", 10.0 * B * H * Tq * Tk * dk, lambda: check(
        lib.asr_sdpa_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), delta.numel() * 4, _p(dq), _p(dk_), _p(dv), _p(k_len),
                         B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, int(causal), int(window), scale, float(drop_p),
                         int(drop_seed) & 0xFFFFFFFF, _p(o_lo), _dt(q), _stream()), "asr_sdpa_bwd"),
          B * H * (4.0 * Tq + 4.0 * Tk) * dk * e)    # Q, O, dO, dQ + K, V, dK, dV (8 x 16.4 MB at config 2); 5 products
    return dq, dk_, dv


# --------------------------------------------------------------------------------- losses
def _frame_rows(logits):
    """Row stride (elements) of a (B, T, V) tensor of frames: dense, or a view of rows padded to `ld` >= V elements."""
    B, T, V = logits.shape
    ld = logits.stride(1) if B * T > 1 else V
    assert logits.stride(2) == 1 and ld >= V and (B == 1 or logits.stride(0) == T * ld), f"frames must be rows of one (B*T, ld) buffer: {logits.stride()}"
    return ld


def ctc_fwd_bwd(logits, in_len, labels, lab_len, ws, blank=0, grad_scale=1.0, zero_infinity=False, dlogits=None,
                want_grad=True, nll=None, grad_scale_div=None, best_path=None):
    """logits (B,T,V); returns (nll (B,), dlogits or None).  dlogits may alias logits.
    grad_scale_div: optional 1-element f32 device tensor; the gradient scale is then grad_scale / grad_scale_div[0].
    best_path: optional (B, T) int32 tensor that receives the frame-wise argmax of the logits (the greedy CTC path; see ctc_collapse)."""
    B, T, V = logits.shape
    ld = _frame_rows(logits)
    _chk_i32(in_len, labels, lab_len)
    Lmax = labels.shape[1]
    assert labels.shape[0] == B and in_len.numel() == B and lab_len.numel() == B
    nll = torch.empty(B, dtype=torch.float32, device=logits.device) if nll is None else nll
    if want_grad and dlogits is None:
        dlogits = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
    if dlogits is not None:
        assert dlogits.shape == logits.shape and dlogits.stride() == logits.stride() and dlogits.dtype == logits.dtype
    w = ws.get(lib.asr_ctc_workspace_bytes(B, T, Lmax))
    _chk_f32(grad_scale_div)
    _chk_i32(best_path)
    assert best_path is None or best_path.numel() == B * T
    timed("ctc", 0.0, lambda: check(
        lib.asr_ctc_fwd_bwd(_p(logits), _p(dlogits), _p(in_len), _p(labels), _p(lab_len), _p(nll), B, T, V, ld, Lmax,
                            int(blank), float(grad_scale), _p(grad_scale_div), int(zero_infinity), _p(best_path), _p(w), w.numel(), _dt(logits), _stream()),
        "asr_ctc_fwd_bwd"), (3.0 if dlogits is not None else 1.0) * logits.numel() * logits.element_size())   # 3 B T V e

    return nll, dlogits


def ctc_greedy_decode(logits, in_len, blank=0):
    """logits (B,T,V) -> (ids (B,T) int32, collapsed and 0-padded; lens (B,) int32)."""
    B, T, V = logits.shape
    ld = _frame_rows(logits)
    _chk_i32(in_len)
    ids = torch.empty(B, T, dtype=torch.int32, device=logits.device)
    lens = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.asr_ctc_greedy_decode(_p(logits), _p(in_len), _p(ids), _p(lens), B, T, V, ld, int(blank), _dt(logits), _stream()),
          "asr_ctc_greedy_decode")
    return ids, lens


def ctc_collapse(path, in_len, blank=0):
    """CTC collapse IN PLACE of a frame-wise best path (B, T) int32 (ctc_fwd_bwd's best_path): repeats merged, blanks dropped, 0-padded.
    Returns (path, lens (B,) int32)."""
    B, T = path.shape
    _chk_i32(path, in_len)
    lens = torch.empty(B, dtype=torch.int32, device=path.device)
    check(lib.asr_ctc_collapse(_p(path), _p(in_len), _p(lens), B, T, int(blank), _stream()), "asr_ctc_collapse")
    return path, lens


def decode_attn(q, k, v, H, dk, Tk_cap, kv_div=1, k_len=None, k_len_uniform=0, len_div=1, scale=None, o=None):
    """Single-query attention of R = q.shape[0] rows over cached keys/values (see include/asr_hip.h)."""
    R = q.shape[0]
    ldq, ldk, ldv = _strided_rows(q, H, dk), _strided_rows(k, H, dk), _strided_rows(v, H, dk)
    assert q.dtype == k.dtype == v.dtype
    _chk_i32(k_len)
    o = torch.empty(R, H * dk, dtype=q.dtype, device=q.device) if o is None else o
    scale = float(dk) ** -0.5 if scale is None else float(scale)
    check(lib.asr_decode_attn(_p(q), _p(k), _p(v), _p(o), _p(k_len), int(k_len_uniform), int(len_div), R, H, dk, int(Tk_cap), int(kv_div),
                              ldq, ldk, ldv, _strided_rows(o, H, dk), scale, _dt(q), _stream()), "asr_decode_attn")
    return o


def logsoftmax_topk(logits, beam):
    R, V = logits.shape
    assert logits.stride(1) == 1
    vals = torch.empty(R, beam, dtype=torch.float32, device=logits.device)
    ids = torch.empty(R, beam, dtype=torch.int32, device=logits.device)
    check(lib.asr_logsoftmax_topk(_p(logits), _p(vals), _p(ids), R, V, logits.stride(0), int(beam), _dt(logits), _stream()), "asr_logsoftmax_topk")
    return vals, ids


def ctc_frame_topk(logits, k, blank=0):
    """logits (R, V) -> (vals (R, k) f32 log_softmax of the k best classes per frame, ids (R, k) int32, blank_lp (R) f32)."""
    R, V = logits.shape
    assert logits.stride(1) == 1
    vals = torch.empty(R, k, dtype=torch.float32, device=logits.device)
    ids = torch.empty(R, k, dtype=torch.int32, device=logits.device)
    blank_lp = torch.empty(R, dtype=torch.float32, device=logits.device)
    check(lib.asr_ctc_frame_topk(_p(logits), _p(vals), _p(ids), _p(blank_lp), R, V, logits.stride(0), int(k), int(blank), _dt(logits), _stream()),
          "asr_ctc_frame_topk")
    return vals, ids, blank_lp


def ctc_prefix_beam(vals, ids, blank_lp, in_len, B, T, beam, nbest, blank=0, max_len=None):
    """CTC prefix beam search on the device over the per-frame candidates of ctc_frame_topk (include/asr_hip.h).
    Returns (tokens (B, nbest, Lcap) int32, lengths (B, nbest) int32 with -1 for missing ranks, scores (B, nbest) float32)."""
    k = vals.shape[1]
    assert vals.shape == (B * T, k) and ids.shape == (B * T, k) and blank_lp.numel() == B * T
    _chk_f32(vals, blank_lp)
    _chk_i32(ids, in_len)
    Lcap = int(T if max_len is None else max_len)
    ws = torch.empty(lib.asr_ctc_prefix_beam_workspace_bytes(B, T, beam), dtype=torch.uint8, device=vals.device)
    out_tok = torch.zeros(B, nbest, Lcap, dtype=torch.int32, device=vals.device)
    out_len = torch.empty(B, nbest, dtype=torch.int32, device=vals.device)
    out_score = torch.empty(B, nbest, dtype=torch.float32, device=vals.device)
    check(lib.asr_ctc_prefix_beam(_p(vals), _p(ids), _p(blank_lp), _p(in_len), _p(ws), ws.numel(), _p(out_tok), _p(out_len), _p(out_score),
                                  B, T, k, int(beam), int(nbest), Lcap, int(blank), _stream()), "asr_ctc_prefix_beam")
    return out_tok, out_len, out_score


def beam_step(top_vals, top_ids, score, alive, last_tok, parent, rec_tok, rec_par, rec_end, rec_score, maxlen, alive_total, B, beam, step, eos):
    _chk_f32(top_vals, score, rec_score)
    _chk_i32(top_ids, alive, last_tok, parent, rec_tok, rec_par, rec_end, maxlen, alive_total)
    check(lib.asr_beam_step(_p(top_vals), _p(top_ids), _p(score), _p(alive), _p(last_tok), _p(parent), _p(rec_tok), _p(rec_par), _p(rec_end),
                            _p(rec_score), _p(maxlen), _p(alive_total), B, beam, int(step), int(eos), _stream()), "asr_beam_step")


def cache_gather(src, dst, parent, L, R, beam, Lcap, n_pos, row_bytes):
    _chk_i32(parent)
    check(lib.asr_cache_gather(_p(src), _p(dst), _p(parent), L, R, beam, Lcap, int(n_pos), int(row_bytes), _stream()), "asr_cache_gather")


def xent_fwd_bwd(logits, gold, n_valid, ignore_index=0, smoothing=0.0, grad_scale=1.0, dlogits=None, want_grad=True,
                 row_nll=None, argmax=None):
    """argmax: optional (M) int32 tensor that receives every row's greedy class (first index of the maximum, ignored rows included)."""
    M, V = logits.shape
    assert logits.is_contiguous() and gold.numel() == M
    _chk_i32(gold, argmax)
    assert argmax is None or argmax.numel() == M
    _chk_f32(n_valid)
    row_nll = torch.empty(M, dtype=torch.float32, device=logits.device) if row_nll is None else row_nll
    if want_grad and dlogits is None:
        dlogits = torch.empty_like(logits)
    timed("xent", 0.0, lambda: check(
        lib.asr_xent_fwd_bwd(_p(logits), _p(gold), _p(n_valid), _p(row_nll), _p(dlogits), M, V, int(ignore_index),
                             float(smoothing), float(grad_scale), _p(argmax), _dt(logits), _stream()), "asr_xent_fwd_bwd"),
          (3.0 if dlogits is not None else 1.0) * logits.numel() * logits.element_size())                      # 3 M V e
    return row_nll, dlogits


def loss_combine(row_nll, n_valid, nll, w_ce, w_ctc, out=None):
    dev = (row_nll if row_nll is not None else nll).device
    out = torch.empty(3, dtype=torch.float32, device=dev) if out is None else out
    M = row_nll.numel() if row_nll is not None else 0
    B = nll.numel() if nll is not None else 0
    check(lib.asr_loss_combine(_p(row_nll), M, _p(n_valid), _p(nll), B, float(w_ce), float(w_ctc), _p(out), _stream()),
          "asr_loss_combine")
    return out


# --------------------------------------------------------------------------------- decoder glue
def dec_preprocess(tgt, sos=2, eos=3, lens64=()):
    """tgt (B, Lmax) int64 zero-padded -> ys_in, ys_out (B, Lmax+1) int32, labels32, dec_len, lab_len, n_valid.
    lens64: up to two (B,) int64 length vectors of the batch (wave_len, tgt_len); their int32 copies are made by the same launch and
    returned as a 7th element (a tuple)."""
    assert tgt.dtype == torch.int64 and tgt.is_contiguous() and tgt.dim() == 2
    B, Lmax = tgt.shape
    dev = tgt.device
    lens64 = tuple(lens64)
    assert len(lens64) <= 2 and all(t.dtype == torch.int64 and t.is_contiguous() and t.numel() == B and t.device == dev for t in lens64)
    lens32 = tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in lens64)
    la, lb = (list(zip(lens64, lens32)) + [(None, None), (None, None)])[:2]
    ys_in = torch.empty(B, Lmax + 1, dtype=torch.int32, device=dev)
    ys_out = torch.empty(B, Lmax + 1, dtype=torch.int32, device=dev)
    labels32 = torch.empty(B, Lmax, dtype=torch.int32, device=dev)
    dec_len = torch.empty(B, dtype=torch.int32, device=dev)
    lab_len = torch.empty(B, dtype=torch.int32, device=dev)
    n_valid = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib.asr_dec_preprocess(_p(tgt), _p(ys_in), _p(ys_out), _p(labels32), _p(dec_len), _p(lab_len), _p(n_valid),
                                 B, Lmax, sos, eos, _p(la[0]), _p(la[1]), _p(lb[0]), _p(lb[1]), _stream()), "asr_dec_preprocess")
    if lens64:
        return ys_in, ys_out, labels32, dec_len, lab_len, n_valid, lens32
    return ys_in, ys_out, labels32, dec_len, lab_len, n_valid


def embed_pe_fwd(ids, emb, pe, scale, B, To, dtype, y=None, drop_p=0.0, drop_seed=0):
    V, d = emb.shape
    _chk_i32(ids)
    _chk_f32(emb, pe)
    assert ids.numel() == B * To and pe.shape[-1] == d and pe.shape[-2] >= To
    y = torch.empty(B * To, d, dtype=dtype, device=emb.device) if y is None else y
    check(lib.asr_embed_pe_fwd(_p(ids), _p(emb), _p(pe), _p(y), float(scale), B, To, d, V, float(drop_p),
                               int(drop_seed) & 0xFFFFFFFF, _dt(y), _stream()), "asr_embed_pe_fwd")
    return y


def embed_bwd(ids, dy, demb, scale, drop_p=0.0, drop_seed=0, dy2=None):
    V, d = demb.shape
    _chk_i32(ids)
    _chk_f32(demb)
    assert dy.is_contiguous() and dy.shape[-1] == d and dy.numel() == ids.numel() * d
    assert dy2 is None or (dy2.is_contiguous() and dy2.shape == dy.shape and dy2.dtype == dy.dtype)
    check(lib.asr_embed_bwd(_p(ids), _p(dy), _p(dy2), _p(demb), float(scale), ids.numel(), d, V, float(drop_p),
                            int(drop_seed) & 0xFFFFFFFF, _dt(dy), _stream()), "asr_embed_bwd")


def dropout_mask(rows, cols, drop_p, drop_seed, device="cuda"):
    """(rows, cols) uint8 keep mask of the LayerNorm / embedding dropout sites (tests)."""
    m = torch.empty(rows, cols, dtype=torch.uint8, device=device)
    check(lib.asr_dropout_mask(_p(m), rows, cols, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream()), "asr_dropout_mask")
    return m


def sdpa_dropout_mask(B, H, Tq, Tk, drop_p, drop_seed, device="cuda"):
    m = torch.empty(B, H, Tq, Tk, dtype=torch.uint8, device=device)
    check(lib.asr_sdpa_dropout_mask(_p(m), B, H, Tq, Tk, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream()), "asr_sdpa_dropout_mask")
    return m


# --------------------------------------------------------------------------------- elementwise
def relu_(x):
    assert x.is_contiguous()
    check(lib.asr_relu_fwd(_p(x), x.numel(), _dt(x), _stream()), "asr_relu_fwd")
    return x


def relu_bwd_(da, a, dbias, ws):
    """da *= (a > 0) in place; dbias (f32) += column sums."""
    assert da.is_contiguous() and a.is_contiguous() and da.shape == a.shape and da.dtype == a.dtype and da.dim() == 2
    _chk_f32(dbias)      # None: mask only (the bias gradient then comes from the weight-gradient GEMM)
    rows, cols = da.shape
    w = ws.get(lib.asr_colsum_workspace_bytes(rows, cols))
    check(lib.asr_relu_bwd(_p(da), _p(a), _p(dbias), _p(w), w.numel(), rows, cols, _dt(da), _stream()), "asr_relu_bwd")
    return da


def colsum(x, out, ws, accumulate=True):
    assert x.dim() == 2 and x.stride(1) == 1
    _chk_f32(out)
    rows, cols = x.shape
    assert out.numel() == cols
    w = ws.get(lib.asr_colsum_workspace_bytes(rows, cols))
    check(lib.asr_colsum(_p(x), _p(out), _p(w), w.numel(), rows, cols, x.stride(0), int(accumulate), _dt(x), _stream()),
          "asr_colsum")
    return out


def cast(src, dst):
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    check(lib.asr_cast(_p(src), _p(dst), src.numel(), _dt(src), _dt(dst), _stream()), "asr_cast")
    return dst


# --------------------------------------------------------------------------------- optimizer
def grad_sumsq(g, out, ws):
    _chk_f32(g, out)
    w = ws.get(lib.asr_sumsq_workspace_bytes(g.numel()))
    timed("grad_sumsq", 0.0, lambda: check(lib.asr_grad_sumsq(_p(g), g.numel(), _p(out), _p(w), w.numel(), _stream()), "asr_grad_sumsq"), 4.0 * g.numel())
    return out


def grad_sumsq_noam(g, out, ws, step, hyper, model_size, warmup, factor, lr_const, b1, b2):
    """grad_sumsq + noam_hyper with the schedule update inside the norm's finalizer (asr_grad_sumsq_noam)."""
    _chk_f32(g, out, hyper)
    _chk_i32(step)
    assert hyper.numel() >= 4
    w = ws.get(lib.asr_sumsq_workspace_bytes(g.numel()))
    timed("grad_sumsq", 0.0, lambda: check(lib.asr_grad_sumsq_noam(_p(g), g.numel(), _p(out), _p(w), w.numel(), _p(step), _p(hyper), float(model_size),
                                                                   float(warmup), float(factor), float(lr_const), float(b1), float(b2), _stream()),
                                           "asr_grad_sumsq_noam"), 4.0 * g.numel())
    return out


def noam_hyper(step, hyper, model_size, warmup, factor, lr_const, b1, b2):
    _chk_i32(step)
    _chk_f32(hyper)
    assert hyper.numel() >= 4
    check(lib.asr_noam_hyper(_p(step), _p(hyper), float(model_size), float(warmup), float(factor), float(lr_const),
                             float(b1), float(b2), _stream()), "asr_noam_hyper")


def adam_step(p, g, m, v, p_lp, hyper, sumsq, max_norm, b1, b2, eps, write_clipped=True):
    _chk_f32(p, g, m, v, hyper, sumsq)
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    if p_lp is not None:
        assert p_lp.dtype == torch.bfloat16 and p_lp.numel() == n and p_lp.is_contiguous()
    # p, g, m, v read + p, m, v (+ clipped g) written in fp32, bf16 shadow written: 28 (+4) (+2) B per parameter (~30)
    timed("adam", 0.0, lambda: check(
        lib.asr_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_lp), n, _p(hyper), _p(sumsq), float(max_norm), float(b1),
                          float(b2), float(eps), int(write_clipped), _stream()), "asr_adam_step"),
          (28.0 + (4.0 if write_clipped else 0.0) + (2.0 if p_lp is not None else 0.0)) * n)


# --------------------------------------------------------------------------------- GEMM
def gemm_nt_supported(M, N, K, lda, ldb, ldc):
    return K % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0 and ldc % 4 == 0


def transpose_batched(src, dst, tiles):
    """Transposed copies of the matrices listed in `tiles` (int32 (ntiles, 6), see include/asr_hip.h) from the flat
    bf16 buffer src into dst (each copy at its own offset and row stride)."""
    assert src.dtype == dst.dtype == torch.bfloat16 and tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.shape[1] == 6
    check(lib.asr_transpose_batched_bf16(_p(src), _p(dst), _p(tiles), tiles.shape[0], _stream()), "asr_transpose_batched_bf16")


def gemm_nt(a, w, bias, out, act=ACT_NONE, res=None, family="gemm_nt"):
    """out (M,N) = act(a (M,K) @ w (N,K)^T + bias) (+ res); bf16 operands, MFMA kernel.
    act = ACT_RELU_MASK: out = (a @ w^T) where res > 0 else 0 (res = the activations of a ReLU whose backward this is; no bias)."""
    assert a.dtype == w.dtype == out.dtype == torch.bfloat16
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and out.shape == (M, N) and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    _chk_f32(bias)
    if res is not None:
        assert res.dtype == torch.bfloat16 and res.shape == out.shape and res.stride() == out.stride()
    timed(family, 2.0 * M * N * K, lambda: check(
        lib.asr_gemm_nt_bf16(_p(a), _p(w), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), int(act),
                             _stream()), "asr_gemm_nt_bf16"))
    return out


_CU_LIMIT_NAME = ctypes.create_string_buffer(b"cu_limit")


def set_cu_limit(n):
    """Tuning option "cu_limit" on the launch path (no ctypes foreign call): the one-workgroup-per-CU kernels launched from now on are sized
    for n CUs (0 = the whole device).  The engine brackets the large launches that run beside the decoder's small kernels with it."""
    check(lib.asr_set_option(ctypes.addressof(_CU_LIMIT_NAME), int(n), None), "asr_set_option")


def set_option(name, value):
    """asr_set_option (include/asr_hip.h): process-wide tuning switches under which every value gives correct results; returns the previous value."""
    prev = ctypes.c_int(0)
    check(_lib.lib.asr_set_option(name.encode(), int(value), ctypes.byref(prev)), "asr_set_option")
    return prev.value


def deterministic():
    """True when the library's reductions run in a fixed order (asr_set_deterministic / ASR_DETERMINISTIC=1)."""
    return bool(lib.asr_get_deterministic())


def set_deterministic(on):
    """Process-wide switch (see include/asr_hip.h); returns the previous value.  Engines read it when they are built."""
    return bool(lib.asr_set_deterministic(int(bool(on))))


def gemm_f32(a, b, out, bias=None, trans_a=False, trans_b=False, act=ACT_NONE, mask=None, accumulate=False):
    """out (M, N) (+)= act(op(a) @ op(b) + bias) in fp32 on the matrix cores (asr_gemm_f32): a is (M, K), or (K, M) when trans_a;
    b is (K, N), or (N, K) when trans_b; mask (ACT_RELU_MASK): the activations of the ReLU whose backward this is, laid out like out."""
    assert a.dtype == b.dtype == out.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    M, N = out.shape
    K = a.shape[0] if trans_a else a.shape[1]
    assert a.shape == ((K, M) if trans_a else (M, K)) and b.shape == ((N, K) if trans_b else (K, N)), (a.shape, b.shape, out.shape)
    _chk_f32(bias)
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == out.shape and mask.stride() == out.stride()
    timed("gemm_f32", 2.0 * M * N * K, lambda: check(
        lib.asr_gemm_f32(_p(a), _p(b), _p(bias), _p(mask), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), int(trans_a), int(trans_b), int(act),
                         int(accumulate), _stream()), "asr_gemm_f32"))
    return out


def gemm_small(a, bm, bias, out, trans_b=False, act=ACT_NONE, mask=None):
    """Small-M projection (see include/asr_hip.h): out (M, N) = act(a (M, K) @ bm^T + bias) with bm (N, K), or a @ bm with bm (K, N) when trans_b."""
    M, K = a.shape
    N = out.shape[1]
    assert a.dtype == bm.dtype == out.dtype == torch.bfloat16 and out.shape[0] == M
    assert bm.shape == ((K, N) if trans_b else (N, K)) and a.stride(1) == 1 and bm.stride(1) == 1 and out.stride(1) == 1
    _chk_f32(bias)
    if mask is not None:
        assert mask.dtype == torch.bfloat16 and mask.shape == out.shape and mask.stride() == out.stride()
    timed("gemm_small", 2.0 * M * N * K, lambda: check(
        lib.asr_gemm_small_bf16(_p(a), _p(bm), _p(bias), _p(mask), _p(out), M, N, K, a.stride(0), bm.stride(0), out.stride(0), int(trans_b), int(act),
                                _stream()), "asr_gemm_small_bf16"))
    return out


def gemm_tn(dy, x, dw, accumulate=True, dbias=None, ws=None):
    """dw (N,K) f32 (+)= dy (M,N)^T @ x (M,K); bf16 operands, MFMA kernel.  dbias (N) f32 += column sums of dy.
    ws: a Workspace - needed in deterministic mode only (partial slabs, one per M-split)."""
    assert dy.dtype == x.dtype == torch.bfloat16 and dw.dtype == torch.float32
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dw.shape == (N, K) and dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
    _chk_f32(dbias)
    assert dbias is None or dbias.numel() == N
    need = lib.asr_gemm_tn_workspace_bytes(M, N, K)
    w = None
    if need:
        w = ws.get(need) if ws is not None else torch.empty(need, dtype=torch.uint8, device=dy.device)
    timed("gemm_tn", 2.0 * M * N * K, lambda: check(
        lib.asr_gemm_tn_bias_bf16(_p(dy), _p(x), _p(dw), _p(dbias), M, N, K, dy.stride(0), x.stride(0), dw.stride(0), int(accumulate), _p(w),
                                  w.numel() if w is not None else 0, _stream()), "asr_gemm_tn_bias_bf16"))
    return dw


_TN_SIZE = ctypes.sizeof(TnProblem)
_TN_BUF = (ctypes.c_char * (_TN_SIZE * TN_GROUP_MAX))()
_TN_ADDR = ctypes.addressof(_TN_BUF)
_TN_PACK = __import__("struct").Struct("<QQQQiiiiii").pack_into
assert _TN_SIZE == 56      # asr_tn_problem: four pointers, six ints


def gemm_tn_grouped(problems, accumulate=True):
    """problems: list of (dy (M,N) bf16, x (M,K) bf16, dw (N,K) f32, dbias (N) f32 or None);
    dw_p (+)= dy_p^T @ x_p and dbias_p += column sums of dy_p for all of them, TN_GROUP_MAX per launch."""
    for i in range(0, len(problems), TN_GROUP_MAX):
        chunk = problems[i:i + TN_GROUP_MAX]
        flops = 0.0
        for j, (dy, x, dw, dbias) in enumerate(chunk):
            assert dy.dtype == x.dtype == torch.bfloat16 and dw.dtype == torch.float32
            M, N = dy.shape
            Kd = x.shape[1]
            assert x.shape[0] == M and dw.shape == (N, Kd) and dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
            _chk_f32(dbias)
            assert dbias is None or dbias.numel() == N
            # one pack per problem into a reused host buffer (the launch copies the descriptors into kernel arguments before it
            # returns): field-by-field ctypes stores cost ~25 us per decoder layer, in the host-bound part of the joint step
            _TN_PACK(_TN_BUF, j * _TN_SIZE, _p(dy), _p(x), _p(dw), _p(dbias) or 0, M, N, Kd, dy.stride(0), x.stride(0), dw.stride(0))
            flops += 2.0 * M * N * Kd
        timed("gemm_tn", flops, lambda: check(lib.asr_gemm_tn_grouped_bf16(_TN_ADDR, len(chunk), int(accumulate), _stream()),
                                              "asr_gemm_tn_grouped_bf16"))


def token_table(id2token, device):
    """(tok_cp, tok_off, max_tok_len) for cer(): code points of every token string, back to back."""
    cps, off = [], [0]
    for tok in id2token:
        cps.extend(ord(c) for c in tok)
        off.append(len(cps))
    mx = max((off[i + 1] - off[i] for i in range(len(id2token))), default=0)
    return (torch.tensor(cps if cps else [0], dtype=torch.int32, device=device), torch.tensor(off, dtype=torch.int32, device=device), mx)


def cer(hyp, ref, table, pad_id, hyp_len=None, ref_len=None):
    """Per-utterance character error rate (B,) f32 on the device, reference string convention
    (score.py:4-13 over vocab.py:75-79 strings).  hyp (B, Lh), ref (B, Lr) int32 ids."""
    tok_cp, tok_off, mx = table
    _chk_i32(hyp, ref, hyp_len, ref_len)
    assert hyp.dim() == 2 and ref.dim() == 2 and hyp.shape[0] == ref.shape[0] and hyp.stride(1) == 1 and ref.stride(1) == 1
    B = hyp.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=hyp.device)
    check(lib.asr_cer(_p(hyp), _p(hyp_len), hyp.shape[1], hyp.stride(0), _p(ref), _p(ref_len), ref.shape[1], ref.stride(0), _p(tok_cp), _p(tok_off),
                      tok_off.numel() - 1, mx, int(pad_id), B, _p(out), _stream()), "asr_cer")
    return out


# --------------------------------------------------------------------------------- front end
def logmel(wav, wav_len, window, melfb, Tmax, feat=None):
    _chk_f32(wav, window, melfb)
    _chk_i32(wav_len)
    B, Smax = wav.shape
    n_mels = melfb.shape[1]
    assert melfb.shape[0] == 201 and window.numel() == 400 and wav_len.numel() == B
    feat = torch.empty(B, Tmax, n_mels, dtype=torch.float32, device=wav.device) if feat is None else feat
    check(lib.asr_logmel_fwd(_p(wav), _p(wav_len), _p(window), _p(melfb), _p(feat), B, Smax, Tmax, n_mels, _stream()),
          "asr_logmel_fwd")
    return feat


def utt_norm_lfr(feat, wav_len, m, n, Tlfr_max, dtype=torch.float32, masks=None):
    """masks: optional (B, 4) int32 [t0, t1, f0, f1] SpecAugment ranges (see include/asr_hip.h)."""
    _chk_f32(feat)
    _chk_i32(wav_len, masks)
    B, Tmax, n_mels = feat.shape
    assert masks is None or tuple(masks.shape) == (B, 4)
    out = torch.empty(B, Tlfr_max, m * n_mels, dtype=dtype, device=feat.device)
    out_len = torch.empty(B, dtype=torch.int32, device=feat.device)
    check(lib.asr_utt_norm_augment_lfr_fwd(_p(feat), _p(wav_len), _p(masks), _p(out), _p(out_len), B, Tmax, n_mels, m, n, Tlfr_max,
                                           _dt(out), _stream()), "asr_utt_norm_augment_lfr_fwd")
    return out, out_len
