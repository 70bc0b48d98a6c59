"""TransformerOffical / TransformerCTC: the reference's model plugin surface on the HIP engine.

Mirrors Predictor/Models/transformer_official.py:34-125 of the reference:
    Model(config, vocab); .cuda(); .parameters(); .state_dict() with the SAME key names;
    .iterate(pack, optimizer=..., is_train=...) -> (metrics Pack{loss, cer}, None);
    .forward(pack) -> Pack{pred, gold}; .cal_metrics(output, pack); .get_default_config()
so the reference's main.py / Trainer11 drive it unchanged.  What runs underneath is the explicit
forward/backward engine (engine.py) on hand-written HIP kernels - there is no autograd graph
and NO CPU fallback: iterate() on CPU tensors raises.

Additions required by BASELINE.json's north_star (not in the reference):
  * config.ctc_weight = lambda in [0,1]: joint loss lambda*CTC + (1-lambda)*CE via a `ctc_lo`
    head on the encoder output (lambda = 0 reproduces the reference exactly),
  * TransformerCTC: encoder + CTC only (BASELINE config 2),
  * config.cross_mask: "ref_compat" reproduces the reference's decoder cross-attention mask built
    from TEXT lengths (transformer_official.py:78, 301-303); "wave_len" is the corrected mask,
  * config.dtype: "bf16" (default) or "fp32" (exact-fp32 parity mode),
  * config.attn_window: +-w frame band on encoder self-attention (long-form config), -1 = full.
"""
import math

import torch

from .. import engine as E
from .. import kernels as K
from ..Bases import BaseConfig, BaseModel
from ..Utils import Pack

SOS_ID, EOS_ID, PAD_ID = 2, 3, 0   # transformer_official.py:53-54, Utils/loss.py:5
PE_MAXLEN = 5000                   # transformer_official.py:49, 65
CLIP_NORM = 5.0                    # transformer_official.py:102


def _set_nested(root, dotted, value, is_buffer=False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, torch.nn.Module())
        mod = getattr(mod, p)
    if is_buffer:
        mod.register_buffer(parts[-1], value)
    else:
        mod.register_parameter(parts[-1], value)


class _SpeechTransformer(BaseModel):
    USE_DECODER = True

    def __init__(self, config, vocab):
        super().__init__()
        self.config = config
        self.vocab = vocab
        c = config
        self.use_decoder = self.USE_DECODER
        lam = float(getattr(c, "ctc_weight", 0.0))
        self.ctc_weight = lam if self.use_decoder else 1.0
        self.use_ctc = (lam > 0.0) or not self.use_decoder
        self.cross_mask = getattr(c, "cross_mask", "ref_compat")
        self.lowp = str(getattr(c, "dtype", "bf16")).lower() in ("bf16", "bfloat16")
        self.attn_window = int(getattr(c, "attn_window", -1))
        self.label_smoothing = float(getattr(c, "label_smoothing", 0.0))   # Utils/loss.py:30-45 (the reference never enables it)
        self.cer_in_iterate = bool(getattr(c, "cer_in_iterate", True))
        self._step_seed = int(getattr(c, "seed", 0))   # advanced once per training step (dropout masks)
        d, H, dk, ff = c.d_model, c.num_head, c.hidden_size, c.ff_size
        d_in = c.n_mels * c.lfr_m
        V = vocab.vocab_size
        self.V = V

        # ---- parameter blocks in FORWARD order (backward finishes them in reverse: dist.py)
        blocks = [[("encoder.linear_in.weight", (d, d_in))], [("encoder.linear_in.bias", (d,))],
                  [("encoder.layer_norm_in.weight", (d,))], [("encoder.layer_norm_in.bias", (d,))]]
        for i in range(c.layer_num):
            blocks += E.mha_param_block(f"encoder.layer_stack.{i}.slf_attn.", H, dk, d)
            blocks += E.ffn_param_block(f"encoder.layer_stack.{i}.pos_ffn.", d, ff)
        if self.use_ctc:
            blocks += [[("ctc_lo.weight", (V, d))], [("ctc_lo.bias", (V,))]]
        if self.use_decoder:
            blocks += [[("decoder.tgt_word_emb.weight", (V, d))]]
            # the cross-attention K | V projections of ALL decoder layers side by side: they all multiply the same encoder output
            # (transformer_official.py:309-314, 446-458), so the six projections are one (L 2 H dk, d) matrix to the engine
            blocks += E.cross_kv_param_blocks([f"decoder.layer_stack.{i}.enc_attn." for i in range(c.layer_num)], H, dk, d)
            for i in range(c.layer_num):
                blocks += E.mha_param_block(f"decoder.layer_stack.{i}.slf_attn.", H, dk, d)
                blocks += E.cross_q_param_block(f"decoder.layer_stack.{i}.enc_attn.", H, dk, d)
                blocks += E.ffn_param_block(f"decoder.layer_stack.{i}.pos_ffn.", d, ff)
        self._flat = E.FlatParams(blocks)
        self._engine = None
        self._views_checked = False   # parameters verified to be views of the flat buffer
        self._grads_checked = False   # .grad attributes verified to be views of the flat gradient
        self._flat_device = None

        # ---- nn.Parameters with the reference's names / shapes / init distributions
        order = self._state_order(c.layer_num)
        for name in order:
            if name.endswith("positional_encoding.pe"):
                _set_nested(self, name, E.positional_encoding(PE_MAXLEN, d), is_buffer=True)
            elif name == "decoder.tgt_word_prj.weight":
                _set_nested(self, name, self.decoder.tgt_word_emb.weight)   # tied (transformer_official.py:253-256)
            else:
                wname = name[:-5] + ".weight" if name.endswith(".bias") else name
                wshape = self._flat.index[wname][1]
                _set_nested(self, name, torch.nn.Parameter(self._init_tensor(name, self._flat.index[name][1], wshape, d, dk)))

    # state_dict key order of the reference (module registration order)
    def _state_order(self, L):
        def mha(pre):
            return [pre + n for n in ("w_qs.weight", "w_qs.bias", "w_ks.weight", "w_ks.bias", "w_vs.weight", "w_vs.bias",
                                      "layer_norm.weight", "layer_norm.bias", "fc.weight", "fc.bias")]

        def ffn(pre):
            return [pre + n for n in ("w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias", "layer_norm.weight", "layer_norm.bias")]

        names = ["encoder.linear_in.weight", "encoder.linear_in.bias", "encoder.layer_norm_in.weight",
                 "encoder.layer_norm_in.bias", "encoder.positional_encoding.pe"]
        for i in range(L):
            names += mha(f"encoder.layer_stack.{i}.slf_attn.") + ffn(f"encoder.layer_stack.{i}.pos_ffn.")
        if self.use_decoder:
            names += ["decoder.tgt_word_emb.weight", "decoder.positional_encoding.pe"]
            for i in range(L):
                names += mha(f"decoder.layer_stack.{i}.slf_attn.") + mha(f"decoder.layer_stack.{i}.enc_attn.") + ffn(f"decoder.layer_stack.{i}.pos_ffn.")
            names += ["decoder.tgt_word_prj.weight"]
        if self.use_ctc:
            names += ["ctc_lo.weight", "ctc_lo.bias"]
        return names

    @staticmethod
    def _init_tensor(name, shape, wshape, d, dk):
        """attention.py:16-28, transformer_official.py:147-156, 242-256, torch defaults elsewhere."""
        t = torch.empty(*shape)
        leaf = name.rsplit(".", 2)[-2]
        if leaf in ("layer_norm", "layer_norm_in"):
            return t.fill_(1.0) if name.endswith(".weight") else t.zero_()
        fan_out, fan_in = wshape[0], wshape[1]
        if name == "decoder.tgt_word_emb.weight":
            return t.normal_(0.0, 1.0)            # nn.Embedding default; the tied projection reuses it
        if name.endswith(".weight"):
            if leaf in ("w_qs", "w_ks", "w_vs"):
                return t.normal_(0.0, math.sqrt(2.0 / (d + dk)))
            if leaf in ("fc", "linear_in", "ctc_lo"):
                return t.normal_(0.0, math.sqrt(2.0 / (fan_in + fan_out)))   # xavier_normal_
        bound = 1.0 / math.sqrt(fan_in)            # Linear / Conv1d default: U(+-1/sqrt(fan_in)), weights and biases
        return t.uniform_(-bound, bound)

    # ------------------------------------------------------------------ flat storage management
    def _named_flat_params(self):
        for name, p in self.named_parameters():
            yield name, p

    def _ensure_engine(self, device):
        """(Re)build the flat HBM buffers when the parameters are not views of them (first call,
        after .cuda()/.to(), after load_state_dict on a fresh module)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:   # "cuda" and "cuda:0" must compare equal below
            device = torch.device("cuda", torch.cuda.current_device())
        if device.type != "cuda":
            raise RuntimeError("the HIP engine runs on an MI355X only: move the model and batch to 'cuda' "
                               "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
        f = self._flat
        ok = self._engine is not None and self._flat_device == device
        if ok and self._views_checked:      # fast path of every step: nothing has touched the parameters' storage
            return self._engine
        if ok:
            for name, p in self._named_flat_params():
                off, shape = f.index[name]
                if p.data_ptr() != f.p.data_ptr() + 4 * off:
                    ok = False
                    break
        if ok:
            self._views_checked = True
            return self._engine
        K.bind_device(device)
        old = {name: p.detach().to(device=device, dtype=torch.float32) for name, p in self._named_flat_params()}
        old_mv = (f.m.to(device), f.v.to(device)) if f.m is not None else None   # keep Adam state across a device move
        f.allocate(device, self.lowp)
        if old_mv is not None:
            f.m.copy_(old_mv[0])
            f.v.copy_(old_mv[1])
        for name, p in self._named_flat_params():
            view = f.view(f.p, name)
            view.copy_(old[name].view(view.shape))
            p.data = view
            p.grad = f.view(f.g, name)
        f.refresh_lowp()
        for b in self.buffers():
            b.data = b.data.to(device)
        pe = self.encoder.positional_encoding.pe[0].to(device).contiguous()
        self._engine = E.Engine(f, self.config, self.V, self.use_decoder, self.use_ctc, pe)
        self._flat_device = device
        self._views_checked = True
        self._grads_checked = True
        return self._engine

    def zero_grad(self, set_to_none=True):
        self._grads_checked = False      # nn.Module.zero_grad may drop the .grad views
        return super().zero_grad(set_to_none=set_to_none)

    def _apply(self, fn, *a, **kw):
        # .cuda() / .to() / .float() replace the parameters' storage: re-validate the flat views on the next step
        self._views_checked = False
        self._grads_checked = False
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k: v for k, v in state_dict.items()}
        self._views_checked = False
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._flat.p is not None:
            self._flat.refresh_lowp()
        return out

    def zero_flat_grads(self):
        """Zero the flat gradient buffer for the step that follows.  With the multi-stream engine the fill (160 MB at the joint model's size,
        ~20 us) is not queued in front of the forward pass on the main stream: train_step hands it to the weight-gradient stream - idle during
        the forward pass - in front of the transposed weight copies, whose event every backward path waits for before its first gradient write."""
        f = self._flat
        eng = self._engine
        self._zero_lazy = eng is not None and eng._tr_tiles is not None and eng.overlap_wgrad
        if not self._zero_lazy:
            f.g.zero_()
        if self._grads_checked:             # fast path: .grad views were verified and nobody reset them
            return
        for name, p in self._named_flat_params():
            if p.grad is None or p.grad.data_ptr() != f.g.data_ptr() + 4 * f.index[name][0]:
                p.grad = f.view(f.g, name)
        self._grads_checked = True

    # ------------------------------------------------------------------ reference API
    def _prepare(self, input, training=False):
        wave = input.wave
        eng = self._ensure_engine(wave.device)
        eng.training = bool(training)
        if training:
            self._step_seed = (self._step_seed + 1) & 0x7FFFFFFF
            eng.step_seed = self._step_seed
        x = wave if wave.dtype == eng.dtype else wave.to(eng.dtype)
        x = x.contiguous()
        tgt = input.tgt_for_input.contiguous()
        # the batch contract hands lengths over as int64 (ai_shell_1.py:75-88): the label preprocessing launch also makes the int32 copies
        lens64 = [t.contiguous() for t in (input.wave_len, input.tgt_len) if t is not None and t.dtype == torch.int64 and t.device == tgt.device]
        out = K.dec_preprocess(tgt, SOS_ID, EOS_ID, lens64=lens64)
        prep, lens32 = out[:6], (out[6] if lens64 else ())
        wave_len = self._len32_of(input.wave_len, lens64, lens32)
        self._tgt_len32 = self._len32_of(input.tgt_len, lens64, lens32) if input.tgt_len is not None else None
        return eng, x, wave_len, prep

    @staticmethod
    def _len32_of(t, lens64, lens32):
        """int32 copy of a length vector: the one dec_preprocess made when the vector went through it, else a cast."""
        for a, b in zip(lens64, lens32):
            if a.data_ptr() == t.data_ptr() and a.numel() == t.numel():
                return b
        return t.to(torch.int32)

    def forward(self, input):
        """transformer_official.py:68-81 (inference-style forward; no gradients)."""
        eng, x, wave_len, prep = self._prepare(input)
        B, T, _ = x.shape
        enc, _ = eng.encoder_fwd(x, wave_len, self.attn_window)
        pack = Pack()
        pack.add(encoder_out=enc.view(B, T, -1))
        if self.use_decoder:
            cross_len = self._tgt_len32 if self.cross_mask == "ref_compat" else wave_len
            pred, _ = eng.decoder_fwd(prep, enc, cross_len, B, T)
            pack.add(pred=pred.view(B, -1, self.V), gold=prep[1].long())
        if self.use_ctc:
            logits = eng.ctc_lo.fwd(enc)
            pack.add(ctc_logits=logits.view(B, T, self.V))
        return pack

    def _tok_table(self, device):
        if getattr(self, "_tok_tab", None) is None or self._tok_tab[0].device != torch.device(device):
            self._tok_tab = K.token_table(self.vocab._id2token, device)
        return self._tok_tab

    def _cer_ids(self, ids, gold, hyp_len=None, ref_len=None):
        """CER in percent of a batch as a 1-element DEVICE tensor: strings, edit distance and the mean stay on
        the GPU (asr_cer) - the reference copies the ids to the host and loops over the batch every step
        (transformer_official.py:87-91, score.py:4-13)."""
        per = K.cer(ids.int().contiguous(), gold.int().contiguous(), self._tok_table(ids.device), self.vocab._token2id[self.vocab.PAD],
                    hyp_len=hyp_len, ref_len=ref_len)
        return (per.sum() * (100.0 / per.numel())).reshape(1)

    CER_BESIDE_BACKWARD = True      # False: score the step's CER on the main stream behind the optimizer (A/B, tests)

    def _cer_beside_backward(self, eng, pg):
        """The per-step CER of the greedy ids (five small launches, ~65 us behind the optimizer on the main stream: 1.3 % of the joint step) on
        the auxiliary stream instead, which is idle once the decoder's backward pass has joined it: it runs beside the encoder's backward pass,
        and iterate() makes the main stream wait for its event before handing the metric out.  Returns (cer, event); the ids stay referenced by
        the model until the next step (blocks of the main stream's pool read on another stream)."""
        if not (self.CER_BESIDE_BACKWARD and eng.aux_overlap) or torch.cuda.is_current_stream_capturing():
            return pg
        eng._disarm()      # this fork is for the auxiliary stream: it must not consume a hand-over meant for the weight-gradient stream
        eng._fork(eng.ctc_stream)
        with torch.cuda.stream(eng.ctc_stream):
            cer = self._cer_ids(pg[0], pg[1])
        if getattr(self, "_cer_event", None) is None:
            self._cer_event = torch.cuda.Event()
        self._cer_event.record(eng.ctc_stream)
        self._cer_keep = pg
        return cer, self._cer_event

    def _ctc_cer_beside_backward(self, eng, path, wave_len, labels32, lab_len):
        """CTC-only model: collapse the greedy path and score it against the label strings (as cal_metrics does for a CTC-only model), on the
        auxiliary stream beside the encoder's backward pass when the step runs on several streams.  Returns what _cer_of takes."""
        def score():
            ids, lens = K.ctc_collapse(path, wave_len, PAD_ID)
            return self._cer_ids(ids, labels32, hyp_len=lens, ref_len=lab_len)
        if not (self.CER_BESIDE_BACKWARD and eng.aux_overlap) or torch.cuda.is_current_stream_capturing():
            return score(), None
        eng._disarm()
        eng._fork(eng.ctc_stream)
        with torch.cuda.stream(eng.ctc_stream):
            cer = score()
        if getattr(self, "_cer_event", None) is None:
            self._cer_event = torch.cuda.Event()
        self._cer_event.record(eng.ctc_stream)
        self._cer_keep = (path, wave_len, labels32, lab_len)      # blocks of the main stream's pool read on another stream: alive until the next step
        return cer, self._cer_event

    def _cer_of(self, pg):
        """The CER tensor of what train_step returned: (cer, event) when it was computed on the auxiliary stream (the current stream then waits
        for the event), else (ids, gold) to be scored here."""
        if isinstance(pg[1], torch.cuda.Event):
            torch.cuda.current_stream().wait_event(pg[1])
            return pg[0]
        if pg[1] is None:      # already scored on the current stream
            return pg[0]
        return self._cer_ids(pg[0], pg[1])

    def _cer(self, pred, gold):
        """transformer_official.py:87-91.  Greedy ids by argmax (first index wins ties - the
        reference's topk(1) tie order at exactly-zero padded rows is implementation-defined)."""
        return self._cer_ids(pred.argmax(-1), gold)

    def cal_metrics(self, output, input):
        """transformer_official.py:83-94 on a forward() output (evaluation path)."""
        pack = Pack()
        row_nll = nll = n_valid = None
        if self.use_decoder:
            pred, gold = output.pred, output.gold
            B, To, V = pred.shape
            n_valid = (gold != PAD_ID).sum().float().reshape(1)
            row_nll, _ = K.xent_fwd_bwd(pred.reshape(B * To, V).contiguous(), gold.reshape(-1).int(), n_valid, PAD_ID, smoothing=self.label_smoothing, want_grad=False)
        if self.use_ctc:
            prep = K.dec_preprocess(input.tgt_for_input.contiguous(), SOS_ID, EOS_ID)
            nll, _ = K.ctc_fwd_bwd(output.ctc_logits.contiguous(), input.wave_len.to(torch.int32), prep[2], prep[4], self._engine.ws, want_grad=False)
        lam = self.ctc_weight
        loss = K.loss_combine(row_nll, n_valid, nll, 1.0 - lam if self.use_ctc else 1.0, lam)
        assert not torch.isinf(loss[0])
        pack.add(loss=loss[0])
        if self.use_decoder:
            pack.add(cer=self._cer(output.pred, output.gold))
        if self.use_ctc:
            # best-path CTC decoding on the device; the label strings (no sos/eos) are the reference
            cer = self._ctc_cer(output.ctc_logits.contiguous(), input.wave_len.to(torch.int32), prep[2], prep[4])
            pack.add(**({"ctc_cer": cer} if self.use_decoder else {"cer": cer}))
        return pack

    def beam_search(self, input, beam_size=5, nbest=1, decode_max_len=0, ctc_weight=0.0):
        """Attention-decoder beam search for a batch (Decoder.recognize_beam, transformer_official.py:
        331-434, batched on the GPU with key/value caches): per utterance a list of at most `nbest`
        {'yseq': [sos, ..., eos], 'score': float}.
        ctc_weight > 0 (joint models): the beam's hypotheses are re-ranked by ctc_weight * log p_ctc + (1 - ctc_weight) *
        log p_att (decode.joint_beam_search; entries then also carry 'att_score' and 'ctc_score')."""
        from .. import decode
        if ctc_weight > 0.0:
            return decode.joint_beam_search(self, input, beam_size, nbest, decode_max_len, ctc_weight)
        return decode.beam_search(self, input, beam_size, nbest, decode_max_len)

    def ctc_prefix_beam_search(self, input, beam_size=5, nbest=1, frame_topk=10, on_device=None):
        """CTC prefix beam search over the CTC head (decode.ctc_prefix_beam_search): per utterance at most `nbest`
        {'yseq': [ids], 'score': log p(yseq | x)}.  on_device: None = the device kernel when beam * (frame_topk + 1) <= 64."""
        from .. import decode
        return decode.ctc_prefix_beam_search(self, input, beam_size, nbest, frame_topk, on_device)

    def ctc_greedy_search(self, input):
        """Best-path CTC hypotheses of a batch: list of id lists (repeats merged, blanks removed)."""
        with torch.no_grad():
            out = self.forward(input)
        ids, lens = K.ctc_greedy_decode(out.ctc_logits.contiguous(), input.wave_len.to(torch.int32), PAD_ID)
        ids, lens = ids.cpu(), lens.cpu()
        return [ids[b, : int(lens[b])].tolist() for b in range(ids.shape[0])]

    def _ctc_cer(self, logits, wave_len, labels32, lab_len):
        ids, lens = K.ctc_greedy_decode(logits, wave_len, PAD_ID)
        return self._cer_ids(ids, labels32, hyp_len=lens, ref_len=lab_len)

    def train_step(self, input, loss_scale=1.0, count_hook=None):
        """Forward + backward into the flat gradient buffer (no optimizer).  Returns the metrics
        tensor [loss, ce, ctc] (device) and, for CER, (pred, gold) or None.
        count_hook (data parallelism, dist.DataParallel): called as count_hook(n_valid, B) right after the label
        preprocessing; it starts the all-reduce of [non-pad token count, batch size] and returns an object whose wait()
        yields the two GLOBAL values as 1-element device tensors - the loss kernels then normalise by those."""
        eng, x, wave_len, prep = self._prepare(input, training=self.training)
        B, T, _ = x.shape
        lam = self.ctc_weight
        ys_in, ys_out, labels32, dec_len, lab_len, n_valid = prep
        pending = count_hook(n_valid, B) if count_hook is not None else None
        zero, self._zero_lazy = (self._flat.g if getattr(self, "_zero_lazy", False) else None), False
        eng.refresh_transposes(zero=zero)      # W^T copies for this step's input-gradient GEMMs (side stream, beside the forward pass)
        enc, ecache = eng.encoder_fwd(x, wave_len, self.attn_window)
        batch_div = None
        if pending is not None:
            n_valid, batch_div = pending.wait()
        row_nll = nll = None
        d_enc = None
        pg = None
        ctc_done = None
        ctc_scale = dict(grad_scale=lam * loss_scale, grad_scale_div=batch_div) if batch_div is not None else dict(grad_scale=lam * loss_scale / float(B))
        ctc_async = (self.use_decoder and self.use_ctc and eng.overlap_ctc and not eng.deterministic and not torch.cuda.is_current_stream_capturing())
        if ctc_async:      # joint model: the CTC branch runs beside the decoder's forward pass
            eng.decoder_kv_async(prep, enc, B, T)      # ... behind the K|V projections the decoder's first cross-attention waits for
            nll, d_enc, ctc_done = eng.ctc_branch_async(enc, wave_len, labels32, lab_len, B, T, **ctc_scale)
        if self.use_decoder:
            cross_len = self._tgt_len32 if self.cross_mask == "ref_compat" else wave_len
            pred, dcache = eng.decoder_fwd(prep, enc, cross_len, B, T)
            # the greedy ids of the step's CER come out of the loss kernel (it reads every row anyway; the gradient overwrites the logits in place)
            ids = torch.empty(ys_out.shape, dtype=torch.int32, device=pred.device) if self.cer_in_iterate else None
            w_ce = (1.0 - lam) if self.use_ctc else 1.0
            row_nll, dpred = K.xent_fwd_bwd(pred, ys_out.reshape(-1), n_valid, PAD_ID, smoothing=self.label_smoothing, grad_scale=w_ce * loss_scale, dlogits=pred,
                                            argmax=ids)
            if ids is not None:
                pg = (ids, ys_out)
        if self.use_ctc and not ctc_async:
            # CTC-only model: the step's CER (the reference's trainer reads metrics.cer every step, Trainer/trainer11.py:73-75) is scored on the
            # greedy CTC path, which the loss kernels hand out (the gradient overwrites the logits in place)
            path = torch.empty(B, T, dtype=torch.int32, device=enc.device) if (self.cer_in_iterate and not self.use_decoder) else None
            nll, d_enc = eng.ctc_fwd_bwd(enc, wave_len, labels32, lab_len, B, T, best_path=path, **ctc_scale)
            if path is not None:
                pg = self._ctc_cer_beside_backward(eng, path, wave_len, labels32, lab_len)
        if self.use_decoder:
            if d_enc is None:
                d_enc = torch.zeros_like(enc)
            eng.decoder_bwd(dcache, dpred, d_enc, d_enc_ready=ctc_done)
            if pg is not None:
                pg = self._cer_beside_backward(eng, pg)
        eng.encoder_bwd(ecache, d_enc)
        loss = K.loss_combine(row_nll, n_valid, nll, (1.0 - lam) if self.use_ctc else 1.0, lam)
        return loss, pg

    def iterate(self, input, optimizer=None, is_train=True):
        """transformer_official.py:96-104: forward, metrics, and - when training - zero_grad,
        backward, clip_grad_norm_(5.0), optimizer.step()."""
        if optimizer is None or not is_train:
            with torch.no_grad():
                output = self.forward(input)
                return self.cal_metrics(output, input), None
        self._ensure_engine(input.wave.device)
        optimizer.zero_grad()
        if getattr(optimizer, "fused_step", None) is None or not getattr(optimizer, "keeps_grad_views", False):
            # a stock torch.optim optimizer (or the reference's own NoamOpt) sets every .grad to None in zero_grad():
            # re-attach the views of the flat gradient buffer, or clip / step would see no gradients at all
            self._grads_checked = False
        self.zero_flat_grads()
        loss, pg = self.train_step(input)
        fused = getattr(optimizer, "fused_step", None)
        if fused is not None:
            fused(self._flat, CLIP_NORM)                      # sumsq + clip + Noam + Adam, 3 launches
        else:                                                 # any torch.optim-style optimizer
            torch.nn.utils.clip_grad_norm_(self.parameters(), CLIP_NORM)
            optimizer.step()
            self._flat.refresh_lowp()
        metrics = Pack()
        metrics.add(loss=loss[0])
        if self.use_decoder and self.use_ctc:
            metrics.add(ce=loss[1], ctc=loss[2])
        if pg is not None:
            metrics.add(cer=self._cer_of(pg))                 # no device-to-host copy, no sync
        return metrics, None

    def greedy_search(self, input, decode_max_len=0):
        """transformer_official.py:106-107 is an empty stub in the reference; here: beam search with one beam."""
        return self.beam_search(input, 1, 1, decode_max_len)

    @classmethod
    def get_default_config(cls):
        class ModelConfig(BaseConfig):      # transformer_official.py:115-122 (+ the additions above)
            d_model = 512
            hidden_size = 64
            ff_size = 1024
            num_head = 8
            dropout = 0.1
            layer_num = 6
            share_weight = False
            ctc_weight = 0.0
            label_smoothing = 0.0
            cross_mask = "ref_compat"
            dtype = "bf16"
            attn_window = -1

        return ModelConfig


class TransformerOffical(_SpeechTransformer):
    """Encoder-decoder with CE (and CTC when config.ctc_weight > 0)."""
    USE_DECODER = True


class TransformerCTC(_SpeechTransformer):
    """Encoder + CTC head only (BASELINE.json config 2)."""
    USE_DECODER = False
