/*
 * asr_hip.h - C ABI of libasr_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the training
 * hot path of the Speech-Transformer of zqs01/ASR_chinese_e2e (+ the CTC branch that
 * BASELINE.json's north_star adds).
 *
 * The reference is pure Python on stock PyTorch ops and has NO native / FFI boundary of its own
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference op SEQUENCE
 * (file:line under the reference tree) that it replaces; the Python-side binding a maintainer
 * would add is shown in INTEGRATION.md (ctypes, mirrored in asr_chinese_e2e_amd/_lib.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No torch types, no exceptions.
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host.
 *   - no allocation, no synchronisation, no host<->device copy inside any entry point: each call
 *     only enqueues kernels on `stream` (a hipStream_t passed as void*), so every entry point can
 *     be captured into a hipGraph.  Scratch memory comes from the caller (`ws`, sized by the
 *     matching asr_*_workspace_bytes query).
 *   - return value: ASR_OK (0) or a negative ASR_E* code; asr_last_error() gives the per-thread
 *     message.  Stateless and re-entrant; ordering is stream order only.
 *   - activations are padded-dense row-major (B, T, d) == (B*T, d); utterance b owns rows
 *     [b*T, (b+1)*T) of which the first len[b] are valid.  All masks are derived in-kernel from
 *     int32 length vectors - the reference's materialised (B,Tq,Tk) bool masks
 *     (Predictor/Models/utils.py:100-144) never exist.
 *   - dtype: ASR_F32 or ASR_BF16 is the STORAGE type of activation tensors; all arithmetic
 *     accumulates in fp32.  Parameters, optimizer state, statistics and losses are fp32.
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_ABI_VERSION 10

typedef enum { ASR_F32 = 0, ASR_BF16 = 1 } asr_dtype_t;

#define ASR_OK 0
#define ASR_EINVAL (-1)     /* bad shape / null pointer / unsupported size */
#define ASR_EDTYPE (-2)     /* unsupported dtype for this op */
#define ASR_EWORKSPACE (-3) /* workspace too small */
#define ASR_EHIP (-4)       /* HIP runtime reported an error at launch */

#define ASR_ACT_NONE 0
#define ASR_ACT_RELU 1
#define ASR_ACT_RELU_MASK 2 /* C = (A W^T) where res > 0, else 0 (no bias: gradients have none) - the ReLU backward of
                               module.py:70-71 applied in the store tail of the input-gradient GEMM (res = the activations) */

int asr_abi_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int asr_last_error(char* buf, size_t n);
/* Deterministic mode (process-wide; initial value from the environment variable ASR_DETERMINISTIC).  The reference's
 * autograd on CPU sums every gradient in a fixed order (torch.autograd through transformer_official.py:100-103);
 * the default HIP path combines partial weight-gradient tiles, bias-gradient slices and embedding rows with fp32
 * atomics, whose arrival order varies from run to run.  With the mode on, every such reduction runs in a fixed
 * order: asr_gemm_tn_* write one partial slab per M-split into `ws` (asr_gemm_tn_workspace_bytes is then non-zero)
 * and add the slabs in split order, the column-sum finalisers use one workgroup per column group, asr_embed_bwd
 * adds the token rows in order, asr_gemm_tn_grouped_bf16 refuses.  Results are then bit-identical from run to run.
 * asr_set_deterministic returns the previous value. */
/* Stream ordering helper of the host runtime: work queued on `to_stream` after this call runs after the work queued on
 * `from_stream` before it (hipEventRecord + hipStreamWaitEvent on a pooled event).  The reference has one stream and
 * synchronises every step (trainer11.py:73-74); the engine runs weight gradients / the CTC branch / communication on side
 * streams and forks ~50 times per step.  Must not be called while either stream is being captured into a hipGraph. */
int asr_stream_fork(void* from_stream, void* to_stream);
/* Hand-over WITHOUT a marker packet in the producer's queue (ABI 7).  asr_stream_arm(from, to): the next entry point of this library
 * that supports it (asr_gemm_nt_bf16 on the loader / consumer kernel, asr_add_ln_bwd in its partial-sum form, the fused asr_sdpa_bwd,
 * asr_ctc_fwd_bwd with a gradient) launches its LAST kernel on `from` with the completion event bound to the dispatch packet itself
 * (hipExtLaunchKernelGGL) and makes `to` wait for it - hipEventRecord would put a barrier packet behind that kernel, and the next
 * kernel of the queue starts ~3.5 us later.  asr_stream_arm_pending() returns 1 (and clears the arm) when no launch took it: the
 * caller then uses asr_stream_fork.  One arm at a time; host-side state only. */
int asr_stream_arm(void* from_stream, void* to_stream);
int asr_stream_arm_pending(void);
/* A non-blocking stream created by the HIP runtime THIS library is bound to: priority < 0 = the lowest priority the device
 * offers (weight-gradient stream: off the critical path of the step), 0 = default, > 0 = the highest.  Lives as long as the
 * process.  (The reference has one stream, trainer11.py:73-74; torch.cuda.Stream offers no low priority.) */
int asr_stream_create(int priority, void** out_stream);
/* Tuning options: process-wide integer switches under which every value gives correct results, settable at run time.  ABI 8 keeps one:
 * "cu_limit" (> 0: the one-workgroup-per-CU kernels - persistent NT GEMM grid, weight-gradient M-splits - size their launches for this
 * many CUs instead of the device's; the engine sets it around the large launches that run beside the decoder's chain of small kernels;
 * 0 = the whole device).  (Rounds 2 - 3 carried switches between kernel variants here - store policy, tile shape, split plans; the variants
 * that lost their A/B left the library, see DESIGN.md section 4 "Tried".)  Initial value 0.
 * "tn_multi" (round 5; initial value 1): asr_gemm_tn_grouped_bf16 runs a group of problems over the same >= 4096 rows on the 128 x 128-tile
 * code of the single-problem kernel in one launch; 0 = always the 256 x 128-tile grouped kernel (A/B timing).
 * "sdpa_pair" (round 5; initial value 0): 1 = asr_sdpa_fwd without causal / band mask and without dropout on the kernel that takes both 32-query
 * blocks of a wave through one pass over the key tiles (same bits; measured slower, kept for A/B).
 * previous (may be NULL) receives the old value.  Unknown name: ASR_EINVAL. */
int asr_set_option(const char* name, int value, int* previous);
int asr_get_deterministic(void);
int asr_set_deterministic(int on);

/* ---------------------------------------------------------------------------------------------
 * Fused residual-add + LayerNorm (+ positional encoding) (+ pad-row zeroing).
 * Replaces:  layer_norm(fc(x)+residual) ; enc_output *= non_pad_mask
 *              Predictor/Models/attention.py:59-60, module.py:72-75,
 *              transformer_official.py:208, 211, 449, 453, 456
 *            layer_norm_in(linear_in(x)) + positional_encoding   transformer_official.py:175-177
 *   z = x + res (res may be NULL);  xhat = (z - mean) * rstd  (eps 1e-5, biased variance)
 *   y = xhat * gamma + beta  (+ pe[t] if pe != NULL)   ;   y = 0 for rows t >= lens[b] if lens
 * x, res, y, xhat: (B*T, d) `dtype`; xhat may alias x.  gamma, beta: (d) f32; pe: (>=T, d) f32;
 * rstd: (B*T) f32; lens: (B) int32 or NULL.  d <= 2048.
 * Dropout (drop_p > 0; reference sites attention.py:59, module.py:73, transformer_official.py:175):
 *   drop_mode ASR_DROP_PRE : x is dropped (x * keep / (1-p)) before the residual add;
 *   drop_mode ASR_DROP_POST: the output (after the PE add, before pad zeroing) is dropped.
 * The keep mask of element (row, col) is a counter hash of (row*d + col, drop_seed): forward and
 * backward regenerate it, nothing is stored (asr_dropout_mask materialises it for tests).
 */
#define ASR_DROP_PRE 1
#define ASR_DROP_POST 2
int asr_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                   const float* pe, const int32_t* lens, void* y, void* xhat, float* rstd,
                   int B, int T, int d, float drop_p, uint32_t drop_seed, int drop_mode, int dtype,
                   void* stream);

/* Backward of the above.  dy (+ dy2 if not NULL) is the gradient wrt y.
 *   g = (dy + dy2) * mask * gamma ;  dz = rstd * (g - mean(g) - xhat * mean(g * xhat))
 * dz: (B*T, d) `dtype` (gradient wrt x and wrt res).  Column sums over all rows are ACCUMULATED
 * (+=) into f32 vectors: dgamma += sum (dy*mask*xhat), dbeta += sum (dy*mask), and, if dbias is
 * not NULL, dbias += sum dz (bias gradient of the GEMM that produced x).
 * With ASR_DROP_PRE dropout the gradient wrt x differs from the residual gradient: dz stays the
 * residual gradient and dx (required then, (B*T, d) `dtype`) receives dz * keep / (1-p); dbias sums
 * dx.  With ASR_DROP_POST the mask is applied to (dy + dy2) first.  dx may be NULL otherwise.
 * ws: asr_add_ln_bwd_workspace_bytes(B*T, d) bytes of scratch.
 */
size_t asr_add_ln_bwd_workspace_bytes(int rows, int d);
int asr_add_ln_bwd(const void* dy, const void* dy2, const void* xhat, const float* rstd,
                   const float* gamma, const int32_t* lens, void* dz, void* dx, float* dgamma,
                   float* dbeta, float* dbias, void* ws, size_t ws_bytes, int B, int T, int d,
                   float drop_p, uint32_t drop_seed, int drop_mode, int dtype, void* stream);
/* With dgamma == dbeta == NULL asr_add_ln_bwd leaves the per-workgroup partial sums of the parameter gradients in
 * `ws` (which the caller then keeps); this call adds the partial sums of up to ASR_LN_REDUCE_MAX such sites into
 * their dgamma / dbeta / dbias in ONE launch (the 13 LayerNorm sites of an encoder backward would otherwise cost
 * 13 small launches).  `items` is a host array, read during the call; rows = B*T of the site's asr_add_ln_bwd. */
#define ASR_LN_REDUCE_MAX 16
typedef struct asr_ln_reduce_item {
    const void* ws;
    float* dgamma;
    float* dbeta;
    float* dbias; /* or NULL */
    int rows;
} asr_ln_reduce_item;
int asr_add_ln_bwd_reduce_batched(const asr_ln_reduce_item* items, int n, int d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Masked scaled-dot-product attention, flash style (scores never materialised).
 * Replaces:  bmm(q,k^T)/temperature -> masked_fill(mask,-inf) -> softmax -> bmm(attn,v)
 *              Predictor/Models/attention.py:76-84, with the head split/merge permutes of
 *              attention.py:43-57 folded into strided addressing, and the masks of
 *              utils.py:100-144 / transformer_official.py:292-303 computed from lengths.
 * q: rows b*Tq+t, k/v: rows b*Tk+t; head h lives at columns [h*dk, (h+1)*dk) of a row whose
 * stride is ldq / ldk / ldv / ldo ELEMENTS (so q,k,v can point into one fused QKV buffer).
 * Key j is visible to query i of utterance b iff  j < k_len[b]  and (!causal or j <= i)
 * and (window < 0 or |i - j| <= window).  lse: (B, H, Tq) f32 = log sum exp of scaled scores.
 * ASR_BF16 runs on MFMA (dk must be 64); ASR_F32 is an exact-fp32 VALU path (dk <= 128).
 * drop_p > 0: dropout on the attention probabilities after the softmax (attention.py:83): the
 * output uses p * keep / (1-p), the normaliser does not.  Mask of (b,h,q,k) = counter hash of
 * (((b*H+h)*Tq+q)*Tk_even + k, drop_seed), Tk_even = Tk rounded up to even.
 * o_lo (ABI 10; ASR_BF16 only, may be NULL): a second buffer of o's layout that receives the LOW-ORDER piece of the output,
 * bf16(x - float(bf16(x))) of the fp32 value x whose bf16 rounding is stored in o.  Nothing but asr_sdpa_bwd reads it: with o alone
 * delta = rowsum(do * o) carries 2^-9 |o| of rounding per coordinate, which dq = sum_j p_j (dp_j - delta) k_j multiplies by the MEAN key
 * (the reference's fp32 softmax backward has no such term): measured 0.989 instead of >= 0.9995 gradient cosine of the top encoder
 * layer's Q / K projections at the full-size configuration.  The single-pass backward kernel (Tk <= 512) has its own remedy (centred keys,
 * dK's mean over the keys removed) and ignores the piece; the band form (Tk > 512 inside a window) and the two-kernel path read it.
 * The forward pass WRITES the piece on the tiled path (Tk > 512) and clears it everywhere else.
 */
int asr_sdpa_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                 const int32_t* k_len, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk,
                 int ldv, int ldo, int causal, int window, float scale, float drop_p,
                 uint32_t drop_seed, void* o_lo, int dtype, void* stream);

/* Backward: given do (same layout as o) computes dq, dk, dv (layouts/strides of q, k, v).
 * delta: f32 scratch of delta_bytes >= asr_sdpa_bwd_workspace_bytes(...) bytes, written by the call: (B, H, Tq) row sums
 * rowsum(do*o) for the two-kernel paths, or - bf16 self-attention inside a +-window band over more than 512 keys (the long-form
 * configuration, T = 2000: the band precedent is Predictor/Models/transformer_new.py:53) - the fp32 dQ partials of the query tiles
 * that straddle a 512-key block boundary of the single-pass band kernel.  With a smaller scratch (>= B*H*Tq floats) the band shapes
 * run on the two-kernel path. */
size_t asr_sdpa_bwd_workspace_bytes(int B, int H, int Tq, int Tk, int dk, int causal, int window, int dtype);
int asr_sdpa_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                 const float* lse, float* delta, size_t delta_bytes, void* dq, void* dk_, void* dv,
                 const int32_t* k_len, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk,
                 int ldv, int ldo, int causal, int window, float scale, float drop_p,
                 uint32_t drop_seed, const void* o_lo, int dtype, void* stream);      /* o_lo: what asr_sdpa_fwd wrote there, or NULL */

/* Test helpers: materialise the keep masks the kernels regenerate (1 = kept), uint8.
 * asr_dropout_mask: (rows, cols) mask of the LayerNorm / embedding sites;
 * asr_sdpa_dropout_mask: (B, H, Tq, Tk) mask of the attention-probability site. */
int asr_dropout_mask(uint8_t* mask, int rows, int cols, float drop_p, uint32_t drop_seed, void* stream);
int asr_sdpa_dropout_mask(uint8_t* mask, int B, int H, int Tq, int Tk, float drop_p, uint32_t drop_seed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CTC loss, forward-backward, fused with log-softmax over the vocabulary.
 * NOT in the reference (it has only cross-entropy, Predictor/Utils/loss.py:26-51); required by
 * BASELINE.json north_star.  Semantics = torch.nn.functional.ctc_loss(log_softmax(logits), ...,
 * blank, reduction='none') and its gradient wrt logits (ATen native/LossCTC.cpp).
 * logits: (B, T, V) `dtype`, rows `ld` elements apart (ABI 6: ld = V for a dense tensor; the training engine pads rows to a
 * multiple of 64 elements so that every row starts on a 128-byte line - with V = 4232 the head GEMM that writes them is 14 %
 * faster; columns V .. ld are never read or written);  in_len: (B) int32 frames per utterance (<= T);
 * labels: (B, Lmax) int32 padded; lab_len: (B) int32 (<= Lmax <= 255).
 * nll: (B) f32 = -log p(labels | x) (+inf when infeasible; 0 if zero_infinity).
 * dlogits: (B, T, V) `dtype`, same row stride (may alias logits) = scale * d(sum_b nll_b)/dlogits,
 * rows t >= in_len[b] are 0.  If dlogits is NULL only nll is computed.
 * scale = grad_scale, or grad_scale / *grad_scale_div when grad_scale_div (a DEVICE f32 scalar) is not NULL: under data
 * parallelism the CTC term is normalised by the GLOBAL batch, which arrives from an all-reduce on the device - the
 * host never waits for it (ABI 3; the cross-entropy kernel takes its token count the same way).
 * best_path (ABI 9; may be NULL): (B, T) int32, the frame-wise argmax of the logits (first index on ties, `blank` for frames
 * t >= in_len[b]) - the greedy CTC path, taken by the kernel that holds the row anyway.  With dlogits aliasing logits the
 * logits are gone after the call; the training step's per-batch CER (the reference's trainer reads metrics.cer every step,
 * Trainer/trainer11.py:73-75, computed by transformer_official.py:83-94) collapses this path with asr_ctc_collapse.
 */
size_t asr_ctc_workspace_bytes(int B, int T, int Lmax);
int asr_ctc_fwd_bwd(const void* logits, void* dlogits, const int32_t* in_len,
                    const int32_t* labels, const int32_t* lab_len, float* nll, int B, int T, int V, int ld,
                    int Lmax, int blank, float grad_scale, const float* grad_scale_div, int zero_infinity,
                    int32_t* best_path, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Greedy CTC decoding: per-frame argmax over the vocabulary (first index wins ties, as
 * torch.argmax), frames t >= in_len[b] count as blank, then the CTC collapse (merge repeats, drop
 * blanks).  NOT in the reference (no CTC there; its decoder-side search is the Python beam loop
 * transformer_official.py:331-434) - SURVEY.md 8(f) rank 1.
 * logits: (B, T, V) `dtype`, rows `ld` elements apart (ABI 6; V for a dense tensor); out_ids: (B, T) int32 = collapsed label ids,
 * 0-padded; out_len: (B).
 * ABI 9: the two halves are entry points of their own - asr_ctc_frame_argmax writes the frame-wise best path (B, T), asr_ctc_collapse
 * collapses a path in place (ids: (B, T) -> collapsed ids, 0-padded; out_len) - so that the training step can collapse the path
 * asr_ctc_fwd_bwd hands out (best_path) without a second pass over the logits.
 */
int asr_ctc_greedy_decode(const void* logits, const int32_t* in_len, int32_t* out_ids,
                          int32_t* out_len, int B, int T, int V, int ld, int blank, int dtype, void* stream);
int asr_ctc_frame_argmax(const void* logits, const int32_t* in_len, int32_t* path, int B, int T, int V, int ld, int blank,
                         int dtype, void* stream);
int asr_ctc_collapse(int32_t* ids, const int32_t* in_len, int32_t* out_len, int B, int T, int blank, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Beam search of the attention decoder, batched over utterances x beams
 * with key/value caches.  Replaces Decoder.recognize_beam, transformer_official.py:331-434, which
 * re-runs the whole decoder over the growing prefix for every hypothesis at every step in a
 * Python loop, one utterance at a time.  Same search: every live hypothesis is extended by its
 * `beam` best tokens (log_softmax scores, no length normalisation), the best `beam` extensions
 * survive (stable order on ties), hypotheses that emit eos leave the beam, at step maxlen-1 eos is
 * appended to every survivor.
 *
 * asr_decode_attn: single-query attention.  q: (R, H*dk) rows of stride ldq; keys/values of row r
 *   are rows (r / kv_div) * Tk_cap + t, t < len, of k / v (strides ldk / ldv, head h at columns
 *   [h*dk, (h+1)*dk)); len = k_len[r / len_div] if k_len else k_len_uniform.  No mask other than the
 *   length (recognize_beam passes dec_enc_attn_mask=None and a causal mask that only hides the
 *   future, which a cache never contains).  o: (R, H*dk), stride ldo.
 * asr_logsoftmax_topk: vals/ids (R, beam) = the `beam` largest log_softmax(logits[r]) entries,
 *   descending, ties by ascending index (transformer_official.py:381-384).
 * asr_beam_step: one search step for B utterances (beam <= 8).  In/out state score, alive,
 *   last_tok, parent: (B, beam).  Records of step `step` at [(step*B + b)*beam + slot]: rec_tok,
 *   rec_par (slot of the parent at the previous step), rec_end (0 live, 1 ended by its own eos,
 *   2 eos appended at the last step), rec_score.  maxlen: (B) int32 steps allowed per utterance.
 *   *alive_total += number of hypotheses still live after the step.
 * asr_cache_gather: dst[l][r][t] = src[l][(r / beam) * beam + parent[r]][t] for t < n_pos, with
 *   L caches of R rows x Lcap positions x row_bytes bytes (multiple of 16).
 */
/* asr_ctc_frame_topk: per frame (row) of the CTC head's logits, the k largest log_softmax entries (as asr_logsoftmax_topk) and the
 *   log_softmax of the blank class: the per-frame candidates of CTC prefix beam search (the reference has
 *   no CTC and leaves greedy_search / beam_search as empty stubs, transformer_official.py:106-110).  The prefix bookkeeping
 *   (merging paths that spell the same prefix) runs over these k candidates per frame.
 * asr_ctc_prefix_beam (ABI 4): that bookkeeping on the device - CTC prefix beam search (Hannun et al. 2014, algorithm 1 without a
 *   language model) for B utterances, one wave per utterance.  vals / ids: (B*T, k) from asr_ctc_frame_topk, blank_lp: (B*T),
 *   in_len: (B) frames per utterance (NULL = T).  The beam's prefixes are nodes of a trie in `ws`
 *   (asr_ctc_prefix_beam_workspace_bytes); beam * (k + 1) <= 64 (one wave ranks a frame's candidates), nbest <= beam.
 *   Out, per utterance and rank r < nbest, best first: out_tok[(b*nbest + r)*Lcap ..] the prefix (its first Lcap tokens),
 *   out_len its length (-1: fewer than nbest prefixes have non-zero probability), out_score log p(prefix | x) summed over
 *   alignments (fp64 inside).  Same results as the host restatement oracle/decode_ref.py::ctc_prefix_beam_search with the same
 *   per-frame candidates. */
int asr_ctc_frame_topk(const void* logits, float* vals, int32_t* ids, float* blank_lp, int R, int V, int ld, int k,
                       int blank, int dtype, void* stream);
size_t asr_ctc_prefix_beam_workspace_bytes(int B, int T, int beam);
int asr_ctc_prefix_beam(const float* vals, const int32_t* ids, const float* blank_lp, const int32_t* in_len, void* ws, size_t ws_bytes,
                        int32_t* out_tok, int32_t* out_len, float* out_score, int B, int T, int k, int beam, int nbest, int Lcap,
                        int blank, void* stream);
int asr_decode_attn(const void* q, const void* k, const void* v, void* o, const int32_t* k_len,
                    int k_len_uniform, int len_div, int R, int H, int dk, int Tk_cap, int kv_div,
                    int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
int asr_logsoftmax_topk(const void* logits, float* vals, int32_t* ids, int R, int V, int ld,
                        int beam, int dtype, void* stream);
int asr_beam_step(const float* top_vals, const int32_t* top_ids, float* score, int32_t* alive,
                  int32_t* last_tok, int32_t* parent, int32_t* rec_tok, int32_t* rec_par,
                  int32_t* rec_end, float* rec_score, const int32_t* maxlen, int32_t* alive_total,
                  int B, int beam, int step, int eos, void* stream);
int asr_cache_gather(const void* src, void* dst, const int32_t* parent, int L, int R, int beam,
                     int Lcap, int n_pos, int row_bytes, void* stream);

/* Character error rate per utterance on the device.
 * Replaces: calculate_cer (Predictor/Utils/score.py:4-13) over Vocab.convert_id2str strings
 *           (data_handler/vocab.py:75-79), called per step from cal_metrics
 *           (transformer_official.py:87-91) after a device-to-host copy of the greedy ids.
 * Convention kept: ids equal to pad_id are dropped, the token strings are joined by ONE space, the edit
 * distance runs over code points (spaces included) and is divided by (spaces in the reference string + 1).
 * hyp (B, Lh) ldh, ref (B, Lr) ldr: int32 ids; hyp_len / ref_len (B) or NULL = whole rows.
 * tok_cp: code points of all V token strings back to back; tok_off (V + 1): start of each;
 * max_tok_len: longest token string.  per_utt (B) f32 = distance / words of each utterance. */
int asr_cer(const int32_t* hyp, const int32_t* hyp_len, int Lh, int ldh, const int32_t* ref,
            const int32_t* ref_len, int Lr, int ldr, const int32_t* tok_cp, const int32_t* tok_off,
            int V, int max_tok_len, int pad_id, int B, float* per_utt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Softmax cross-entropy with ignore_index, forward + gradient in one pass over the logits.
 * Replaces:  F.cross_entropy(pred, gold, ignore_index=0, reduction='mean')  Utils/loss.py:47-49
 *            (+ label smoothing branch Utils/loss.py:30-45 when smoothing > 0)
 * logits: (M, V) `dtype`; gold: (M) int32.  row_nll: (M) f32 per-row loss (0 for ignored rows).
 * dlogits (may alias logits, may be NULL) = grad_scale / n_valid * d(sum row_nll)/dlogits where
 * n_valid is read from device memory (*n_valid, f32, e.g. written by asr_dec_preprocess).
 * argmax_out (ABI 9; may be NULL): (M) int32, the greedy class of EVERY row (ignored ones included; first index on ties) - the ids of
 * cal_metrics' pred.topk(1) (transformer_official.py:87-91), taken by the pass that reads the row anyway (with dlogits aliasing logits
 * the logits are gone after the call).
 */
int asr_xent_fwd_bwd(const void* logits, const int32_t* gold, const float* n_valid,
                     float* row_nll, void* dlogits, int M, int V, int ignore_index,
                     float smoothing, float grad_scale, int32_t* argmax_out, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder target preparation on device.
 * Replaces:  Decoder.preprocess (python list loop) transformer_official.py:260-275
 * tgt: (B, Lmax) int64 zero-padded label ids.  Writes (B, Lmax+1) int32:
 *   ys_in  = [sos, y..., eos-padding],  ys_out = [y..., eos, 0-padding]
 * dec_len[b] = 1 + #nonzero(tgt[b]) (rows of ys_in that are not eos padding),
 * lab_len[b] = #nonzero(tgt[b]), labels32 = compacted labels (B, Lmax) int32 (for CTC),
 * *n_valid = number of non-zero entries of ys_out (f32).
 * len_a / len_b (ABI 8, each may be NULL): (B) int64 length vectors of the batch contract (wave_len, tgt_len: collat.__call__
 * ai_shell_1.py:75-88 hands them over as int64) copied to len32_a / len32_b as the int32 the kernels take, in the same launch.
 */
int asr_dec_preprocess(const int64_t* tgt, int32_t* ys_in, int32_t* ys_out, int32_t* labels32,
                       int32_t* dec_len, int32_t* lab_len, float* n_valid, int B, int Lmax, int sos, int eos,
                       const int64_t* len_a, int32_t* len32_a, const int64_t* len_b, int32_t* len32_b, void* stream);

/* Embedding gather * scale + positional encoding.
 * Replaces:  tgt_word_emb(ys_in) * x_logit_scale + positional_encoding
 *              transformer_official.py:306-307
 * ids: (B*To) int32; emb: (V, d) f32 master embedding; pe: (>=To, d) f32; y: (B*To, d) `dtype`.
 * drop_p > 0: dropout on the output (transformer_official.py:306), mask as in asr_dropout_mask. */
int asr_embed_pe_fwd(const int32_t* ids, const void* emb, const float* pe, void* y, float scale,
                     int B, int To, int d, int V, float drop_p, uint32_t drop_seed, int dtype,
                     void* stream);
/* demb (V, d) f32 += scale * scatter-add over rows of ((dy + dy2) * keep / (1-p)) ((B*To, d) `dtype`; dy2 may be NULL: ABI 9 - the
 * gradient wrt the first decoder layer's input arrives as a (projection path, residual path) pair, added here in fp32). */
int asr_embed_bwd(const int32_t* ids, const void* dy, const void* dy2, float* demb, float scale, int rows, int d,
                  int V, float drop_p, uint32_t drop_seed, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise / reductions used between GEMMs.
 */
/* x = relu(x) in place, (n) `dtype`.  Replaces F.relu in module.py:70. */
int asr_relu_fwd(void* x, size_t n, int dtype, void* stream);
/* da = da * (a > 0) in place over (rows, cols); if dbias != NULL also dbias (cols) f32 +=
 * column sums of the masked da.  ws: asr_colsum_workspace_bytes(rows, cols). */
int asr_relu_bwd(void* da, const void* a, float* dbias, void* ws, size_t ws_bytes, int rows,
                 int cols, int dtype, void* stream);
/* out (cols) f32 (+)= column sums of x (rows, cols) with row stride ld elements. */
size_t asr_colsum_workspace_bytes(int rows, int cols);
int asr_colsum(const void* x, float* out, void* ws, size_t ws_bytes, int rows, int cols, int ld,
               int accumulate, int dtype, void* stream);
/* dst (n) `dst_dtype` = src (n) `src_dtype`  (f32 <-> bf16 conversion / copy) */
int asr_cast(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, void* stream);
/* Transposed copies of many matrices that live in one flat bf16 buffer, one launch: for tile t,
 * tiles[6t..6t+5] = {element offset of its matrix in src, rows N, cols K, (tile row << 16) | tile col (64 x 64 tiles),
 * element offset of the copy in dst, row stride ldd >= N of the copy} (ABI 6: the copy has its own offset and stride, so that
 * the CTC head's W^T - rows of V = 4232 elements - can be padded to whole 128-byte lines);
 * dst[dst_off + k * ldd + n] = src[src_off + n * K + k].  Used for the W^T copies the input-gradient GEMMs
 * (dX = dY W as an NT product, replacing autograd's mm backward of nn.Linear, attention.py:43-59, module.py:70-71) read. */
int asr_transpose_batched_bf16(const void* src, void* dst, const int32_t* tiles, int ntiles, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimizer over the flat parameter / gradient buffers.
 * Replaces:  clip_grad_norm_(params, 5.0) ; NoamOpt.step() -> Adam.step()
 *              transformer_official.py:102-103, Trainer/optimizer.py:15-28, main.py:81-83
 * asr_grad_sumsq: *sumsq (f32) = sum g^2 over n f32 elements (ws: asr_sumsq_workspace_bytes(n)).
 * asr_noam_hyper:  step += 1 (device int32); hyper[0] = lr = factor * model_size^-0.5 *
 *   min(step^-0.5, step*warmup^-1.5) (or lr_const if warmup <= 0); hyper[1] = 1 - b1^step;
 *   hyper[2] = sqrt(1 - b2^step).
 * asr_adam_step: coef = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (no clipping if max_norm <= 0);
 *   g' = g*coef; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 *   p -= lr/bc1 * m / (sqrt(v)/bc2s + eps); if p_lp != NULL the bf16 shadow copy is refreshed.
 *   If write_clipped != 0 the clipped gradient is written back to g (clip_grad_norm_ is in-place).
 */
size_t asr_sumsq_workspace_bytes(size_t n);
int asr_grad_sumsq(const float* g, size_t n, float* sumsq, void* ws, size_t ws_bytes, void* stream);
int asr_noam_hyper(int32_t* step, float* hyper, float model_size, float warmup, float factor,
                   float lr_const, float b1, float b2, void* stream);
/* asr_grad_sumsq followed by asr_noam_hyper in two launches instead of three (ABI 8): the single-workgroup finalizer of the squared norm
 * also advances the device-side step counter and writes the step's hyper-parameters. */
int asr_grad_sumsq_noam(const float* g, size_t n, float* sumsq, void* ws, size_t ws_bytes, int32_t* step, float* hyper, float model_size,
                        float warmup, float factor, float lr_const, float b1, float b2, void* stream);
int asr_adam_step(float* p, float* g, float* m, float* v, void* p_lp, size_t n,
                  const float* hyper, const float* sumsq, float max_norm, float b1, float b2,
                  float eps, int write_clipped, void* stream);
/* loss[0] = w_ce * sum(row_nll[0..M)) / *n_valid + w_ctc * sum(nll[0..B)) / B ;
 * loss[1] = the CE term, loss[2] = the CTC term (either input may be NULL with weight 0). */
int asr_loss_combine(const float* row_nll, int M, const float* n_valid, const float* nll, int B,
                     float w_ce, float w_ctc, float* loss, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense projections on MFMA:  C = act(A * W^T + bias)  ("NT": both operands K-contiguous).
 * Replaces:  nn.Linear / Conv1d(k=1) forward  attention.py:43-45, 59; module.py:70;
 *            transformer_official.py:176, 321  (and, with transposed operands, their dgrad).
 * A: (M, K) lda, W: (N, K) ldb, C: (M, N) ldc, all bf16; bias: (N) f32 or NULL;
 * if res != NULL, C += res (M, N) ldc (used to accumulate the residual gradient).
 */
int asr_gemm_nt_bf16(const void* A, const void* W, const float* bias, const void* res, void* C,
                     int M, int N, int K, int lda, int ldb, int ldc, int act, void* stream);
/* Small-M form of the projections (the decoder's B*To ~ 550 rows; transformer_official.py:446-458 through attention.py:43-59 and
 * module.py:70-71): 64 x 64 tiles, many short workgroups instead of a dozen long ones.
 *   trans_b = 0: C (M, N) = act(A (M, K) * Bm (N, K)^T + bias)        forward, Bm = the weight as stored
 *   trans_b = 1: C (M, N) = A (M, K) * Bm (K, N) (+ bias)             input gradient dx = dy W, Bm = the SAME weight (no transposed copy)
 * act: ASR_ACT_NONE / ASR_ACT_RELU / ASR_ACT_RELU_MASK (C = 0 where mask <= 0; mask (M, N) ldc bf16 = the activations of the ReLU whose
 * backward this is).  All bf16, fp32 accumulation; N, K, lda, ldb multiples of 8. */
int asr_gemm_small_bf16(const void* A, const void* Bm, const float* bias, const void* mask, void* C, int M, int N, int K,
                        int lda, int ldb, int ldc, int trans_b, int act, void* stream);
/* fp32 projections on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains): every GEMM of the PARITY mode (dtype fp32),
 * which is what the tests compare with the reference's own CPU outputs - nn.Linear / Conv1d(k=1) forward, input gradient and weight
 * gradient of attention.py:43-45,59, module.py:70-71, transformer_official.py:176,321 - and the odd shapes the bf16 kernels refuse.
 *   C (M, N) ldc  (+)=  act( opA (M, K) * opB (K, N) + bias (N) )
 *   trans_a = 0: A stored (M, K) lda;  1: A stored (K, M) lda      trans_b = 0: B stored (K, N) ldb;  1: B stored (N, K) ldb
 *   forward y = x W^T + b: (0, 1);  input gradient dx = dy W: (0, 0);  weight gradient dW += dy^T x: (1, 0) with accumulate = 1.
 * act: ASR_ACT_NONE / ASR_ACT_RELU / ASR_ACT_RELU_MASK (C = 0 where mask <= 0; mask (M, N) ldc f32); accumulate != 0: C += result.
 * The reduction is never split across workgroups: results are deterministic (same bits every run) in either mode. */
int asr_gemm_f32(const float* A, const float* B, const float* bias, const float* mask, float* C, int M, int N, int K,
                 int lda, int ldb, int ldc, int trans_a, int trans_b, int act, int accumulate, void* stream);
/* Weight gradient  dW (N, K) f32 (+)= dY^T (M, N)^T * X (M, K)   ("TN": reduction over rows).
 * ws: NULL / 0 in the default mode; in deterministic mode a 16-byte aligned buffer of
 * asr_gemm_tn_workspace_bytes(M, N, K) bytes (partial slabs, one per M-split). */
size_t asr_gemm_tn_workspace_bytes(int M, int N, int K);
int asr_gemm_tn_bf16(const void* dY, const void* X, float* dW, int M, int N, int K, int ldy,
                     int ldx, int ldw, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same, and dbias (N) f32 += column sums of dY (the bias gradient of the projection), taken
 * from the dY tiles the kernel stages anyway - no separate pass over dY.  dbias == NULL: as above. */
int asr_gemm_tn_bias_bf16(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K,
                          int ldy, int ldx, int ldw, int accumulate, void* ws, size_t ws_bytes,
                          void* stream);
/* Grouped form: nprob (<= 8) independent weight gradients dW_p (+)= dY_p^T X_p (+ bias gradients
 * where dbias != NULL) in ONE launch - e.g. all projections of a Transformer layer, whose
 * autograd weight-gradient GEMMs the reference runs one by one (torch.autograd through
 * attention.py:43-59, module.py:70-71).  Together the problems fill the GPU with larger tiles and
 * fewer splits of the M = B*T reduction than each would alone.  `probs` is a HOST array, read
 * during the call.  Per problem: dY (M, N) ldy, X (M, K) ldx bf16; dW (N, K) ldw f32.
 * Round 5: up to four problems over the SAME M >= 4096 rows whose 128 x 128 tiles fill the device with a few M-splits (the two projections
 * of a feed-forward or attention block: 64 tiles x 4 splits) run on the tile code of asr_gemm_tn_bias_bf16 in one launch - every workgroup
 * of either form ends by adding its fp32 tile to memory with atomics, so one launch for two problems halves that traffic; other groups
 * (different M per problem: a decoder layer's) take the 256 x 128-tile kernel as before. */
typedef struct asr_tn_problem {
    const void* dY;
    const void* X;
    float* dW;
    float* dbias;
    int M, N, K, ldy, ldx, ldw;
} asr_tn_problem;
int asr_gemm_tn_grouped_bf16(const asr_tn_problem* probs, int nprob, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One decoder layer as ONE host call: the fixed launch sequence of DecoderLayer.forward (transformer_official.py:446-458: self-attention,
 * encoder-decoder attention, position-wise FFN; each with post-LayerNorm and pad zeroing, attention.py:33-62, module.py:68-75) and of
 * autograd through it, issued from C++ on caller-owned buffers.  The decoder's ~23 kernels per layer and direction take 4 - 13 us each on
 * its B*To ~ 550 rows: with one foreign call per kernel from Python the host paced that part of the joint step.  Same kernels, same order,
 * same results as calling asr_gemm_small_bf16 / asr_sdpa_* / asr_add_ln_* / asr_gemm_nt_bf16 one by one.  bf16 activations only
 * (M = B*To rows; d, H*dk, ff multiples of 8).  All pointers are device pointers unless stated; the struct itself is HOST memory, read
 * during the call.  Not included (they stay with the caller): weight-gradient GEMMs (the dY / X operands are the buffers below), the
 * reduction of the LayerNorm parameter-gradient partial sums left in part_* (asr_add_ln_bwd_reduce_batched), data-parallel marks. */
typedef struct asr_dec_layer_plan {
    int B, To, T, d, H, dk, ff;           /* utterances, target positions, encoder frames per utterance, model width, heads, head dim, FFN width */
    float drop_p;                          /* dropout of every site of the layer (0 = off) */
    uint32_t seed[5];                      /* per-step mask seeds: self-attn probabilities, after self fc, cross-attn probabilities, after cross fc, after w_2 */
    int ld_kv_c_T;                         /* row stride (elements) of w_kv_c_T */
    const int32_t* dec_len;                /* (B) valid target positions (self-attention keys, pad zeroing) */
    const int32_t* cross_len;              /* (B) visible encoder frames of the encoder-decoder attention */
    /* parameters: bf16 weights as stored (out, in), f32 biases and LayerNorm gains / biases */
    const void *w_qkv_s, *w_fc_s, *w_q_c, *w_fc_c, *w_1, *w_2;
    const float *b_qkv_s, *b_fc_s, *b_q_c, *b_fc_c, *b_1, *b_2;
    const float *g_s, *be_s, *g_c, *be_c, *g_f, *be_f;
    const void* w_kv_c_T;                  /* (d, 2 H dk) bf16: transposed K|V projection weight of the cross attention (backward; may be NULL) */
    /* forward: input, then every activation the backward pass needs */
    const void* x_in;                      /* (M, d) layer input */
    void *qkv_s, *ctx_s, *a_s, *y_s;       /* (M, 3 H dk), (M, H dk), (M, d) = fc output then xhat, (M, d) block output */
    float *lse_s, *rstd_s;                 /* (B, H, To), (M) */
    void* q_c;                             /* (M, H dk) */
    const void* kv_c;                      /* (B*T, 2 H dk) K|V of the encoder frames, projected by the caller (row stride ld_kv_c) */
    void* kv_ready_event;                  /* hipEvent_t after which kv_c is complete, or NULL (HOST handle) */
    void *ctx_c, *a_c, *y_c;
    float *lse_c, *rstd_c;
    void *h, *o, *y_f;                     /* (M, ff) ReLU output, (M, d) w_2 output then xhat, (M, d) layer output */
    float* rstd_f;
    /* backward: gradients wrt activations (bf16), bias gradients (f32, accumulated), scratch */
    void *dz_f, *g_o, *g_h, *dx_f;         /* residual gradient of the FFN block, dY of w_2 (only with dropout, else dz_f is it), dY of w_1, dX of w_1 */
    void *dz_c, *g_ac, *g_qc, *g_kvc, *dx_c;   /* cross block: residual gradient, dY of fc (dropout only), dY of the Q projection (M, H dk), dY of K|V (B*T, 2 H dk), dX of Q */
    void *dz_s, *g_as, *g_qkv, *dx_s;      /* self block: residual gradient, dY of fc (dropout only), dY of Q|K|V (M, 3 H dk), dX of Q|K|V */
    void* dctx;                            /* (M, H dk) scratch: gradient wrt an attention output */
    float *gb_2, *gb_fc_c, *gb_fc_s;       /* bias gradients of w_2 and the two out-projections: come out of the LayerNorm backward (partial sums) */
    void *part_f, *part_c, *part_s;        /* asr_add_ln_bwd_workspace_bytes(M, d) bytes each: partial sums of the LayerNorm parameter gradients */
    float* delta;                          /* scratch of asr_sdpa_bwd */
    size_t delta_bytes;
    void* d_enc;                           /* (B*T, d) bf16 gradient wrt the encoder output, accumulated in place (or NULL) */
    void* wgrad_stream;                    /* backward (ABI 7): stream that will run the layer's weight gradients, or NULL.  The LAST kernel of
                                              asr_decoder_layer_bwd then hands over to it by its own completion event (asr_stream_arm): check
                                              asr_stream_arm_pending() afterwards and fall back to asr_stream_fork when it returns 1 */
    int aux_cus;                           /* backward (ABI 8): > 0 = the (B*T)-row GEMM d_enc += g_kvc W_kv that the layer puts on aux_stream is
                                              sized for this many CUs (tuning option "cu_limit" around that one launch), so that the rest stay free
                                              for the layer's own chain of small kernels; 0 = the whole device */
    /* ABI 9: the K | V projections of ALL decoder layers read the same encoder output (transformer_official.py:309-314, 446-458): the caller may
     * keep them - and the gradients wrt them - side by side in one (B*T, L 2 H dk) buffer and run the encoder-output gradient once per GROUP of
     * layers with the reduction over the group's columns, instead of one read-modify-write of d_enc per layer. */
    int ld_kv_c;                           /* row stride (elements) of kv_c and g_kvc; 0 = 2 H dk (a buffer per layer) */
    int kv_dgrad_cols;                     /* 0: d_enc += g_kvc W_kv over this layer's 2 H dk columns (as before).  > 0: this layer is the LAST of a
                                              group to run its backward pass: d_enc += g_kv_group W_group over kv_dgrad_cols columns, with g_kv_group
                                              (B*T, kv_dgrad_cols) at row stride ld_kv_c and w_kv_c_T = the group's (d, kv_dgrad_cols) slice of the
                                              transposed weight; the other layers of the group pass d_enc = NULL */
    const void* g_kv_group;
    /* ABI 10: the low-order pieces of the two attention outputs (asr_sdpa_fwd's o_lo), (M, H dk) each, or NULL */
    void *ctx_s_lo, *ctx_c_lo;
} asr_dec_layer_plan;
int asr_decoder_layer_fwd(const asr_dec_layer_plan* plan, void* stream);
/* (dy, dy2): gradient wrt y_f (dy2 may be NULL; the two are added).  Results: plan->dx_s and plan->dz_s = gradient wrt x_in through the
 * projections and along the residual path (to be added by the consumer).  aux_stream: stream for d_enc += g_kvc W_kv (forked behind
 * `stream`), or NULL = on `stream`. */
int asr_decoder_layer_bwd(const asr_dec_layer_plan* plan, const void* dy, const void* dy2, void* stream, void* aux_stream);

/* ---------------------------------------------------------------------------------------------
 * Log-mel front end on device.
 * Replaces:  MelSpectrogram(sr=16000, ws=400, hop=160, n_mels) -> log(x+1e-20)
 *              Predictor/data_handler/processor.py:33-40
 *            (f - mean)/std (scalar, unbiased) and build_LFR_features   processor.py:42-46, 74-100
 * wav: (B, Smax) f32, wav_len: (B) int32 samples.  feat: (B, Tmax, n_mels) f32 log-mel with
 * Tmax >= 1 + Smax/160; frames t >= 1 + wav_len[b]/160 are written as 0.
 * melfb: (201, n_mels) f32 filterbank; window: (400) f32; twiddle: (400) float2 cos/sin table.
 */
int asr_logmel_fwd(const float* wav, const int32_t* wav_len, const float* window,
                   const float* melfb, float* feat, int B, int Smax, int Tmax, int n_mels,
                   void* stream);
/* out: (B, Tlfr_max, m*n_mels) `dtype`, out_len[b] = ceil(T_b / n); padded rows 0. */
int asr_utt_norm_lfr_fwd(const float* feat, const int32_t* wav_len, void* out, int32_t* out_len,
                         int B, int Tmax, int n_mels, int m, int n, int Tlfr_max, int dtype,
                         void* stream);
/* The same with SpecAugment between normalisation and frame stacking, as AudioParser.parse(augment=
 * True) does (processor.py:52-58, 67-69; augments.py:4-42): masks (B, 4) int32 = [t0, t1, f0, f1]
 * per utterance (frames / mel channels of the un-stacked feature, host RNG); frames [t0, t1) are
 * filled with the mean of the normalised feature, then channels [f0, f1) with the mean of the
 * time-masked feature.  masks == NULL: no augmentation. */
int asr_utt_norm_augment_lfr_fwd(const float* feat, const int32_t* wav_len, const int32_t* masks,
                                 void* out, int32_t* out_len, int B, int Tmax, int n_mels, int m,
                                 int n, int Tlfr_max, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
