// Native launch sequencer of one DECODER layer (host code only: no kernels here).
//
// The decoder works on B * To ~ 550 rows: its ~23 kernels per layer and direction run 4 .. 13 us each, and with the launch loop in
// Python (tensor allocation, wrapper checks, one foreign call per kernel: ~8 - 10 us of host time each) the host - not the GPU -
// paced that part of the joint step (kernel trace, round 3: 8-us kernels start 17 us apart; the joint step took 5.65 ms against
// 3.44 ms for the encoder + CTC alone).  asr_decoder_layer_fwd / _bwd issue the fixed kernel sequence of one layer from C++ - the
// same entry points the Python engine calls (asr_gemm_small_bf16, asr_sdpa_fwd / _bwd, asr_add_ln_fwd / _bwd, asr_gemm_nt_bf16),
// in the same order, on caller-owned buffers described by a plan struct - so the host cost per kernel is hipLaunchKernel's.
// Reference: DecoderLayer.forward (transformer_official.py:446-458) = self-attention, encoder-decoder attention, position-wise FFN,
// each MultiHeadAttention / PositionwiseFeedForwardUseConv with post-LayerNorm and pad zeroing (attention.py:33-62, module.py:68-75),
// and autograd through them.  Weight gradients, LayerNorm parameter-gradient reductions and data-parallel marks stay with the caller.
#include "asr_common.h"

// a failing call ends the layer: an arm this sequencer set (asr_stream_arm) must not survive it - the next armed-capable launch of the
// process would otherwise carry a completion event nobody asked for, and the caller's next fork would be skipped
#define DEC_TRY(call)                \
    do {                             \
        const int rc__ = (call);     \
        if (rc__ != ASR_OK) {        \
            asr_stream_arm_pending(); \
            return rc__;             \
        }                            \
    } while (0)

static int dec_check(const asr_dec_layer_plan* p, const char* who) {
    if (!p) ASR_FAIL(ASR_EINVAL, "%s: null plan", who);
    if (p->B <= 0 || p->To <= 0 || p->T <= 0 || p->H <= 0 || p->dk <= 0 || p->d <= 0 || p->ff <= 0)
        ASR_FAIL(ASR_EINVAL, "%s: bad dims B=%d To=%d T=%d d=%d H=%d dk=%d ff=%d", who, p->B, p->To, p->T, p->d, p->H, p->dk, p->ff);
    if (p->d % 8 || (p->H * p->dk) % 8 || p->ff % 8) ASR_FAIL(ASR_EINVAL, "%s: d, H*dk and ff must be multiples of 8", who);
    return ASR_OK;
}

extern "C" int asr_decoder_layer_fwd(const asr_dec_layer_plan* p, void* stream) {
    DEC_TRY(dec_check(p, "asr_decoder_layer_fwd"));
    const int B = p->B, To = p->To, T = p->T, d = p->d, H = p->H, dk = p->dk, hd = H * dk, ff = p->ff, M = B * To;
    const float scale = 1.0f / sqrtf((float)dk);
    const size_t e = 2;      // bf16
    char* qkv = (char*)p->qkv_s;
    // ---- self-attention (causal over the target prefix; attention.py:33-62)
    DEC_TRY(asr_gemm_small_bf16(p->x_in, p->w_qkv_s, p->b_qkv_s, nullptr, p->qkv_s, M, 3 * hd, d, d, d, 3 * hd, 0, ASR_ACT_NONE, stream));
    DEC_TRY(asr_sdpa_fwd(qkv, qkv + hd * e, qkv + 2 * hd * e, p->ctx_s, p->lse_s, p->dec_len, B, H, To, To, dk, 3 * hd, 3 * hd, 3 * hd, hd, 1, -1, scale,
                         p->drop_p, p->seed[0], p->ctx_s_lo, ASR_BF16, stream));
    DEC_TRY(asr_gemm_small_bf16(p->ctx_s, p->w_fc_s, p->b_fc_s, nullptr, p->a_s, M, d, hd, hd, hd, d, 0, ASR_ACT_NONE, stream));
    DEC_TRY(asr_add_ln_fwd(p->a_s, p->x_in, p->g_s, p->be_s, nullptr, p->dec_len, p->y_s, p->a_s, p->rstd_s, B, To, d, p->drop_p, p->seed[1], ASR_DROP_PRE,
                           ASR_BF16, stream));
    // ---- encoder-decoder attention: K | V of the encoder frames were projected ahead of time (kv_c), possibly on another stream
    DEC_TRY(asr_gemm_small_bf16(p->y_s, p->w_q_c, p->b_q_c, nullptr, p->q_c, M, hd, d, d, d, hd, 0, ASR_ACT_NONE, stream));
    if (p->kv_ready_event) {
        const hipError_t he = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)p->kv_ready_event, 0);
        if (he != hipSuccess) ASR_FAIL(ASR_EHIP, "asr_decoder_layer_fwd: hipStreamWaitEvent: %s", hipGetErrorString(he));
    }
    char* kv = (char*)p->kv_c;
    const int ldkv = p->ld_kv_c > 0 ? p->ld_kv_c : 2 * hd;      // the K | V of all layers may sit side by side in one buffer
    DEC_TRY(asr_sdpa_fwd(p->q_c, kv, kv + hd * e, p->ctx_c, p->lse_c, p->cross_len, B, H, To, T, dk, hd, ldkv, ldkv, hd, 0, -1, scale, p->drop_p, p->seed[2],
                         p->ctx_c_lo, ASR_BF16, stream));
    DEC_TRY(asr_gemm_small_bf16(p->ctx_c, p->w_fc_c, p->b_fc_c, nullptr, p->a_c, M, d, hd, hd, hd, d, 0, ASR_ACT_NONE, stream));
    DEC_TRY(asr_add_ln_fwd(p->a_c, p->y_s, p->g_c, p->be_c, nullptr, p->dec_len, p->y_c, p->a_c, p->rstd_c, B, To, d, p->drop_p, p->seed[3], ASR_DROP_PRE,
                           ASR_BF16, stream));
    // ---- position-wise feed-forward (module.py:68-75)
    DEC_TRY(asr_gemm_small_bf16(p->y_c, p->w_1, p->b_1, nullptr, p->h, M, ff, d, d, d, ff, 0, ASR_ACT_RELU, stream));
    DEC_TRY(asr_gemm_small_bf16(p->h, p->w_2, p->b_2, nullptr, p->o, M, d, ff, ff, ff, d, 0, ASR_ACT_NONE, stream));
    DEC_TRY(asr_add_ln_fwd(p->o, p->y_c, p->g_f, p->be_f, nullptr, p->dec_len, p->y_f, p->o, p->rstd_f, B, To, d, p->drop_p, p->seed[4], ASR_DROP_PRE, ASR_BF16,
                           stream));
    return ASR_OK;
}

// Backward of the layer: (dy, dy2) = gradient wrt y_f (dy2 may be NULL) -> (dx_out, dz_out) = gradient wrt x_in through the
// projections and along the residual path (the consumer adds them, as asr_add_ln_bwd does with its dy / dy2 pair).
// Also written, for the caller's weight-gradient GEMMs: g_o (dY of w_2), g_h (dY of w_1), g_ac (dY of the cross out-projection),
// g_qc, g_kvc (dY of the cross Q and K|V projections), g_as (dY of the self out-projection), g_qkv (dY of the fused Q|K|V).
// LayerNorm parameter gradients: per-workgroup partial sums are left in part_f / part_c / part_s (asr_add_ln_bwd_reduce_batched).
// d_enc += g_kvc * W_kv (all encoder frames) runs on aux_stream behind a fork when aux_stream != NULL, else on `stream`.
extern "C" int asr_decoder_layer_bwd(const asr_dec_layer_plan* p, const void* dy, const void* dy2, void* stream, void* aux_stream) {
    DEC_TRY(dec_check(p, "asr_decoder_layer_bwd"));
    if (!dy) ASR_FAIL(ASR_EINVAL, "asr_decoder_layer_bwd: null dy");
    const int B = p->B, To = p->To, T = p->T, d = p->d, H = p->H, dk = p->dk, hd = H * dk, ff = p->ff, M = B * To;
    const float scale = 1.0f / sqrtf((float)dk);
    const size_t e = 2;
    const bool drop = p->drop_p > 0.f;
    const size_t part_bytes = asr_add_ln_bwd_workspace_bytes(M, d);
    // ---- feed-forward block
    void* dxg_f = drop ? p->g_o : p->dz_f;      // pre-residual dropout: the gradient wrt the projection output differs from the residual one
    DEC_TRY(asr_add_ln_bwd(dy, dy2, p->o, p->rstd_f, p->g_f, p->dec_len, p->dz_f, drop ? p->g_o : nullptr, nullptr, nullptr, p->gb_2, p->part_f, part_bytes, B, To, d,
                           p->drop_p, p->seed[4], ASR_DROP_PRE, ASR_BF16, stream));
    DEC_TRY(asr_gemm_small_bf16(dxg_f, p->w_2, nullptr, p->h, p->g_h, M, ff, d, d, ff, ff, 1, ASR_ACT_RELU_MASK, stream));      // dh = (dY W_2) masked by the ReLU
    DEC_TRY(asr_gemm_small_bf16(p->g_h, p->w_1, nullptr, nullptr, p->dx_f, M, d, ff, ff, d, d, 1, ASR_ACT_NONE, stream));
    // ---- encoder-decoder attention block
    void* dxg_c = drop ? p->g_ac : p->dz_c;
    DEC_TRY(asr_add_ln_bwd(p->dx_f, p->dz_f, p->a_c, p->rstd_c, p->g_c, p->dec_len, p->dz_c, drop ? p->g_ac : nullptr, nullptr, nullptr, p->gb_fc_c, p->part_c, part_bytes,
                           B, To, d, p->drop_p, p->seed[3], ASR_DROP_PRE, ASR_BF16, stream));
    DEC_TRY(asr_gemm_small_bf16(dxg_c, p->w_fc_c, nullptr, nullptr, p->dctx, M, hd, d, d, hd, hd, 1, ASR_ACT_NONE, stream));
    char* kv = (char*)p->kv_c;
    char* gkv = (char*)p->g_kvc;
    // d_enc += dK|dV W_kv (a (B*T)-row GEMM off the decoder's dependent chain, on aux_stream) needs only the attention backward's dK|dV:
    // that kernel hands over by its own completion event (no event record - a barrier packet - in front of the chain's next kernel)
    const bool kv_dgrad = p->d_enc && p->w_kv_c_T;
    const int ldkv = p->ld_kv_c > 0 ? p->ld_kv_c : 2 * hd;
    if (p->kv_dgrad_cols > 0 && (!p->g_kv_group || p->kv_dgrad_cols % 8 || p->kv_dgrad_cols > ldkv)) ASR_FAIL(ASR_EINVAL, "asr_decoder_layer_bwd: kv_dgrad_cols = %d needs g_kv_group, a multiple of 8, <= ld_kv_c = %d", p->kv_dgrad_cols, ldkv);
    if (kv_dgrad && aux_stream) DEC_TRY(asr_stream_arm(stream, aux_stream));
    DEC_TRY(asr_sdpa_bwd(p->q_c, kv, kv + hd * e, p->ctx_c, p->dctx, p->lse_c, p->delta, p->delta_bytes, p->g_qc, gkv, gkv + hd * e, p->cross_len, B, H, To, T, dk, hd,
                         ldkv, ldkv, hd, 0, -1, scale, p->drop_p, p->seed[2], p->ctx_c_lo, ASR_BF16, stream));
    if (kv_dgrad) {
        void* st2 = aux_stream ? aux_stream : stream;
        if (aux_stream && asr_stream_arm_pending()) DEC_TRY(asr_stream_fork(stream, aux_stream));      // the attention path taken had no armed launch
        // a (B*T)-row GEMM beside the chain of small kernels: sized for p->aux_cus CUs when the caller reserves the rest for the chain
        const int old_lim = p->aux_cus > 0 ? asr_option_set(ASR_OPT_CU_LIMIT, p->aux_cus) : 0;
        // one layer's columns, or - this layer being the last of its group - the whole group's: ONE read-modify-write of d_enc per group
        const void* g_a = p->kv_dgrad_cols > 0 ? p->g_kv_group : p->g_kvc;
        const int cols = p->kv_dgrad_cols > 0 ? p->kv_dgrad_cols : 2 * hd;
        const int rc_kv = asr_gemm_nt_bf16(g_a, p->w_kv_c_T, nullptr, p->d_enc, p->d_enc, B * T, d, cols, ldkv, p->ld_kv_c_T, d, ASR_ACT_NONE, st2);
        if (p->aux_cus > 0) asr_option_set(ASR_OPT_CU_LIMIT, old_lim);
        DEC_TRY(rc_kv);
    }
    DEC_TRY(asr_gemm_small_bf16(p->g_qc, p->w_q_c, nullptr, nullptr, p->dx_c, M, d, hd, hd, d, d, 1, ASR_ACT_NONE, stream));
    // ---- self-attention block
    void* dxg_s = drop ? p->g_as : p->dz_s;
    DEC_TRY(asr_add_ln_bwd(p->dx_c, p->dz_c, p->a_s, p->rstd_s, p->g_s, p->dec_len, p->dz_s, drop ? p->g_as : nullptr, nullptr, nullptr, p->gb_fc_s, p->part_s, part_bytes,
                           B, To, d, p->drop_p, p->seed[1], ASR_DROP_PRE, ASR_BF16, stream));
    DEC_TRY(asr_gemm_small_bf16(dxg_s, p->w_fc_s, nullptr, nullptr, p->dctx, M, hd, d, d, hd, hd, 1, ASR_ACT_NONE, stream));
    char* qkv = (char*)p->qkv_s;
    char* gq = (char*)p->g_qkv;
    DEC_TRY(asr_sdpa_bwd(qkv, qkv + hd * e, qkv + 2 * hd * e, p->ctx_s, p->dctx, p->lse_s, p->delta, p->delta_bytes, gq, gq + hd * e, gq + 2 * hd * e, p->dec_len, B, H, To,
                         To, dk, 3 * hd, 3 * hd, 3 * hd, hd, 1, -1, scale, p->drop_p, p->seed[0], p->ctx_s_lo, ASR_BF16, stream));
    if (p->wgrad_stream) DEC_TRY(asr_stream_arm(stream, p->wgrad_stream));      // the layer's last kernel hands over to the weight-gradient stream
    DEC_TRY(asr_gemm_small_bf16(p->g_qkv, p->w_qkv_s, nullptr, nullptr, p->dx_s, M, d, 3 * hd, 3 * hd, d, d, 1, ASR_ACT_NONE, stream));
    return ASR_OK;
}
