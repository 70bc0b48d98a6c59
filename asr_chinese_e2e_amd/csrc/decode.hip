// Decoding kernels (greedy / beam decoding on the GPU).
//
//   asr_ctc_greedy_decode : best path of the CTC head, collapsed (repeats merged, blanks removed).
//       NOT in the reference (it has no CTC; its only decoder is the per-hypothesis Python beam
//       loop transformer_official.py:331-434).  Semantics: argmax over the vocabulary per frame
//       (first index wins ties, as torch.argmax), then the standard CTC collapse B(.) of Graves 2006.
#include "asr_common.h"

namespace {

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// (value, index) pair reduction over the wave: larger value wins, equal values -> smaller index
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(v, o, 64);
        const int i2 = __shfl_xor(i, o, 64);
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
    }
}

// one wave per frame; frames at or past in_len[b] get `blank`
template <typename T>
__global__ __launch_bounds__(256) void frame_argmax_kernel(const T* __restrict__ logits, const int32_t* __restrict__ in_len, int32_t* __restrict__ path,
                                                           int B, int T_, int V, int ld, int blank) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) {
            if (lane == 0) path[row] = blank;
            continue;
        }
        const T* x = logits + (size_t)row * ld;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if constexpr (sizeof(T) == 2) {
            if (V % 8 == 0 && ((uintptr_t)x % 16) == 0) {
                const int nvec = V >> 3;
                for (int k = lane; k < nvec; k += 64) {   // ascending index inside a lane: strict > keeps the first maximum
                    const u32x4 q = *(const u32x4*)(x + (size_t)k * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = bf16_lo(q[j]), hi = bf16_hi(q[j]);
                        if (lo > best) { best = lo; bi = k * 8 + 2 * j; }
                        if (hi > best) { best = hi; bi = k * 8 + 2 * j + 1; }
                    }
                }
                wave_argmax(best, bi);
                if (lane == 0) path[row] = bi;
                continue;
            }
        }
        for (int i = lane; i < V; i += 64) {
            const float v = to_f32<T>(x[i]);
            if (v > best) { best = v; bi = i; }
        }
        wave_argmax(best, bi);
        if (lane == 0) path[row] = bi;
    }
}

// one wave per utterance, in place: ids[b][0..len) = collapsed path, the rest 0; out_len[b] = len
__global__ __launch_bounds__(64) void ctc_collapse_kernel(int32_t* __restrict__ ids, const int32_t* __restrict__ in_len, int32_t* __restrict__ out_len, int T_,
                                                          int blank) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int32_t* p = ids + (size_t)b * T_;
    const int Tb = min(in_len[b], T_);
    int n = 0, prev = blank;   // prev = label of the frame before the chunk (blank before the first frame: no merge)
    for (int t0 = 0; t0 < Tb; t0 += 64) {
        const int t = t0 + lane;
        const int cur = t < Tb ? p[t] : blank;
        int left = __shfl_up(cur, 1, 64);
        if (lane == 0) left = prev;
        const bool keep = t < Tb && cur != blank && (t == 0 || cur != left);
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        // every lane has read its entry of this chunk before any lane writes (same instruction stream);
        // writes land at positions <= t, never in a later chunk
        if (keep) p[n + rank] = cur;
        n += __popcll(m);
        prev = __shfl(cur, 63, 64);
    }
    for (int t = n + lane; t < T_; t += 64) p[t] = 0;
    if (lane == 0) out_len[b] = n;
}

}  // namespace

extern "C" int asr_ctc_frame_argmax(const void* logits, const int32_t* in_len, int32_t* path, int B, int T, int V, int ld, int blank, int dtype, void* stream) {
    if (!logits || !in_len || !path) ASR_FAIL(ASR_EINVAL, "asr_ctc_frame_argmax: null pointer");
    if (B <= 0 || T <= 0 || V <= 1 || blank < 0 || blank >= V) ASR_FAIL(ASR_EINVAL, "asr_ctc_frame_argmax: bad shape B=%d T=%d V=%d blank=%d", B, T, V, blank);
    if (ld < V) ASR_FAIL(ASR_EINVAL, "asr_ctc_frame_argmax: row stride ld=%d < V=%d", ld, V);
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_ctc_frame_argmax: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * T;
    int g = ceil_div(rows, 4);
    if (g > 4096) g = 4096;
    if (dtype == ASR_F32) frame_argmax_kernel<float><<<g, 256, 0, st>>>((const float*)logits, in_len, path, B, T, V, ld, blank);
    else frame_argmax_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)logits, in_len, path, B, T, V, ld, blank);
    ASR_CHECK_LAUNCH("asr_ctc_frame_argmax");
    return ASR_OK;
}

extern "C" int asr_ctc_collapse(int32_t* ids, const int32_t* in_len, int32_t* out_len, int B, int T, int blank, void* stream) {
    if (!ids || !in_len || !out_len) ASR_FAIL(ASR_EINVAL, "asr_ctc_collapse: null pointer");
    if (B <= 0 || T <= 0 || blank < 0) ASR_FAIL(ASR_EINVAL, "asr_ctc_collapse: bad shape B=%d T=%d blank=%d", B, T, blank);
    ctc_collapse_kernel<<<B, 64, 0, (hipStream_t)stream>>>(ids, in_len, out_len, T, blank);
    ASR_CHECK_LAUNCH("asr_ctc_collapse");
    return ASR_OK;
}

extern "C" int asr_ctc_greedy_decode(const void* logits, const int32_t* in_len, int32_t* out_ids, int32_t* out_len, int B, int T, int V, int ld, int blank,
                                     int dtype, void* stream) {
    if (!out_len) ASR_FAIL(ASR_EINVAL, "asr_ctc_greedy_decode: null pointer");
    const int rc = asr_ctc_frame_argmax(logits, in_len, out_ids, B, T, V, ld, blank, dtype, stream);
    if (rc != ASR_OK) return rc;
    return asr_ctc_collapse(out_ids, in_len, out_len, B, T, blank, stream);
}

// ================================================================================================
// Attention-decoder beam search (transformer_official.py:331-434), batched over utterances and
// beams with key/value caches.  The reference re-runs the whole decoder over the growing prefix for
// every hypothesis and step (O(len^2) layers per hypothesis, Python loop, one utterance at a
// time); here one step costs one token per live hypothesis:
//   asr_decode_attn        single-query attention over a K/V cache (self) or the encoder K/V (cross)
//   asr_logsoftmax_topk    log_softmax over the vocabulary + the `beam` best entries per row
//   asr_beam_step          per utterance: merge beam x beam candidates, keep the best `beam`
//                          (stable order, as Python's sorted()), retire hypotheses that emitted
//                          eos, force eos at the last step; records (token, parent, end, score)
//                          per step for the host-side backtrace
//   asr_cache_gather       reorder the self-attention caches by parent hypothesis
namespace {

// one wave per (row r, head h).  Keys/values of row r are rows (r / kv_div) * Tk_cap + t of k / v.
template <typename T>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ o,
                                                          const int32_t* __restrict__ k_len, int k_len_uniform, int len_div, int R, int H, int dk,
                                                          int Tk_cap, int kv_div, int ldq, int ldk, int ldv, int ldo, float scale) {
    extern __shared__ float sc_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* sc = sc_all + (size_t)w * Tk_cap;
    const int pair = blockIdx.x * 4 + w;
    if (pair >= R * H) return;
    const int r = pair / H, h = pair - r * H;
    int n = k_len ? k_len[r / len_div] : k_len_uniform;
    n = min(n, Tk_cap);
    const T* qp = q + (size_t)r * ldq + h * dk;
    const size_t kv_row0 = (size_t)(r / kv_div) * Tk_cap;
    const T* kp = k + kv_row0 * ldk + h * dk;
    const T* vp = v + kv_row0 * ldv + h * dk;
    // phase 1: lane per key
    float m = -INFINITY;
    for (int t = lane; t < n; t += 64) {
        const T* kr = kp + (size_t)t * ldk;
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s = fmaf(to_f32<T>(qp[d]), to_f32<T>(kr[d]), s);
        s *= scale;
        sc[t] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < n; t += 64) {
        const float e = expf(sc[t] - m);
        sc[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    // phase 2: lane per output dimension
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    for (int d = lane; d < dk; d += 64) {
        float acc = 0.f;
        for (int t = 0; t < n; ++t) acc = fmaf(sc[t], to_f32<T>(vp[(size_t)t * ldv + d]), acc);
        o[(size_t)r * ldo + h * dk + d] = from_f32<T>(acc * inv);
    }
}

// one wave per row: vals[j] = j-th largest log_softmax value (ties: smaller index first), ids[j] its index
template <typename T>
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const T* __restrict__ logits, float* __restrict__ vals, int32_t* __restrict__ ids, int R, int V,
                                                              int ld, int beam, float* __restrict__ extra_lp = nullptr, int extra_id = 0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= R) return;
    const T* x = logits + (size_t)r * ld;
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, to_f32<T>(x[i]));
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += expf(to_f32<T>(x[i]) - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    if (extra_lp && lane == 0) extra_lp[r] = to_f32<T>(x[extra_id]) - lse;      // log_softmax of one fixed class (the CTC blank) beside the top-k
    float last_v = INFINITY;
    int last_i = -1;
    for (int j = 0; j < beam; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < V; i += 64) {
            const float xv = to_f32<T>(x[i]);
            const bool eligible = xv < last_v || (xv == last_v && i > last_i);   // strictly after the previous pick
            if (eligible && (xv > bv || (xv == bv && i < bi))) { bv = xv; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) {
            vals[(size_t)r * beam + j] = bv - lse;
            ids[(size_t)r * beam + j] = bi == 0x7fffffff ? 0 : bi;
        }
        last_v = bv;
        last_i = bi;
    }
}

// one 64-lane workgroup per utterance; beam * beam <= 64
__global__ __launch_bounds__(64) void beam_step_kernel(const float* __restrict__ top_vals, const int32_t* __restrict__ top_ids, float* __restrict__ score,
                                                       int32_t* __restrict__ alive, int32_t* __restrict__ last_tok, int32_t* __restrict__ parent,
                                                       int32_t* __restrict__ rec_tok, int32_t* __restrict__ rec_par, int32_t* __restrict__ rec_end,
                                                       float* __restrict__ rec_score, const int32_t* __restrict__ maxlen, int32_t* __restrict__ alive_total,
                                                       int B, int beam, int step, int eos) {
    const int b = blockIdx.x, c = threadIdx.x;
    const int h = c / beam, j = c - h * beam;
    const bool valid = c < beam * beam && alive[b * beam + h] != 0 && step < maxlen[b];
    const float cs = valid ? score[b * beam + h] + top_vals[(size_t)(b * beam + h) * beam + j] : -INFINITY;
    // stable rank among the valid candidates (Python sorted(reverse=True) keeps first-come order on ties)
    int rank = 0;
    for (int o = 0; o < 64; ++o) {
        const float so = __shfl(cs, o, 64);
        const int vo = __shfl((int)valid, o, 64);
        if (vo && (so > cs || (so == cs && o < c))) ++rank;
    }
    const bool keep = valid && rank < beam;
    const int tok = keep ? top_ids[(size_t)(b * beam + h) * beam + j] : 0;
    const bool last = step == maxlen[b] - 1;
    __syncthreads();   // every lane has read the old state
    // default for the slots no candidate lands in
    if (c < beam) {
        const size_t rec = ((size_t)step * B + b) * beam + c;
        alive[b * beam + c] = 0;
        parent[b * beam + c] = c;
        rec_tok[rec] = 0;
        rec_par[rec] = 0;
        rec_end[rec] = 0;
        rec_score[rec] = -INFINITY;
    }
    __syncthreads();
    if (keep) {
        const int kslot = rank;
        const size_t rec = ((size_t)step * B + b) * beam + kslot;
        // a hypothesis leaves the beam when it emits eos; at the last step eos is APPENDED to every
        // survivor (also after an eos of its own: transformer_official.py:399-403)
        const int end = last ? 2 : (tok == eos ? 1 : 0);
        score[b * beam + kslot] = cs;
        last_tok[b * beam + kslot] = tok;
        parent[b * beam + kslot] = h;
        alive[b * beam + kslot] = end ? 0 : 1;
        rec_tok[rec] = tok;
        rec_par[rec] = h;
        rec_end[rec] = end;
        rec_score[rec] = cs;
        if (!end) atomicAdd(alive_total, 1);
    }
}

// dst[l][r][t][:] = src[l][b*beam + parent[r]][t][:]  for t < n_pos; rows of row_bytes bytes (multiple of 16)
__global__ __launch_bounds__(256) void cache_gather_kernel(const char* __restrict__ src, char* __restrict__ dst, const int32_t* __restrict__ parent, int L, int R,
                                                           int beam, int Lcap, int n_pos, int row_bytes) {
    const int lr = blockIdx.x;   // l * R + r
    const int l = lr / R, r = lr - l * R;
    const int b = r / beam;
    const int pr = b * beam + parent[r];
    const size_t per_row = (size_t)Lcap * row_bytes;
    const u32x4* s = (const u32x4*)(src + ((size_t)l * R + pr) * per_row);
    u32x4* d = (u32x4*)(dst + ((size_t)l * R + r) * per_row);
    const int nvec = n_pos * (row_bytes / 16);
    for (int i = threadIdx.x; i < nvec; i += 256) d[i] = s[i];
}

}  // namespace

extern "C" int asr_decode_attn(const void* q, const void* k, const void* v, void* o, const int32_t* k_len, int k_len_uniform, int len_div, int R, int H,
                               int dk, int Tk_cap, int kv_div, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !o) ASR_FAIL(ASR_EINVAL, "asr_decode_attn: null pointer");
    if (R <= 0 || H <= 0 || dk <= 0 || Tk_cap <= 0 || kv_div <= 0 || len_div <= 0) ASR_FAIL(ASR_EINVAL, "asr_decode_attn: bad shape R=%d H=%d dk=%d Tk=%d", R, H, dk, Tk_cap);
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_decode_attn: dtype %d", dtype);
    const size_t lds = (size_t)4 * Tk_cap * sizeof(float);
    if (lds > 160 * 1024) ASR_FAIL(ASR_EINVAL, "asr_decode_attn: Tk_cap=%d does not fit the LDS score buffer", Tk_cap);
    hipStream_t st = (hipStream_t)stream;
    const int grid = ceil_div(R * H, 4);
    if (dtype == ASR_F32) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)decode_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        decode_attn_kernel<float><<<grid, 256, lds, st>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, k_len, k_len_uniform, len_div, R, H, dk, Tk_cap, kv_div, ldq, ldk, ldv, ldo, scale);
    } else {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)decode_attn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        decode_attn_kernel<bf16_t><<<grid, 256, lds, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, k_len, k_len_uniform, len_div, R, H, dk, Tk_cap, kv_div, ldq, ldk, ldv, ldo, scale);
    }
    ASR_CHECK_LAUNCH("asr_decode_attn");
    return ASR_OK;
}

extern "C" int asr_logsoftmax_topk(const void* logits, float* vals, int32_t* ids, int R, int V, int ld, int beam, int dtype, void* stream) {
    if (!logits || !vals || !ids) ASR_FAIL(ASR_EINVAL, "asr_logsoftmax_topk: null pointer");
    if (R <= 0 || V <= 0 || ld < V || beam <= 0 || beam > V) ASR_FAIL(ASR_EINVAL, "asr_logsoftmax_topk: bad shape R=%d V=%d ld=%d beam=%d", R, V, ld, beam);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_F32) logsoftmax_topk_kernel<float><<<ceil_div(R, 4), 256, 0, st>>>((const float*)logits, vals, ids, R, V, ld, beam);
    else if (dtype == ASR_BF16) logsoftmax_topk_kernel<bf16_t><<<ceil_div(R, 4), 256, 0, st>>>((const bf16_t*)logits, vals, ids, R, V, ld, beam);
    else ASR_FAIL(ASR_EDTYPE, "asr_logsoftmax_topk: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_logsoftmax_topk");
    return ASR_OK;
}

extern "C" int asr_ctc_frame_topk(const void* logits, float* vals, int32_t* ids, float* blank_lp, int R, int V, int ld, int k, int blank, int dtype, void* stream) {
    if (!logits || !vals || !ids || !blank_lp) ASR_FAIL(ASR_EINVAL, "asr_ctc_frame_topk: null pointer");
    if (R <= 0 || V <= 0 || ld < V || k <= 0 || k > V || blank < 0 || blank >= V) ASR_FAIL(ASR_EINVAL, "asr_ctc_frame_topk: bad shape R=%d V=%d ld=%d k=%d blank=%d", R, V, ld, k, blank);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_F32) logsoftmax_topk_kernel<float><<<ceil_div(R, 4), 256, 0, st>>>((const float*)logits, vals, ids, R, V, ld, k, blank_lp, blank);
    else if (dtype == ASR_BF16) logsoftmax_topk_kernel<bf16_t><<<ceil_div(R, 4), 256, 0, st>>>((const bf16_t*)logits, vals, ids, R, V, ld, k, blank_lp, blank);
    else ASR_FAIL(ASR_EDTYPE, "asr_ctc_frame_topk: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_ctc_frame_topk");
    return ASR_OK;
}

extern "C" int asr_beam_step(const float* top_vals, const int32_t* top_ids, float* score, int32_t* alive, int32_t* last_tok, int32_t* parent, int32_t* rec_tok,
                             int32_t* rec_par, int32_t* rec_end, float* rec_score, const int32_t* maxlen, int32_t* alive_total, int B, int beam, int step,
                             int eos, void* stream) {
    if (!top_vals || !top_ids || !score || !alive || !last_tok || !parent || !rec_tok || !rec_par || !rec_end || !rec_score || !maxlen || !alive_total)
        ASR_FAIL(ASR_EINVAL, "asr_beam_step: null pointer");
    if (B <= 0 || beam <= 0 || beam * beam > 64 || step < 0) ASR_FAIL(ASR_EINVAL, "asr_beam_step: bad shape B=%d beam=%d (beam <= 8) step=%d", B, beam, step);
    beam_step_kernel<<<B, 64, 0, (hipStream_t)stream>>>(top_vals, top_ids, score, alive, last_tok, parent, rec_tok, rec_par, rec_end, rec_score, maxlen, alive_total, B, beam, step, eos);
    ASR_CHECK_LAUNCH("asr_beam_step");
    return ASR_OK;
}

extern "C" int asr_cache_gather(const void* src, void* dst, const int32_t* parent, int L, int R, int beam, int Lcap, int n_pos, int row_bytes, void* stream) {
    if (!src || !dst || !parent) ASR_FAIL(ASR_EINVAL, "asr_cache_gather: null pointer");
    if (L <= 0 || R <= 0 || beam <= 0 || R % beam || Lcap <= 0 || n_pos < 0 || n_pos > Lcap || row_bytes <= 0 || row_bytes % 16) ASR_FAIL(ASR_EINVAL, "asr_cache_gather: bad shape");
    if (((uintptr_t)src | (uintptr_t)dst) % 16) ASR_FAIL(ASR_EINVAL, "asr_cache_gather: misaligned pointer");
    if (n_pos > 0) cache_gather_kernel<<<L * R, 256, 0, (hipStream_t)stream>>>((const char*)src, (char*)dst, parent, L, R, beam, Lcap, n_pos, row_bytes);
    ASR_CHECK_LAUNCH("asr_cache_gather");
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------
// CTC prefix beam search on the device (Hannun et al. 2014, algorithm 1 without a language model; the restatement the tests
// check against is oracle/decode_ref.py::ctc_prefix_beam_search).  One wave per utterance walks the frames; the prefixes of the
// beam are nodes of a trie kept in global scratch (parent, token), so "the same string" is "the same (parent node, token)".
// Per frame and beam entry l = (node, last token e, log pb, log pnb), tot = pb (+) pnb:
//     slot 0       stay:       pb' = tot + lp(blank);  pnb' = pnb + lp(e) if e is among the frame's k candidates
//     slot m >= 1  extend c:   pnb' = (c == e ? pb : tot) + lp(c)         (c = m-th candidate, blank skipped)
// An extension (node_l, c) that spells a prefix already in the beam (entry i with parent node_l and token c) is merged into that
// entry's stay slot; the beam * (k + 1) <= 64 slots are ranked (ties: lower slot first, as the stable host sort) and the best
// `beam` form the next beam in rank order; extensions that survive get a new trie node.  Log-probabilities in fp64: rankings
// must not flip against the fp64 host restatement on near ties.  Results: the n best prefixes, spelled by walking parents.
namespace {

__device__ __forceinline__ double pb_logadd(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

constexpr int PB_MAX_BEAM = 16, PB_MAX_K = 32;

__global__ __launch_bounds__(64) void ctc_prefix_beam_kernel(const float* __restrict__ vals, const int32_t* __restrict__ ids, const float* __restrict__ blank_lp,
                                                             const int32_t* __restrict__ in_len, int32_t* __restrict__ nodes, int32_t* __restrict__ out_tok,
                                                             int32_t* __restrict__ out_len, float* __restrict__ out_score, int T, int k, int beam, int nbest,
                                                             int Lcap, int blank) {
    __shared__ double s_pb[PB_MAX_BEAM], s_pnb[PB_MAX_BEAM], s_merge[PB_MAX_BEAM], s_sc[64], s_npb[64], s_npnb[64];
    __shared__ int s_node[PB_MAX_BEAM], s_tok[PB_MAX_BEAM], s_par[PB_MAX_BEAM], s_id[PB_MAX_K], s_nnode[64], s_ntok[64], s_npar[64], s_rank[64];
    __shared__ float s_lp[PB_MAX_K];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cap_nodes = T * beam + 1;                  // node 0 = the empty prefix; at most `beam` new nodes per frame
    int32_t* npar = nodes + (size_t)b * 2 * cap_nodes;   // [parent | token] per node
    int32_t* ntok = npar + cap_nodes;
    const int len = min(in_len ? in_len[b] : T, T);
    if (lane == 0) {
        npar[0] = -1;
        ntok[0] = -1;
        s_node[0] = 0; s_tok[0] = -1; s_par[0] = -1; s_pb[0] = 0.0; s_pnb[0] = -INFINITY;
    }
    int nb = 1, next_node = 1;                           // wave-uniform copies
    __syncthreads();
    const int per = k + 1;
    for (int t = 0; t < len; ++t) {
        const size_t row = (size_t)b * T + t;
        if (lane < k) { s_id[lane] = ids[row * k + lane]; s_lp[lane] = vals[row * k + lane]; }
        if (lane < PB_MAX_BEAM) s_merge[lane] = -INFINITY;
        __syncthreads();
        const double lb = (double)blank_lp[row];
        const int j = lane / per, m = lane - j * per;
        bool valid = j < nb;
        double pb2 = -INFINITY, pnb2 = -INFINITY;
        int c = -1, ident_par = -1;
        if (valid) {
            const double pb = s_pb[j], pnb = s_pnb[j], tot = pb_logadd(pb, pnb);
            const int e = s_tok[j];
            if (m == 0) {                                // stay
                pb2 = tot + lb;
                for (int q = 0; q < k; ++q)
                    if (s_id[q] == e && e != blank) pnb2 = pb_logadd(pnb2, pnb + (double)s_lp[q]);
                c = e;
                ident_par = s_par[j];
            } else {                                     // extend with the (m-1)-th candidate
                c = s_id[m - 1];
                if (c == blank) valid = false;
                else {
                    pnb2 = (c == e ? pb : tot) + (double)s_lp[m - 1];
                    ident_par = s_node[j];
                }
            }
        }
        // an extension that spells a prefix of the current beam goes into that entry's stay slot
        if (valid && m > 0) {
            for (int i = 0; i < nb; ++i)
                if (s_par[i] == ident_par && s_tok[i] == c) {
                    s_merge[i] = pnb2;                   // at most one extension matches an entry (parent and token are unique)
                    valid = false;
                    break;
                }
        }
        __syncthreads();
        if (valid && m == 0) pnb2 = pb_logadd(pnb2, s_merge[j]);
        const double sc = valid ? pb_logadd(pb2, pnb2) : -INFINITY;
        // a slot whose whole probability is zero cannot enter the beam (the host dictionary would hold it with -inf, ranked last)
        valid = valid && sc > -INFINITY;
        s_sc[lane] = sc;
        __syncthreads();
        int rank = 0;
        if (valid) {
            for (int o = 0; o < 64; ++o) {
                const double so = s_sc[o];
                if (so > sc || (so == sc && o < lane && so > -INFINITY)) ++rank;
            }
        }
        const bool keep = valid && rank < beam;
        const unsigned long long keep_mask = __ballot(keep);
        const unsigned long long ext_mask = __ballot(keep && m > 0);
        s_rank[lane] = keep ? rank : -1;
        if (keep) {
            int node = m == 0 ? s_node[j] : next_node + __popcll(ext_mask & ((1ull << lane) - 1ull));
            if (m > 0) { npar[node] = ident_par; ntok[node] = c; }
            s_nnode[rank] = node;
            s_ntok[rank] = c;
            s_npar[rank] = ident_par;
            s_npb[rank] = pb2;
            s_npnb[rank] = pnb2;
        }
        __syncthreads();
        nb = __popcll(keep_mask);
        next_node += __popcll(ext_mask);
        if (lane < nb) {
            s_node[lane] = s_nnode[lane]; s_tok[lane] = s_ntok[lane]; s_par[lane] = s_npar[lane];
            s_pb[lane] = s_npb[lane]; s_pnb[lane] = s_npnb[lane];
        }
        __syncthreads();
    }
    // the beam is in rank order of its total probability: spell the n best
    if (lane < nbest) {
        int32_t* dst = out_tok + ((size_t)b * nbest + lane) * Lcap;
        if (lane < nb) {
            int n = 0;
            for (int nd = s_node[lane]; nd > 0; nd = npar[nd]) ++n;
            out_len[b * nbest + lane] = n;
            out_score[b * nbest + lane] = (float)pb_logadd(s_pb[lane], s_pnb[lane]);
            int pos = min(n, Lcap);
            int skip = n - pos;                          // a prefix longer than the output row keeps its first Lcap tokens
            for (int nd = s_node[lane]; nd > 0; nd = npar[nd]) {
                if (skip > 0) { --skip; continue; }
                dst[--pos] = ntok[nd];
            }
        } else {
            out_len[b * nbest + lane] = -1;              // fewer prefixes than asked for
            out_score[b * nbest + lane] = -INFINITY;
        }
    }
}

}  // namespace

extern "C" size_t asr_ctc_prefix_beam_workspace_bytes(int B, int T, int beam) {
    if (B <= 0 || T <= 0 || beam <= 0) return 0;
    return (size_t)B * 2 * ((size_t)T * beam + 1) * sizeof(int32_t);
}

extern "C" int asr_ctc_prefix_beam(const float* vals, const int32_t* ids, const float* blank_lp, const int32_t* in_len, void* ws, size_t ws_bytes,
                                   int32_t* out_tok, int32_t* out_len, float* out_score, int B, int T, int k, int beam, int nbest, int Lcap, int blank,
                                   void* stream) {
    if (!vals || !ids || !blank_lp || !ws || !out_tok || !out_len || !out_score) ASR_FAIL(ASR_EINVAL, "asr_ctc_prefix_beam: null pointer");
    if (B <= 0 || T <= 0 || k <= 0 || beam <= 0 || nbest <= 0 || Lcap <= 0) ASR_FAIL(ASR_EINVAL, "asr_ctc_prefix_beam: bad shape B=%d T=%d k=%d beam=%d nbest=%d Lcap=%d", B, T, k, beam, nbest, Lcap);
    if (beam > PB_MAX_BEAM || k > PB_MAX_K || beam * (k + 1) > 64 || nbest > beam)
        ASR_FAIL(ASR_EINVAL, "asr_ctc_prefix_beam: one wave ranks the beam * (k + 1) candidates of a frame: beam * (k + 1) <= 64, beam <= %d, nbest <= beam (beam=%d k=%d nbest=%d)", PB_MAX_BEAM, beam, k, nbest);
    if (ws_bytes < asr_ctc_prefix_beam_workspace_bytes(B, T, beam) || ((uintptr_t)ws % 4)) ASR_FAIL(ASR_EWORKSPACE, "asr_ctc_prefix_beam: workspace of %zu bytes needed (got %zu)", asr_ctc_prefix_beam_workspace_bytes(B, T, beam), ws_bytes);
    ctc_prefix_beam_kernel<<<B, 64, 0, (hipStream_t)stream>>>(vals, ids, blank_lp, in_len, (int32_t*)ws, out_tok, out_len, out_score, T, k, beam, nbest, Lcap, blank);
    ASR_CHECK_LAUNCH("asr_ctc_prefix_beam");
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------
// Character error rate on the device (Predictor/Utils/score.py:4-13 with the strings of
// data_handler/vocab.py:75-79): one wave per utterance.  The two strings are assembled in LDS as code points
// (ids != pad, token strings joined by one space), then the Levenshtein rows are computed 64 columns at
// a time: with t[j] = min(prev[j] + 1, prev[j-1] + (a_i != b_j)) the recurrence cur[j] = min(t[j], cur[j-1] + 1)
// is cur[j] - j = prefix-min of (t[k] - k), a wave scan.
namespace {

__device__ __forceinline__ int wave_incl_scan_add(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_scan_min(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v = min(v, u);
    }
    return v;
}

// writes " tok tok tok" (leading space) into s, returns its length
__device__ int cer_build_string(const int32_t* ids, int n, int pad_id, int V, const int32_t* __restrict__ tok_cp, const int32_t* __restrict__ tok_off,
                                int* s, int lane) {
    int base = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const int id = t < n ? ids[t] : pad_id;
        const bool valid = t < n && id != pad_id && id >= 0 && id < V;
        const int o0 = valid ? tok_off[id] : 0;
        const int len = valid ? tok_off[id + 1] - o0 + 1 : 0;
        const int incl = wave_incl_scan_add(len, lane);
        int pos = base + incl - len;
        if (valid) {
            s[pos++] = ' ';
            for (int c = 0; c < len - 1; ++c) s[pos + c] = tok_cp[o0 + c];
        }
        base += __shfl(incl, 63, 64);
    }
    return base;
}

__global__ __launch_bounds__(64) void cer_kernel(const int32_t* __restrict__ hyp, const int32_t* __restrict__ hyp_len, int Lh, int ldh,
                                                 const int32_t* __restrict__ ref, const int32_t* __restrict__ ref_len, int Lr, int ldr,
                                                 const int32_t* __restrict__ tok_cp, const int32_t* __restrict__ tok_off, int V, int pad_id, int cap_a,
                                                 int cap_b, float* __restrict__ per_utt) {
    extern __shared__ int cer_smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    int* sa = cer_smem;              // hypothesis, cap_a code points (with the leading space)
    int* sb = sa + cap_a;            // reference, cap_b
    int* row0 = sb + cap_b;          // cap_b + 1
    int* row1 = row0 + cap_b + 1;
    const int nh = hyp_len ? min(max(hyp_len[b], 0), Lh) : Lh, nr = ref_len ? min(max(ref_len[b], 0), Lr) : Lr;
    int n = cer_build_string(hyp + (size_t)b * ldh, nh, pad_id, V, tok_cp, tok_off, sa, lane);
    int m = cer_build_string(ref + (size_t)b * ldr, nr, pad_id, V, tok_cp, tok_off, sb, lane);
    __syncthreads();
    const int* a = sa + 1;           // drop the leading space
    const int* bb = sb + 1;
    n = n > 0 ? n - 1 : 0;
    m = m > 0 ? m - 1 : 0;
    int spaces = 0;
    for (int j = lane; j < m; j += 64) spaces += bb[j] == ' ';
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) spaces += __shfl_xor(spaces, o, 64);
    for (int j = lane; j <= m; j += 64) row0[j] = j;
    __syncthreads();
    int* prev = row0;
    int* cur = row1;
    for (int i = 1; i <= n; ++i) {
        const int ca = a[i - 1];
        int carry = i;               // cur[0] - 0
        for (int j0 = 1; j0 <= m; j0 += 64) {
            const int j = j0 + lane;
            int v = 0x3fffffff;
            if (j <= m) v = min(prev[j] + 1, prev[j - 1] + (ca != bb[j - 1])) - j;
            v = min(wave_incl_scan_min(v, lane), carry);
            if (j <= m) cur[j] = v + j;
            carry = __shfl(v, 63, 64);
        }
        if (lane == 0) cur[0] = i;
        __syncthreads();
        int* t = prev; prev = cur; cur = t;
    }
    if (lane == 0) per_utt[b] = (float)prev[m] / (float)(spaces + 1);
}

}  // namespace

extern "C" int asr_cer(const int32_t* hyp, const int32_t* hyp_len, int Lh, int ldh, const int32_t* ref, const int32_t* ref_len, int Lr, int ldr,
                       const int32_t* tok_cp, const int32_t* tok_off, int V, int max_tok_len, int pad_id, int B, float* per_utt, void* stream) {
    if (!hyp || !ref || !tok_cp || !tok_off || !per_utt) ASR_FAIL(ASR_EINVAL, "asr_cer: null pointer");
    if (B <= 0 || Lh < 0 || Lr < 0 || ldh < Lh || ldr < Lr || V <= 0 || max_tok_len < 0) ASR_FAIL(ASR_EINVAL, "asr_cer: bad shape B=%d Lh=%d Lr=%d V=%d", B, Lh, Lr, V);
    const long long cap_a = (long long)Lh * (max_tok_len + 1) + 1, cap_b = (long long)Lr * (max_tok_len + 1) + 1;
    const long long bytes = (cap_a + cap_b + 2 * (cap_b + 1)) * 4;
    if (bytes > 150 * 1024) ASR_FAIL(ASR_EINVAL, "asr_cer: strings of up to %lld + %lld code points do not fit the LDS", cap_a, cap_b);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)cer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    cer_kernel<<<B, 64, (size_t)bytes, (hipStream_t)stream>>>(hyp, hyp_len, Lh, ldh, ref, ref_len, Lr, ldr, tok_cp, tok_off, V, pad_id, (int)cap_a, (int)cap_b, per_utt);
    ASR_CHECK_LAUNCH("asr_cer");
    return ASR_OK;
}
