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
                                                           int B, int T_, int V, int blank) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) {
            if (lane == 0) path[row] = blank;
            continue;
        }
        const T* x = logits + (size_t)row * V;
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

extern "C" int asr_ctc_greedy_decode(const void* logits, const int32_t* in_len, int32_t* out_ids, int32_t* out_len, int B, int T, int V, int blank,
                                     int dtype, void* stream) {
    if (!logits || !in_len || !out_ids || !out_len) ASR_FAIL(ASR_EINVAL, "asr_ctc_greedy_decode: null pointer");
    if (B <= 0 || T <= 0 || V <= 1 || blank < 0 || blank >= V) ASR_FAIL(ASR_EINVAL, "asr_ctc_greedy_decode: bad shape B=%d T=%d V=%d blank=%d", B, T, V, blank);
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_ctc_greedy_decode: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * T;
    int g = ceil_div(rows, 4);
    if (g > 4096) g = 4096;
    if (dtype == ASR_F32) frame_argmax_kernel<float><<<g, 256, 0, st>>>((const float*)logits, in_len, out_ids, B, T, V, blank);
    else frame_argmax_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)logits, in_len, out_ids, B, T, V, blank);
    ctc_collapse_kernel<<<B, 64, 0, st>>>(out_ids, in_len, out_len, T, blank);
    ASR_CHECK_LAUNCH("asr_ctc_greedy_decode");
    return ASR_OK;
}
