// CTC forward-backward fused with the log-softmax over the vocabulary, and softmax
// cross-entropy.  Both are HBM-bound passes over (rows, V) logits.
//
// CTC, three kernels on one stream:
//   1. ctc_lse_gather : one wave per frame (b,t < in_len[b]).  Streams the V logits once
//      (16 B per lane per load), online max/sum-exp per lane, wave-shuffle reduction -> lse[b,t];
//      then gathers the S = 2L+1 lattice inputs lp[b,t,s] = x[l'_s] - lse (row is L1/L2-hot).
//   2. ctc_alpha_beta : one workgroup of two waves per utterance.  Wave 0 runs the alpha
//      recursion forward in time, wave 1 the beta recursion backward, state s on lane s%64
//      (register j = s/64), predecessors via DPP/shuffle (no LDS in the recursion), lattice
//      inputs prefetched 2 x 8 timesteps ahead in registers so the serial chain never waits on
//      memory.  Both are renormalised by their running maximum every 8 steps (offsets summed
//      for the loss), so fp32 round-off does not grow with T.
//   3. ctc_grad       : one workgroup per frame.  Wave 0 forms the frame's state posteriors
//      softmax_s(alpha+beta-lp) (exact: the sum over s is p(l|x) at every t) and scatters them into an LDS
//      table indexed by label (ds_add_f32), then one streaming pass: read logits, write
//      dlogits = scale * (softmax - posterior).  Padded frames are written as zeros.
// Algorithmic HBM bytes: logits read twice + dlogits written once = 3 * B*T*V*e;
// the lattice (B*T*S*4 B * 3 arrays) stays in L2 / Infinity Cache.
#include "asr_common.h"

namespace {

constexpr float NEG_INF = -INFINITY;

template <typename T> struct Vec {
    static constexpr int N = 16 / sizeof(T);  // elements per 16-byte load
};

template <typename T>
__device__ __forceinline__ void loadv(const T* p, float (&r)[Vec<T>::N]) {
    if constexpr (sizeof(T) == 2) {
        load8<T>(p, r);
    } else {
        f32x4 v = *(const f32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i];
    }
}
template <typename T>
__device__ __forceinline__ void storev(T* p, const float (&r)[Vec<T>::N]) {
    if constexpr (sizeof(T) == 2) {
        store8<T>(p, r);
    } else {
        f32x4 v = {r[0], r[1], r[2], r[3]};
        *(f32x4*)p = v;
    }
}

// online (max, sum exp(x - max)) merge
__device__ __forceinline__ void ms_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == NEG_INF) { m = mn; s = 0.f; return; }
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

// lse of one row held by a whole wave
template <typename T>
__device__ __forceinline__ float wave_row_lse(const T* __restrict__ x, int V, int lane, float* sum_x) {
    constexpr int N = Vec<T>::N;
    float m = NEG_INF, s = 0.f, sx = 0.f;
    if (V % N == 0) {
        const int nv = V / N;
        for (int i = lane; i < nv; i += 64) {
            float v[N];
            loadv<T>(x + (size_t)i * N, v);
            float lm = v[0];
#pragma unroll
            for (int j = 1; j < N; ++j) lm = fmaxf(lm, v[j]);
            const float mn = fmaxf(m, lm);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) { acc += expf(v[j] - mn); sx += v[j]; }
            s = s * expf(m - mn) + acc;
            m = mn;
        }
    } else {
        for (int i = lane; i < V; i += 64) {
            const float v = to_f32<T>(x[i]);
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
            sx += v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        ms_merge(m, s, m2, s2);
        sx += __shfl_xor(sx, o, 64);
    }
    if (sum_x) *sum_x = sx;
    return m + logf(s);
}

// ---------------------------------------------------------------------------------- kernel 1
template <typename T>
__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const T* __restrict__ logits, const int32_t* __restrict__ in_len,
                                                             const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                             float* __restrict__ lp, float* __restrict__ lse_out, int B, int T_, int V, int Lmax,
                                                             int Smax, int blank) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) continue;
        const T* x = logits + (size_t)row * V;
        const float lse = wave_row_lse<T>(x, V, lane, nullptr);
        if (lane == 0) lse_out[row] = lse;
        const int L = lab_len[b], S = 2 * L + 1;
        float* out = lp + (size_t)row * Smax;
        for (int s = lane; s < S; s += 64) {
            const int c = (s & 1) ? labels[(size_t)b * Lmax + (s >> 1)] : blank;
            out[s] = to_f32<T>(x[c]) - lse;
        }
    }
}

// ---------------------------------------------------------------------------------- kernel 2
// whole-wave shifts by one lane through DPP (no LDS round trip as ds_bpermute would need): the
// lane that has no source keeps -inf.
__device__ __forceinline__ float wave_shr1(float v) {   // lane i <- lane i-1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NEG_INF), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {   // lane i <- lane i+1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NEG_INF), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
// NS = states per lane (S <= 64*NS).  dir = 0: alpha (wave 0), 1: beta (wave 1).
template <int NS>
__device__ __forceinline__ float ctc_recursion(const float* __restrict__ lp, float* __restrict__ out, const int32_t* __restrict__ lab,
                                               int Tb, int L, int Smax, int blank, int lane, bool backward) {
    const int S = 2 * L + 1;
    float off = 0.f;  // sum of the maxima removed so far: true log value = stored value + off
    // transition permissions per owned state
    bool can2[NS];   // alpha: s-2 -> s allowed ; beta: s+2 -> s allowed
    bool live[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int s = lane + 64 * j;
        live[j] = s < S;
        bool c = false;
        if (!backward) {
            if (live[j] && (s & 1) && s >= 2) c = lab[s >> 1] != lab[(s >> 1) - 1];
        } else {
            if (s + 2 < S && (s & 1)) c = lab[s >> 1] != lab[(s >> 1) + 1];
        }
        can2[j] = c;
    }
    constexpr int CH = 8;  // timesteps per prefetch chunk
    float a[NS];
    float buf0[CH][NS], buf1[CH][NS];
    auto fetch = [&](float (&dst)[CH][NS], int chunk) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int step = chunk * CH + k;
            const int t = backward ? Tb - 1 - step : step;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int s = lane + 64 * j;
                dst[k][j] = (step < Tb && live[j]) ? lp[(size_t)t * Smax + s] : NEG_INF;
            }
        }
    };
    auto advance = [&](const float (&cur)[CH][NS], int chunk) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int step = chunk * CH + k;
            if (step >= Tb) break;
            const int t = backward ? Tb - 1 - step : step;
            if (k == 0 && chunk > 0) {
                // renormalise once per chunk: keeps |a| small, so fp32 round-off does not grow with T.
                // Any per-timestep constant cancels in the posterior (it is normalised over s).
                float mx = NEG_INF;
#pragma unroll
                for (int j = 0; j < NS; ++j) mx = fmaxf(mx, a[j]);
                mx = wave_max(mx);
                if (mx != NEG_INF) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) a[j] -= mx;
                    off += mx;
                }
            }
            if (step == 0) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const int s = lane + 64 * j;
                    const bool init = backward ? (s == S - 1 || s == S - 2) : (s == 0 || s == 1);
                    a[j] = (init && live[j]) ? cur[k][j] : NEG_INF;
                }
            } else {
                float n1[NS], n2[NS];
                if (!backward) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        float u1 = wave_shr1(a[j]);
                        float u2 = wave_shr1(u1);
                        // lanes 0/1 take their predecessors from the previous register's top lanes
                        const float p1 = j > 0 ? lane_bcast(a[j > 0 ? j - 1 : 0], 63) : NEG_INF;
                        const float p2a = j > 0 ? lane_bcast(a[j > 0 ? j - 1 : 0], 62) : NEG_INF;
                        if (lane == 0) { u1 = p1; u2 = p2a; }
                        if (lane == 1) { u2 = p1; }
                        n1[j] = u1;
                        n2[j] = can2[j] ? u2 : NEG_INF;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        float u1 = wave_shl1(a[j]);
                        float u2 = wave_shl1(u1);
                        const float p1 = j + 1 < NS ? lane_bcast(a[j + 1 < NS ? j + 1 : j], 0) : NEG_INF;
                        const float p2a = j + 1 < NS ? lane_bcast(a[j + 1 < NS ? j + 1 : j], 1) : NEG_INF;
                        if (lane == 63) { u1 = p1; u2 = p2a; }
                        if (lane == 62) { u2 = p1; }
                        n1[j] = u1;
                        n2[j] = can2[j] ? u2 : NEG_INF;
                    }
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const float m = fmaxf(fmaxf(a[j], n1[j]), n2[j]);
                    float r = NEG_INF;
                    // hardware exp2/log2 (1 ulp): the recursion is renormalised every 8 steps, so the
                    // absolute log-domain error stays ~1e-6 per step
                    if (m != NEG_INF) r = m + __logf(__expf(a[j] - m) + __expf(n1[j] - m) + __expf(n2[j] - m)) + cur[k][j];
                    a[j] = live[j] ? r : NEG_INF;
                }
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int s = lane + 64 * j;
                if (live[j]) out[(size_t)t * Smax + s] = a[j];
            }
        }
    };
    const int nchunks = (Tb + CH - 1) / CH;
    fetch(buf0, 0);
    for (int c = 0; c < nchunks; c += 2) {
        fetch(buf1, c + 1);
        advance(buf0, c);
        fetch(buf0, c + 2);
        advance(buf1, c + 1);
    }
    return off;
}

template <int NS>
__global__ __launch_bounds__(128) void ctc_alpha_beta_kernel(const float* __restrict__ lp, float* __restrict__ alpha, float* __restrict__ beta,
                                                             const int32_t* __restrict__ in_len, const int32_t* __restrict__ labels,
                                                             const int32_t* __restrict__ lab_len, float* __restrict__ nll_out, float* __restrict__ nll_raw,
                                                             int T_, int Lmax, int Smax, int blank, int zero_infinity) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Tb = in_len[b], L = lab_len[b], S = 2 * L + 1;
    const float* lpb = lp + (size_t)b * T_ * Smax;
    float* ab = alpha + (size_t)b * T_ * Smax;
    float* bb = beta + (size_t)b * T_ * Smax;
    const int32_t* lab = labels + (size_t)b * Lmax;
    __shared__ float s_off;
    float off = 0.f;
    if (Tb > 0) off = ctc_recursion<NS>(lpb, w == 0 ? ab : bb, lab, Tb, L, Smax, blank, lane, w == 1);
    if (threadIdx.x == 0) s_off = off;  // wave 0 = alpha
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        float nll;
        if (Tb <= 0) {
            nll = (L == 0) ? 0.f : INFINITY;
        } else {
            const float* last = ab + (size_t)(Tb - 1) * Smax;
            const float l1 = last[S - 1], l2 = S > 1 ? last[S - 2] : NEG_INF;
            const float m = fmaxf(l1, l2);
            nll = (m == NEG_INF) ? INFINITY : -(s_off + m + logf(expf(l1 - m) + expf(l2 - m)));
        }
        nll_raw[b] = nll;
        nll_out[b] = (nll == INFINITY && zero_infinity) ? 0.f : nll;
    }
}

// ---------------------------------------------------------------------------------- kernel 3
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ logits, T* __restrict__ dlogits, const float* __restrict__ lp,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ lse_in, const int32_t* __restrict__ in_len,
                                                       const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                       const float* __restrict__ nll_raw, int B, int T_, int V, int Lmax, int Smax, int blank,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float occ[];  // V floats: posterior mass per label
    constexpr int N = Vec<T>::N;
    const int rows = B * T_;
    for (int i = threadIdx.x; i < V; i += 256) occ[i] = 0.f;
    __syncthreads();
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / T_, t = row - b * T_;
        T* dl = dlogits + (size_t)row * V;
        // padded frame, or infeasible utterance (nll = +inf): zero gradient
        if (t >= in_len[b] || nll_raw[b] == INFINITY) {
            if (V % N == 0) {
                float z[N];
#pragma unroll
                for (int j = 0; j < N; ++j) z[j] = 0.f;
                for (int i = threadIdx.x; i < V / N; i += 256) storev<T>(dl + (size_t)i * N, z);
            } else {
                for (int i = threadIdx.x; i < V; i += 256) dl[i] = from_f32<T>(0.f);
            }
            continue;
        }
        const T* x = logits + (size_t)row * V;
        const int L = lab_len[b], S = 2 * L + 1;
        const float lse = lse_in[row];
        if (threadIdx.x < 64) {
            // log posterior of state s at this frame, up to a per-frame constant: alpha+beta-lp.
            // sum_s alpha_t(s) beta_t(s) / y_t(l'_s) = p(l|x) for EVERY t, so normalising over s is
            // exact and needs neither nll nor the recursion offsets (no large-number cancellation).
            const size_t o = (size_t)row * Smax;
            float mx = NEG_INF;
            for (int si = threadIdx.x; si < S; si += 64) mx = fmaxf(mx, alpha[o + si] + beta[o + si] - lp[o + si]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int si = threadIdx.x; si < S; si += 64) sum += expf(alpha[o + si] + beta[o + si] - lp[o + si] - mx);
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            for (int si = threadIdx.x; si < S; si += 64) {
                const int c = (si & 1) ? labels[(size_t)b * Lmax + (si >> 1)] : blank;
                atomicAdd(&occ[c], expf(alpha[o + si] + beta[o + si] - lp[o + si] - mx) * inv);
            }
        }
        __syncthreads();
        if (V % N == 0) {
            for (int i = threadIdx.x; i < V / N; i += 256) {
                float v[N];
                loadv<T>(x + (size_t)i * N, v);
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] = scale * (expf(v[j] - lse) - occ[i * N + j]);
                storev<T>(dl + (size_t)i * N, v);
            }
        } else {
            for (int i = threadIdx.x; i < V; i += 256) dl[i] = from_f32<T>(scale * (expf(to_f32<T>(x[i]) - lse) - occ[i]));
        }
        __syncthreads();
        for (int si = threadIdx.x; si < S; si += 256) {
            const int c = (si & 1) ? labels[(size_t)b * Lmax + (si >> 1)] : blank;
            occ[c] = 0.f;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------- xent
template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(const T* __restrict__ logits, const int32_t* __restrict__ gold, const float* __restrict__ n_valid,
                                                   float* __restrict__ row_nll, T* __restrict__ dlogits, int M, int V, int ignore_index,
                                                   float smoothing, float grad_scale) {
    __shared__ float red[16];
    constexpr int N = Vec<T>::N;
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * V;
    const int g = gold[row];
    const bool ignored = (g == ignore_index) || g < 0 || g >= V;
    T* dl = dlogits ? dlogits + (size_t)row * V : nullptr;
    if (ignored) {
        if (threadIdx.x == 0) row_nll[row] = 0.f;
        if (dl) for (int i = threadIdx.x; i < V; i += 256) dl[i] = from_f32<T>(0.f);
        return;
    }
    float m = NEG_INF, s = 0.f, sx = 0.f;
    const bool vec = (V % N == 0);
    if (vec) {
        for (int i = threadIdx.x; i < V / N; i += 256) {
            float v[N];
            loadv<T>(x + (size_t)i * N, v);
            float lm = v[0];
#pragma unroll
            for (int j = 1; j < N; ++j) lm = fmaxf(lm, v[j]);
            const float mn = fmaxf(m, lm);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) { acc += expf(v[j] - mn); sx += v[j]; }
            s = s * expf(m - mn) + acc;
            m = mn;
        }
    } else {
        for (int i = threadIdx.x; i < V; i += 256) {
            const float v = to_f32<T>(x[i]);
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
            sx += v;
        }
    }
    const float gm = block_max(m, red);
    const float gs = block_sum(gm == NEG_INF ? 0.f : s * expf(m - gm), red);
    const float lse = gm + logf(gs);
    const float xg = to_f32<T>(x[g]);
    float q_gold = 1.f, q_other = 0.f;
    float loss = lse - xg;
    if (smoothing > 0.f) {  // Utils/loss.py:30-45
        const float tot_x = block_sum(sx, red);
        q_gold = 1.f - smoothing;
        q_other = smoothing / (float)V;
        const float sum_logp = tot_x - (float)V * lse;
        loss = -(q_gold * (xg - lse) + q_other * (sum_logp - (xg - lse)));
    }
    if (threadIdx.x == 0) row_nll[row] = loss;
    if (!dl) return;
    const float qsum = q_gold + q_other * (float)(V - 1);
    const float sc = grad_scale / *n_valid;
    if (vec) {
        for (int i = threadIdx.x; i < V / N; i += 256) {
            float v[N];
            loadv<T>(x + (size_t)i * N, v);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float q = (i * N + j == g) ? q_gold : q_other;
                v[j] = sc * (qsum * expf(v[j] - lse) - q);
            }
            storev<T>(dl + (size_t)i * N, v);
        }
    } else {
        for (int i = threadIdx.x; i < V; i += 256) {
            const float q = (i == g) ? q_gold : q_other;
            dl[i] = from_f32<T>(sc * (qsum * expf(to_f32<T>(x[i]) - lse) - q));
        }
    }
}

static inline int smax_of(int Lmax) { return (2 * Lmax + 1 + 3) & ~3; }

}  // namespace

extern "C" size_t asr_ctc_workspace_bytes(int B, int T, int Lmax) {
    return ((size_t)3 * B * T * smax_of(Lmax) + (size_t)B * T + (size_t)B) * sizeof(float);
}

extern "C" int asr_ctc_fwd_bwd(const void* logits, void* dlogits, const int32_t* in_len, const int32_t* labels, const int32_t* lab_len,
                               float* nll, int B, int T, int V, int Lmax, int blank, float grad_scale, int zero_infinity, void* ws,
                               size_t ws_bytes, int dtype, void* stream) {
    if (!logits || !in_len || !labels || !lab_len || !nll || !ws) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: null pointer");
    if (B <= 0 || T <= 0 || V <= 1 || Lmax <= 0 || blank < 0 || blank >= V) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: bad shape B=%d T=%d V=%d Lmax=%d blank=%d", B, T, V, Lmax, blank);
    if (2 * Lmax + 1 > 1024) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: 2*Lmax+1 = %d > 1024 lattice states", 2 * Lmax + 1);
    if (ws_bytes < asr_ctc_workspace_bytes(B, T, Lmax)) ASR_FAIL(ASR_EWORKSPACE, "asr_ctc_fwd_bwd: workspace %zu < %zu", ws_bytes, asr_ctc_workspace_bytes(B, T, Lmax));
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_ctc_fwd_bwd: dtype %d", dtype);
    const size_t lds = (size_t)((V + 3) & ~3) * sizeof(float);
    if (dlogits && lds > 160 * 1024) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: V=%d does not fit the LDS posterior table", V);
    hipStream_t st = (hipStream_t)stream;
    const int Smax = smax_of(Lmax);
    float* lp = (float*)ws;
    float* alpha = lp + (size_t)B * T * Smax;
    float* beta = alpha + (size_t)B * T * Smax;
    float* lse = beta + (size_t)B * T * Smax;
    float* nll_raw = lse + (size_t)B * T;
    const int rows = B * T;
    int g1 = ceil_div(rows, 4);
    if (g1 > 4096) g1 = 4096;
    if (dtype == ASR_F32) ctc_lse_gather_kernel<float><<<g1, 256, 0, st>>>((const float*)logits, in_len, labels, lab_len, lp, lse, B, T, V, Lmax, Smax, blank);
    else ctc_lse_gather_kernel<bf16_t><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, Lmax, Smax, blank);
    const int NS = ceil_div(2 * Lmax + 1, 64);
#define AB(N) ctc_alpha_beta_kernel<N><<<B, 128, 0, st>>>(lp, alpha, beta, in_len, labels, lab_len, nll, nll_raw, T, Lmax, Smax, blank, zero_infinity)
    if (NS <= 1) AB(1);
    else if (NS <= 2) AB(2);
    else if (NS <= 4) AB(4);
    else if (NS <= 8) AB(8);
    else AB(16);
#undef AB
    if (dlogits) {
        int g3 = rows < 2048 ? rows : 2048;
        if (dtype == ASR_F32) ctc_grad_kernel<float><<<g3, 256, lds, st>>>((const float*)logits, (float*)dlogits, lp, alpha, beta, lse, in_len, labels, lab_len, nll_raw, B, T, V, Lmax, Smax, blank, grad_scale);
        else ctc_grad_kernel<bf16_t><<<g3, 256, lds, st>>>((const bf16_t*)logits, (bf16_t*)dlogits, lp, alpha, beta, lse, in_len, labels, lab_len, nll_raw, B, T, V, Lmax, Smax, blank, grad_scale);
    }
    ASR_CHECK_LAUNCH("asr_ctc_fwd_bwd");
    return ASR_OK;
}

extern "C" int asr_xent_fwd_bwd(const void* logits, const int32_t* gold, const float* n_valid, float* row_nll, void* dlogits, int M,
                                int V, int ignore_index, float smoothing, float grad_scale, int dtype, void* stream) {
    if (!logits || !gold || !row_nll || (dlogits && !n_valid)) ASR_FAIL(ASR_EINVAL, "asr_xent_fwd_bwd: null pointer");
    if (M <= 0 || V <= 1) ASR_FAIL(ASR_EINVAL, "asr_xent_fwd_bwd: bad shape M=%d V=%d", M, V);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_F32) xent_kernel<float><<<M, 256, 0, st>>>((const float*)logits, gold, n_valid, row_nll, (float*)dlogits, M, V, ignore_index, smoothing, grad_scale);
    else if (dtype == ASR_BF16) xent_kernel<bf16_t><<<M, 256, 0, st>>>((const bf16_t*)logits, gold, n_valid, row_nll, (bf16_t*)dlogits, M, V, ignore_index, smoothing, grad_scale);
    else ASR_FAIL(ASR_EDTYPE, "asr_xent_fwd_bwd: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_xent_fwd_bwd");
    return ASR_OK;
}
