// Fused residual-add + LayerNorm (+PE, + pad-row zeroing), forward and backward.
// HBM-bound: one wave owns one row at a time, every element is read once and written once with
// 16-byte (bf16) / 32-byte (f32) per-lane accesses; row statistics use wave shuffles only.
// Algorithmic bytes per row (d elements of size e): fwd = (x + res read, y + xhat written) = 4*d*e
// (3*d*e without residual); bwd = (dy [+dy2] + xhat read, dz written) = 3*d*e (4*d*e with dy2).
#include <stdlib.h>

#include "asr_common.h"

namespace {

constexpr int LN_WAVES = 4;  // waves (rows in flight) per workgroup
#ifndef LN_UNROLL
#define LN_UNROLL 2
#endif

// per-lane slice of one row: N values. VEC8: value c*8+j is column c*512 + lane*8 + j;
// otherwise value k is column k*64 + lane (masked by col < d).
// eight consecutive elements as they sit in memory (no conversion: a conversion right behind its load makes the load a wait)
template <typename T> struct Pack8;
template <> struct Pack8<bf16_t> { bf16x8 v; };
template <> struct Pack8<float> { f32x4 a, b; };
__device__ __forceinline__ void fetch8(const bf16_t* p, Pack8<bf16_t>& r) { r.v = *(const bf16x8*)p; }
__device__ __forceinline__ void fetch8(const float* p, Pack8<float>& r) { r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 4); }
__device__ __forceinline__ void unpack8(const Pack8<bf16_t>& r, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r.v[i];
}
__device__ __forceinline__ void unpack8(const Pack8<float>& r, float* v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
}

template <typename T, int N, bool VEC8> struct RowSlice {
    // Two-step form of load(): fetch() only issues the loads of a row slice, unpack() converts.  The kernels fetch every operand of
    // every row of a trip before the first unpack: with load() = load + convert per operand (and the optional operand behind a
    // branch) the compiler put a full s_waitcnt vmcnt(0) between the operands of ONE row.
    struct Raw {
        Pack8<T> p[VEC8 ? N / 8 : 1];
        T s[VEC8 ? 1 : N];
    };
    static __device__ __forceinline__ void fetch(const T* p, int d, int lane, Raw& r) {
        if constexpr (VEC8) {
#pragma unroll
            for (int c = 0; c < N / 8; ++c) fetch8(p + c * 512 + lane * 8, r.p[c]);
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int col = k * 64 + lane;
                r.s[k] = p[col < d ? col : 0];
            }
        }
    }
    static __device__ __forceinline__ void unpack(const Raw& r, int d, int lane, float (&v)[N]) {
        if constexpr (VEC8) {
#pragma unroll
            for (int c = 0; c < N / 8; ++c) unpack8(r.p[c], &v[c * 8]);
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = (k * 64 + lane < d) ? to_f32<T>(r.s[k]) : 0.f;
        }
    }
    static __device__ __forceinline__ void load(const T* p, int d, int lane, float (&v)[N]) {
        if constexpr (VEC8) {
#pragma unroll
            for (int c = 0; c < N / 8; ++c) load8<T>(p + c * 512 + lane * 8, *(float(*)[8]) & v[c * 8]);
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                int col = k * 64 + lane;
                v[k] = col < d ? to_f32<T>(p[col]) : 0.f;
            }
        }
    }
    static __device__ __forceinline__ int col(int i, int lane) {
        if constexpr (VEC8) return (i >> 3) * 512 + lane * 8 + (i & 7);
        else return i * 64 + lane;
    }
    // v[i] = keep(row, col) ? v[i] * scale : 0   (dropout mask regenerated from the counter hash)
    static __device__ __forceinline__ void dropout(float (&v)[N], uint32_t row_base, int lane, uint32_t seed, uint32_t thr, float scale) {
        if constexpr (VEC8) {      // a lane's values are runs of 8 consecutive columns: one hash per element PAIR (row_base and the run start are even)
#pragma unroll
            for (int i = 0; i < N; i += 2) {
                const uint32_t h = drop_hash((row_base + (uint32_t)col(i, lane)) >> 1, seed);
                v[i] = drop_keep(h, 0, thr) ? v[i] * scale : 0.f;
                v[i + 1] = drop_keep(h, 1, thr) ? v[i + 1] * scale : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = drop_keep_at(row_base + (uint32_t)col(i, lane), seed, thr) ? v[i] * scale : 0.f;
        }
    }
    static __device__ __forceinline__ void loadf(const float* p, int d, int lane, float (&v)[N]) {
        RowSlice<float, N, VEC8>::load(p, d, lane, v);
    }
    static __device__ __forceinline__ void store(T* p, int d, int lane, const float (&v)[N]) {
        if constexpr (VEC8) {
#pragma unroll
            for (int c = 0; c < N / 8; ++c) store8<T>(p + c * 512 + lane * 8, *(const float(*)[8]) & v[c * 8]);
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                int col = k * 64 + lane;
                if (col < d) p[col] = from_f32<T>(v[k]);
            }
        }
    }
    static __device__ __forceinline__ void storef(float* p, int d, int lane, const float (&v)[N]) {
        RowSlice<float, N, VEC8>::store(p, d, lane, v);
    }
};

// DROP: 0 = none, 1 = dropout on x before the residual add (attention.py:59, module.py:73),
//       2 = dropout on the output after the PE add (transformer_official.py:175-177)
template <typename T, int N, bool VEC8, int DROP>
__global__ __launch_bounds__(LN_WAVES* WAVE) void add_ln_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ pe, const int32_t* __restrict__ lens,
    T* __restrict__ y, T* __restrict__ xhat, float* __restrict__ rstd_out, int rows, int T_, int d,
    uint32_t seed, uint32_t thr, float dscale) {
    using RS = RowSlice<T, N, VEC8>;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: row indices, and with them lens[b], live in scalar registers
    float g[N], bt[N];
    RS::loadf(gamma, d, lane, g);
    RS::loadf(beta, d, lane, bt);
    const float inv_d = 1.f / (float)d;
    // LN_U rows per wave and trip, their loads issued together: one row per wave kept ~32 KB per CU in flight (16 waves x
    // 2 KB), half of what the memory latency needs at this bandwidth (measured 3.7 of ~6 TB/s streaming rate).
    constexpr int LN_U = (N <= 8) ? LN_UNROLL : ((N <= 16) ? 2 : 1);
    const int stride = gridDim.x * LN_WAVES;
    for (int row0 = blockIdx.x * LN_WAVES + w; row0 < rows; row0 += LN_U * stride) {
        float z[LN_U][N], r[LN_U][N];
        {
            typename RS::Raw zx[LN_U], zr[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int row = min(row0 + u * stride, rows - 1);      // a row past the end re-reads the last row (its result is not used)
                RS::fetch(x + (size_t)row * d, d, lane, zx[u]);
                RS::fetch((res ? res : x) + (size_t)row * d, d, lane, zr[u]);      // no residual: the same line again (a cache hit), not used
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                RS::unpack(zx[u], d, lane, z[u]);
                RS::unpack(zr[u], d, lane, r[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < LN_U; ++u) {
            const int row = row0 + u * stride;
            if (row >= rows) break;
            const int b = row / T_, t = row - b * T_;
            if constexpr (DROP == 1) RS::dropout(z[u], (uint32_t)row * (uint32_t)d, lane, seed, thr, dscale);
            if (res) {
#pragma unroll
                for (int i = 0; i < N; ++i) z[u][i] += r[u][i];
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < N; ++i) s += z[u][i];
            const float mean = wave_sum(s) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                // padded lanes of the scalar path hold z = 0: exclude them from the variance
                float c = z[u][i] - mean;
                if constexpr (!VEC8) c = (i * 64 + lane < d) ? c : 0.f;
                z[u][i] = c;
                q += c * c;
            }
            const float rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
            const bool keep = lens ? (t < lens[b]) : true;
            float out[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                z[u][i] *= rstd;
                out[i] = z[u][i] * g[i] + bt[i];
            }
            if (pe) {
                float p[N];
                RS::loadf(pe + (size_t)t * d, d, lane, p);
#pragma unroll
                for (int i = 0; i < N; ++i) out[i] += p[i];
            }
            if constexpr (DROP == 2) RS::dropout(out, (uint32_t)row * (uint32_t)d, lane, seed, thr, dscale);
            if (!keep) {
#pragma unroll
                for (int i = 0; i < N; ++i) out[i] = 0.f;
            }
            RS::store(xhat + (size_t)row * d, d, lane, z[u]);
            RS::store(y + (size_t)row * d, d, lane, out);
            if (lane == 0) rstd_out[row] = rstd;
        }
    }
}

// partial column sums land in ws as [blockIdx][3][d] f32 (the 4 waves are combined through LDS)
template <typename T, int N, bool VEC8, int DROP>
__global__ __launch_bounds__(LN_WAVES* WAVE) void add_ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ xhat,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma,
    const int32_t* __restrict__ lens, T* __restrict__ dz, T* __restrict__ dx, float* __restrict__ ws, int rows, int T_,
    int d, uint32_t seed, uint32_t thr, float dscale) {
    using RS = RowSlice<T, N, VEC8>;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: row indices, and with them lens[b], live in scalar registers
    float g[N], acc_g[N], acc_b[N], acc_z[N];
    RS::loadf(gamma, d, lane, g);
#pragma unroll
    for (int i = 0; i < N; ++i) acc_g[i] = acc_b[i] = acc_z[i] = 0.f;
    const float inv_d = 1.f / (float)d;
    // LN_U rows per wave and trip with their loads issued together (see add_ln_fwd_kernel)
    constexpr int LN_U = (N <= 8) ? LN_UNROLL : ((N <= 16) ? 2 : 1);
    const int stride = gridDim.x * LN_WAVES;
    for (int row0 = blockIdx.x * LN_WAVES + w; row0 < rows; row0 += LN_U * stride) {
        float gy[LN_U][N], xh[LN_U][N], e[LN_U][N], rs[LN_U];
        bool keep[LN_U];
        {
            // the lengths first (scalar loads: row indices are uniform), then every operand of both rows before the first conversion;
            // a padded row (no gradient) costs no memory traffic (see below)
            typename RS::Raw rg[LN_U], re[LN_U], rx[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int row = row0 + u * stride;
                keep[u] = false;
                if (row < rows) {      // wave-uniform
                    const int b = row / T_, t = row - b * T_;
                    keep[u] = lens ? (t < lens[b]) : true;
                }
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                // no branch around the loads (behind `if (keep)` the compiler threads the whole row - loads, conversion, arithmetic -
                // into one arm, one row after the other): a padded row reads row 0 instead, the same line for every such row
                const size_t lrow = keep[u] ? (size_t)(row0 + u * stride) : 0;
                RS::fetch(dy + lrow * d, d, lane, rg[u]);
                RS::fetch((dy2 ? dy2 : dy) + lrow * d, d, lane, re[u]);
                RS::fetch(xhat + lrow * d, d, lane, rx[u]);
                rs[u] = rstd_in[lrow];
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                RS::unpack(rg[u], d, lane, gy[u]);
                RS::unpack(re[u], d, lane, e[u]);
                RS::unpack(rx[u], d, lane, xh[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < LN_U; ++u) {
            const int row = row0 + u * stride;
            if (row >= rows) break;
            float o[N];
            if (keep[u]) {
                if (dy2) {
#pragma unroll
                    for (int i = 0; i < N; ++i) gy[u][i] += e[u][i];
                }
                if constexpr (DROP == 2) RS::dropout(gy[u], (uint32_t)row * (uint32_t)d, lane, seed, thr, dscale);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    acc_g[i] += gy[u][i] * xh[u][i];
                    acc_b[i] += gy[u][i];
                    gy[u][i] *= g[i];
                    s1 += gy[u][i];
                    s2 += gy[u][i] * xh[u][i];
                }
                s1 = wave_sum(s1) * inv_d;
                s2 = wave_sum(s2) * inv_d;
                const float rstd = rs[u];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    o[i] = rstd * (gy[u][i] - s1 - xh[u][i] * s2);
                    if constexpr (!VEC8) o[i] = (i * 64 + lane < d) ? o[i] : 0.f;
                    if constexpr (DROP != 1) acc_z[i] += o[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) o[i] = 0.f;
            }
            RS::store(dz + (size_t)row * d, d, lane, o);
            if constexpr (DROP == 1) {   // gradient wrt the pre-dropout GEMM output (and its bias)
                RS::dropout(o, (uint32_t)row * (uint32_t)d, lane, seed, thr, dscale);
#pragma unroll
                for (int i = 0; i < N; ++i) acc_z[i] += o[i];
                RS::store(dx + (size_t)row * d, d, lane, o);
            }
        }
    }
    // LN_WAVES x d floats of dynamic LDS (8 KiB at d = 512): a fixed 32-KiB array left room for only
    // one of these workgroups per CU beside the weight-gradient GEMM's 128-KiB ring on the other stream
    extern __shared__ __attribute__((aligned(16))) float sred[];
    float* slot = ws + (size_t)blockIdx.x * 3 * d;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        RS::storef(sred + (size_t)w * d, d, lane, k == 0 ? acc_g : (k == 1 ? acc_b : acc_z));
        __syncthreads();
        for (int c = threadIdx.x; c < d; c += LN_WAVES * WAVE) slot[k * d + c] = sred[c] + sred[d + c] + sred[2 * d + c] + sred[3 * d + c];
        __syncthreads();
    }
}

}  // namespace

// forward: 1024 workgroups measured best (5.15 TB/s vs 3.9 at 512); backward: its grid is also the
// number of partial rows the finalize pass has to sum
static int ln_grid_fwd(int rows) {
    const int cap = 1024;
    int g = ceil_div(rows, LN_WAVES);
    return g < cap ? g : cap;
}
static int ln_grid(int rows) {
    // round 1 (one row per wave and trip), stand-alone: 256 -> 29.7, 512 -> 21.3, 1024 -> 19.4, 2048 -> 23.5 us.  With two rows per
    // trip the step decides (tools/env_sweep.sh, ms per step): 256 -> 3.49, 384 -> 3.39, 512 -> 3.35, 1024 -> 3.38, 2048 -> 3.54
    // (half the partial sums to write and to reduce, and the kernel holds fewer CUs beside the weight-gradient stream)
    const int cap = 512;
    int g = ceil_div(rows, LN_WAVES);
    return g < cap ? g : cap;
}

extern "C" size_t asr_add_ln_bwd_workspace_bytes(int rows, int d) {
    return (size_t)ln_grid(rows) * 3 * d * sizeof(float);
}

template <typename T, int DROP>
static int launch_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                         const float* pe, const int32_t* lens, void* y, void* xhat, float* rstd,
                         int rows, int T_, int d, uint32_t seed, uint32_t thr, float dscale, hipStream_t st) {
    const int grid = ln_grid_fwd(rows);
#define LN_FWD(N, V)                                                                              \
    add_ln_fwd_kernel<T, N, V, DROP><<<grid, LN_WAVES * WAVE, 0, st>>>(                            \
        (const T*)x, (const T*)res, gamma, beta, pe, lens, (T*)y, (T*)xhat, rstd, rows, T_, d, seed, thr, dscale)
    if (d % 512 == 0 && d <= 2048) {
        switch (d / 512) {
            case 1: LN_FWD(8, true); break;
            case 2: LN_FWD(16, true); break;
            case 3: LN_FWD(24, true); break;
            default: LN_FWD(32, true); break;
        }
    } else {
        const int nk = ceil_div(d, 64);
        if (nk <= 1) LN_FWD(1, false);
        else if (nk <= 2) LN_FWD(2, false);
        else if (nk <= 4) LN_FWD(4, false);
        else if (nk <= 8) LN_FWD(8, false);
        else if (nk <= 16) LN_FWD(16, false);
        else LN_FWD(32, false);
    }
#undef LN_FWD
    return 0;
}

extern "C" int asr_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                              const float* pe, const int32_t* lens, void* y, void* xhat,
                              float* rstd, int B, int T, int d, float drop_p, uint32_t drop_seed, int drop_mode,
                              int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !xhat || !rstd) ASR_FAIL(ASR_EINVAL, "asr_add_ln_fwd: null pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d > 2048) ASR_FAIL(ASR_EINVAL, "asr_add_ln_fwd: bad shape B=%d T=%d d=%d", B, T, d);
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && drop_mode != 1 && drop_mode != 2) || (drop_p > 0.f && (d & 1)))
        ASR_FAIL(ASR_EINVAL, "asr_add_ln_fwd: bad dropout p=%f mode=%d (d must be even)", drop_p, drop_mode);
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_add_ln_fwd: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int mode = drop_p > 0.f ? drop_mode : 0;
    const uint32_t thr = drop_thr16(drop_p);
    const float ds = 1.f / (1.f - drop_p);
#define LN_FWD_D(TT, MODE) launch_ln_fwd<TT, MODE>(x, res, gamma, beta, pe, lens, y, xhat, rstd, B * T, T, d, drop_seed, thr, ds, st)
    if (dtype == ASR_F32) { if (mode == 0) LN_FWD_D(float, 0); else if (mode == 1) LN_FWD_D(float, 1); else LN_FWD_D(float, 2); }
    else { if (mode == 0) LN_FWD_D(bf16_t, 0); else if (mode == 1) LN_FWD_D(bf16_t, 1); else LN_FWD_D(bf16_t, 2); }
#undef LN_FWD_D
    ASR_CHECK_LAUNCH("asr_add_ln_fwd");
    return ASR_OK;
}

template <typename T, int DROP>
static void launch_ln_bwd(const void* dy, const void* dy2, const void* xhat, const float* rstd,
                          const float* gamma, const int32_t* lens, void* dz, void* dx, float* ws, int rows,
                          int T_, int d, uint32_t seed, uint32_t thr, float dscale, hipStream_t st) {
    const int grid = ln_grid(rows);
    // the partial-sum form (no dgamma: the sums stay in ws) ends with this kernel: it may carry an armed completion event (asr_stream_arm)
#define LN_BWD(N, V)                                                                      \
    asr_launch_armed(add_ln_bwd_kernel<T, N, V, DROP>, dim3(grid), dim3(LN_WAVES * WAVE), (size_t)LN_WAVES * d * sizeof(float), st,    \
        (const T*)dy, (const T*)dy2, (const T*)xhat, rstd, gamma, lens, (T*)dz, (T*)dx, ws, rows, T_, d, seed, thr, dscale)
    if (d % 512 == 0 && d <= 2048) {
        switch (d / 512) {
            case 1: LN_BWD(8, true); break;
            case 2: LN_BWD(16, true); break;
            case 3: LN_BWD(24, true); break;
            default: LN_BWD(32, true); break;
        }
    } else {
        const int nk = ceil_div(d, 64);
        if (nk <= 1) LN_BWD(1, false);
        else if (nk <= 2) LN_BWD(2, false);
        else if (nk <= 4) LN_BWD(4, false);
        else if (nk <= 8) LN_BWD(8, false);
        else if (nk <= 16) LN_BWD(16, false);
        else LN_BWD(32, false);
    }
#undef LN_BWD
}

extern "C" int asr_add_ln_bwd(const void* dy, const void* dy2, const void* xhat, const float* rstd,
                              const float* gamma, const int32_t* lens, void* dz, void* dx, float* dgamma,
                              float* dbeta, float* dbias, void* ws, size_t ws_bytes, int B, int T,
                              int d, float drop_p, uint32_t drop_seed, int drop_mode, int dtype, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !dz || !ws || (!dgamma != !dbeta)) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd: null pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d > 2048) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd: bad shape B=%d T=%d d=%d", B, T, d);
    const int rows = B * T;
    if (ws_bytes < asr_add_ln_bwd_workspace_bytes(rows, d)) ASR_FAIL(ASR_EWORKSPACE, "asr_add_ln_bwd: workspace %zu < %zu", ws_bytes, asr_add_ln_bwd_workspace_bytes(rows, d));
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_add_ln_bwd: dtype %d", dtype);
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && drop_mode != 1 && drop_mode != 2) || (drop_p > 0.f && (d & 1)))
        ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd: bad dropout p=%f mode=%d", drop_p, drop_mode);
    const int mode = drop_p > 0.f ? drop_mode : 0;
    if (mode == 1 && !dx) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd: pre-residual dropout needs the dx output");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_thr16(drop_p);
    const float ds = 1.f / (1.f - drop_p);
#define LN_BWD_D(TT, MODE) launch_ln_bwd<TT, MODE>(dy, dy2, xhat, rstd, gamma, lens, dz, dx, (float*)ws, rows, T, d, drop_seed, thr, ds, st)
    if (dtype == ASR_F32) { if (mode == 0) LN_BWD_D(float, 0); else if (mode == 1) LN_BWD_D(float, 1); else LN_BWD_D(float, 2); }
    else { if (mode == 0) LN_BWD_D(bf16_t, 0); else if (mode == 1) LN_BWD_D(bf16_t, 1); else LN_BWD_D(bf16_t, 2); }
#undef LN_BWD_D
    const int P = ln_grid(rows);
    const int ncols = dbias ? 3 * d : 2 * d;
    if (dgamma) {   // dgamma == dbeta == NULL: the partial sums stay in ws for asr_add_ln_bwd_reduce_batched
        // the partial rows are split between 4 workgroups per column group (atomic adds): 3.806 vs 3.830 ms per step
        const int fsplit = 4;
        colsum_finalize_kernel<<<dim3(ceil_div(ncols, 32), (P >= 256 * fsplit && !asr_deterministic()) ? fsplit : 1), 1024, 0, st>>>((const float*)ws, P, (size_t)3 * d, ncols, d, dgamma, dbeta, dbias, 1);
    }
    ASR_CHECK_LAUNCH("asr_add_ln_bwd");
    return ASR_OK;
}

// One launch for the parameter-gradient reductions of several LayerNorm sites (blockIdx.z = site): every launch
// costs ~4-5 us of dispatch and drain on top of its work, and backward has 13 of these reductions per step.
namespace {
struct LnReduceBatch {
    const float* part[ASR_LN_REDUCE_MAX];
    float* dgamma[ASR_LN_REDUCE_MAX];
    float* dbeta[ASR_LN_REDUCE_MAX];
    float* dbias[ASR_LN_REDUCE_MAX];
    int P[ASR_LN_REDUCE_MAX];
};
__global__ __launch_bounds__(1024) void ln_reduce_batched_kernel(const LnReduceBatch bt, int d) {
    __shared__ float red[32][33];
    const int it = blockIdx.z, P = bt.P[it];
    const float* __restrict__ part = bt.part[it];
    const int ncols = bt.dbias[it] ? 3 * d : 2 * d;
    const int cg = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cg;
    const int ns = P >= 1024 ? (int)gridDim.y : 1;   // few partial rows: one workgroup per column group, plain (deterministic) adds
    if (blockIdx.x * 32 >= ncols || (int)blockIdx.y >= ns) return;
    const int per = (P + ns - 1) / ns, pbeg = blockIdx.y * per, pend = min(P, pbeg + per);
    const size_t pstride = (size_t)3 * d;
    float s = 0.f;
    if (col < ncols) {
        int p = pbeg + pg;
        for (; p + 7 * 32 < pend; p += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u * 32) * pstride + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < pend; p += 32) s += part[(size_t)p * pstride + col];
    }
    red[pg][cg] = s;
    __syncthreads();
    if (pg == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][cg];
        const int k = col / d, c = col - k * d;
        float* out = k == 0 ? bt.dgamma[it] : (k == 1 ? bt.dbeta[it] : bt.dbias[it]);
        if (ns > 1) atomicAdd(out + c, t);
        else out[c] += t;
    }
}
}  // namespace

extern "C" int asr_add_ln_bwd_reduce_batched(const asr_ln_reduce_item* items, int n, int d, void* stream) {
    if (!items || n <= 0 || n > ASR_LN_REDUCE_MAX) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd_reduce_batched: 1..%d items per call (got %d)", ASR_LN_REDUCE_MAX, n);
    if (d <= 0 || d > 2048) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd_reduce_batched: bad d=%d", d);
    LnReduceBatch bt;
    memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < n; ++i) {
        if (!items[i].ws || !items[i].dgamma || !items[i].dbeta || items[i].rows <= 0) ASR_FAIL(ASR_EINVAL, "asr_add_ln_bwd_reduce_batched: bad item %d", i);
        bt.part[i] = (const float*)items[i].ws;
        bt.dgamma[i] = items[i].dgamma;
        bt.dbeta[i] = items[i].dbeta;
        bt.dbias[i] = items[i].dbias;
        bt.P[i] = ln_grid(items[i].rows);
    }
    ln_reduce_batched_kernel<<<dim3(ceil_div(3 * d, 32), asr_deterministic() ? 1 : 4, n), 1024, 0, (hipStream_t)stream>>>(bt, d);
    ASR_CHECK_LAUNCH("asr_add_ln_bwd_reduce_batched");
    return ASR_OK;
}
