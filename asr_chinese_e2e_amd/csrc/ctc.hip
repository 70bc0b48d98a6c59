// CTC forward-backward fused with the log-softmax over the vocabulary, and softmax
// cross-entropy.  Both are HBM-bound passes over (rows, V) logits.
//
// CTC, three kernels on one stream:
//   1. ctc_lse_gather : one wave per frame (b,t < in_len[b]).  Streams the V logits once
//      (16 B per lane per load), online max/sum-exp per lane, wave-shuffle reduction -> lse[b,t];
//      then gathers the S = 2L+1 lattice inputs lp[b,t,s] = x[l'_s] - lse (row is L1/L2-hot).
//   2. ctc_alpha_beta : one workgroup of two waves per utterance.  Wave 0 runs the alpha
//      recursion forward in time, wave 1 the beta recursion backward, state s on lane s%64
//      (register j = s/64), predecessors via DPP wave shifts (no LDS in the recursion), lattice
//      inputs prefetched 2 x 8 timesteps ahead in registers so the serial chain never waits on
//      memory.  Linear domain, rescaled by the column maximum every 2 steps (logs of the scales
//      summed for the loss): 5 VALU per step instead of 3 exp + 1 log.
//   3. ctc_grad       : one workgroup per frame.  Wave 0 forms the frame's state posteriors
//      softmax_s(alpha+beta-lp) (exact: the sum over s is p(l|x) at every t) and scatters them into an LDS
//      table indexed by label (ds_add_f32), then one streaming pass: read logits, write
//      dlogits = scale * (softmax - posterior).  Padded frames are written as zeros.
// Algorithmic HBM bytes: logits read twice + dlogits written once = 3 * B*T*V*e;
// the lattice (B*T*S*4 B * 3 arrays) stays in L2 / Infinity Cache.
#include "asr_common.h"

namespace {

constexpr float NEG_INF = -INFINITY;
typedef __attribute__((ext_vector_type(2))) double f64x2;

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

// ---- bf16 fast path: a wave holds one whole row in registers (NV 16-byte vectors per lane)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int NV>
__device__ __forceinline__ void wave_row_load(const bf16_t* __restrict__ x, int nvec, int lane, u32x4 (&xv)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        // out-of-range vectors repeat the row's last vector: harmless for the max, masked in the sum
        xv[k] = *(const u32x4*)(x + (size_t)min(i, nvec - 1) * 8);
    }
}
// log-sum-exp of the row in xv: all loads in flight at once, ONE maximum, then one fma + v_exp_f32
// per element (the online form costs an extra rescale exp per vector and a precise expf per element)
template <int NV>
__device__ __forceinline__ float wave_row_lse_regs(const u32x4 (&xv)[NV], int nvec, int lane, float* row_max = nullptr) {
    float m = NEG_INF;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(bf16_lo(xv[k][j]), bf16_hi(xv[k][j])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (row_max) *row_max = m;
    const float mb = m * LOG2E;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            a += __builtin_amdgcn_exp2f(fmaf(bf16_lo(xv[k][j]), LOG2E, -mb)) + __builtin_amdgcn_exp2f(fmaf(bf16_hi(xv[k][j]), LOG2E, -mb));
        s += (lane + 64 * k < nvec) ? a : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return m + __logf(s);
}

// Index of the FIRST element of the row that equals its maximum m (torch.argmax's tie rule): every lane scans its vectors from the
// last to the first (a lane's indices ascend with k and inside a vector, so the last overwrite is its smallest), then a wave minimum.
// Out-of-range vectors repeat the row's last vector (wave_row_load): excluded.
template <int NV>
__device__ __forceinline__ int wave_row_argmax_regs(const u32x4 (&xv)[NV], int nvec, int lane, float m) {
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = NV - 1; k >= 0; --k) {
        const int i = lane + 64 * k;
        const int base = i < nvec ? i * 8 : 0x7ffffff0;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            bi = bf16_hi(xv[k][j]) == m ? base + 2 * j + 1 : bi;
            bi = bf16_lo(xv[k][j]) == m ? base + 2 * j : bi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bi = min(bi, __shfl_xor(bi, o, 64));
    return bi;
}

// ---------------------------------------------------------------------------------- lattice layout
// The 2L+1 states of the blank-augmented label sequence are kept as two arrays per frame:
//   blanks  Bk[i] = state 2i     (i = 0..L)        labels  Lb[i] = state 2i+1  (i = 0..L-1)
// stored as one row of 2W doubles, INTERLEAVED: entry i = (Bk[i], Lb[i]) at doubles 2i, 2i+1 (round 5: one 16-byte load / store per
// lane and frame instead of two 8-byte ones - half the memory instructions the recursion has in flight, so that its prefetch can run
// three chunks ahead inside the 6-bit vmcnt counter), W = 32/64/128/256 >= L+1, zero beyond the valid entries.  In this form the recursions need ONE lane shift per step instead of two:
//   alpha: Bk'[i] = yB (Bk[i] + Lb[i-1])          Lb'[i] = yL[i] (Lb[i] + Bk[i]   + c[i]  Lb[i-1])
//   beta : Bk'[i] = yB (Bk[i] + Lb[i])            Lb'[i] = yL[i] (Lb[i] + Bk[i+1] + c'[i] Lb[i+1])
// with c[i] = lab[i] != lab[i-1], c'[i] = lab[i] != lab[i+1] (the skip transitions).

// ---------------------------------------------------------------------------------- kernel 1
template <typename T>
__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const T* __restrict__ logits, const int32_t* __restrict__ in_len,
                                                             const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                             double* __restrict__ lp, float* __restrict__ lse_out, int B, int T_, int V, int ld, int Lmax,
                                                             int W, int blank) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) continue;
        const T* x = logits + (size_t)row * ld;
        const float lse = wave_row_lse<T>(x, V, lane, nullptr);
        if (lane == 0) lse_out[row] = lse;
        const int L = lab_len[b];
        double* out = lp + (size_t)row * 2 * W;
        const double yb = exp((double)(to_f32<T>(x[blank]) - lse));   // y_t(blank): linear domain, fp64
        for (int i = lane; i < W; i += 64) {   // zero beyond the valid entries: the recursion runs unpredicated
            const int c = labels[(size_t)b * Lmax + min(i, Lmax - 1)];
            const f64x2 e = {i <= L ? yb : 0.0, i < L ? exp((double)(to_f32<T>(x[c]) - lse)) : 0.0};
            *(f64x2*)(out + 2 * i) = e;
        }
    }
}

// WRITE: the wave that holds the row also writes scale * softmax(row) to dlogits (zeros for padded frames) - the
// bulk of the CTC gradient, which does not depend on the lattice; ctc_label_fix_kernel later corrects the
// (at most L + 1) classes that occur in the label sequence.  One pass over the logits less than with a
// separate gradient kernel (270 instead of 406 MB at config 2).  dlogits may alias logits (no __restrict__:
// the gathers below must stay in front of the stores).
// PATH: the wave also writes the frame's best class (first index of the row maximum; `blank` for padded frames) to best_path - the
// greedy CTC path of the training step's CER, taken from the row while it is in registers (the gradient overwrites the logits in place).
template <int NV, bool WRITE, bool PATH>
__global__ __launch_bounds__(256) void ctc_lse_gather_rows_kernel(const bf16_t* logits, const int32_t* __restrict__ in_len,
                                                                  const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                                  double* __restrict__ lp, float* __restrict__ lse_out, int B, int T_, int V, int ld, int Lmax,
                                                                  int W, int blank, bf16_t* dlogits, float scale_in, const float* __restrict__ scale_div,
                                                                  int32_t* __restrict__ best_path) {
    const float scale = scale_div ? scale_in / *scale_div : scale_in;     // scale_div: a device scalar (global batch under data parallelism)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_, nvec = V >> 3;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) {
            if (WRITE) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                for (int i = lane; i < nvec; i += 64) *(u32x4*)(dlogits + (size_t)row * ld + (size_t)i * 8) = z;
            }
            if (PATH && lane == 0) best_path[row] = blank;
            continue;
        }
        const bf16_t* x = logits + (size_t)row * ld;
        u32x4 xv[NV];
        wave_row_load<NV>(x, nvec, lane, xv);
        const int L = lab_len[b];
        // the gathers are issued before the reduction so that their latency overlaps it
        float xl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            xl[j] = (i < W) ? (float)x[labels[(size_t)b * Lmax + min(i, Lmax - 1)]] : 0.f;
        }
        const float xb = (float)x[blank];
        float row_max;
        const float lse = wave_row_lse_regs<NV>(xv, nvec, lane, &row_max);
        if (lane == 0) lse_out[row] = lse;
        if (PATH) {
            const int bi = wave_row_argmax_regs<NV>(xv, nvec, lane, row_max);
            if (lane == 0) best_path[row] = bi;
        }
        if (WRITE) {
            bf16_t* dl = dlogits + (size_t)row * ld;
            const float lb = lse * LOG2E;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = lane + 64 * k;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] = scale * __builtin_amdgcn_exp2f(fmaf(bf16_lo(xv[k][j]), LOG2E, -lb));
                    v[2 * j + 1] = scale * __builtin_amdgcn_exp2f(fmaf(bf16_hi(xv[k][j]), LOG2E, -lb));
                }
                if (i < nvec) store8<bf16_t>(dl + (size_t)i * 8, v);
            }
        }
        double* out = lp + (size_t)row * 2 * W;
        const double yb = exp((double)(xb - lse));   // y_t(blank): linear domain, fp64
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            if (i < W) {   // zero beyond the valid entries: the recursion runs unpredicated
                const f64x2 e = {i <= L ? yb : 0.0, i < L ? exp((double)(xl[j] - lse)) : 0.0};
                *(f64x2*)(out + 2 * i) = e;
            }
        }
    }
}

// ---------------------------------------------------------------------------------- kernel 2
// The recursion runs in the LINEAR domain with rescaling (the classic scaled forward-backward) in
// fp64 (alpha and beta of one frame can sit > 1e38 apart at the state where their PRODUCT peaks,
// which fp32 cannot hold once a column is scaled to its maximum): one step is a lane shift, an
// add, an fma and a multiply instead of 3 exp + 1 log, and the rescaling is an exact power of two
// folded into a LATER frame's y so that the wave-wide max reduction is off the dependent chain.
// The kernel is a pure dependent chain on B workgroups: instructions per step are what it costs.
// log p(l|x) = ln2 * (sum of exponents taken out) + log of the final column; per-frame scale
// factors cancel in the posterior (it is normalised over the states).
// Whole-wave shifts by one lane through DPP (no LDS round trip as ds_bpermute would need): the
// lane that has no source gets 0.
__device__ __forceinline__ double wave_shr1(double v) {   // lane i <- lane i-1 (two 32-bit DPP moves)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double v) {   // lane i <- lane i+1
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// max over the 64 lanes of unsigned values, DPP only (prefix max inside each 16-lane row, then
// row broadcasts); the result is uniform.  Used on the HIGH words of non-negative doubles, whose
// bit patterns order like the values.
// LAST = highest active lane (63, or 31 when only half the wave runs).
template <int LAST>
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define DPP_MAX(ctrl, rmask)                                                                                   \
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DPP_MAX(0x111, 0xf);   // row_shr:1
    DPP_MAX(0x112, 0xf);   // row_shr:2
    DPP_MAX(0x114, 0xf);   // row_shr:4
    DPP_MAX(0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds the row maximum
    DPP_MAX(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    if (LAST == 63) DPP_MAX(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, LAST);
}

// W = row half-width (compile time: 32, 64, 128, 256), NS = registers per lane per array (entry
// i = lane + 64 j).  Every load and store of the steady-state loop is unconditional, at an
// immediate offset from one running pointer, and the loop body is one basic block, so the
// compiler counts outstanding memory operations (s_waitcnt vmcnt(N), N > 0) and the prefetched
// chunk really overlaps the dependent chain.  (With predicated loads it fell back to vmcnt(0)
// per chunk: ~1.5 us of exposed latency 60 times per utterance.)
// Returns ln of the scale taken out (true value = stored value * exp(ret)), or -inf once every
// state has become 0 (infeasible alignment).
template <int W, bool BWD>
__device__ __forceinline__ double ctc_recursion(const double* __restrict__ y, double* __restrict__ out, const int32_t* __restrict__ lab,
                                                int Tb, int L, int lane) {
    constexpr int NS = W <= 64 ? 1 : W / 64;
    constexpr ptrdiff_t RW = 2 * W;             // row stride
    constexpr ptrdiff_t DT = BWD ? -RW : RW;    // the frame of recursion step k is t = k (alpha) or Tb-1-k (beta)
    int esum = 0;          // sum of the binary exponents scaled away so far
    bool dead = false;     // every state reached 0
    double skip[NS];       // 1.0 where the label-to-label skip transition into (alpha) / out of (beta) Lb[i] exists
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int i = lane + 64 * j;
        bool c;
        if (!BWD) c = i >= 1 && i < L && lab[i] != lab[i - 1];
        else c = i + 1 < L && lab[i] != lab[i + 1];
        skip[j] = c ? 1.0 : 0.0;
    }
    // entry i = lane + 64 j of a row sits at doubles 2 i (blank), 2 i + 1 (label): one 16-byte access per lane and register
    const double* yp = y + (BWD ? (size_t)(Tb - 1) * RW : 0) + 2 * lane;   // step 0
    double* op = out + (BWD ? (size_t)(Tb - 1) * RW : 0) + 2 * lane;
    double bk[NS], lb[NS];
    // step 0: the two entry states (alpha: Bk[0], Lb[0]; beta: Bk[L], Lb[L-1])
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int i = lane + 64 * j;
        const f64x2 y0 = *(const f64x2*)(yp + 128 * j);
        bk[j] = (BWD ? i == L : i == 0) ? y0[0] : 0.0;
        lb[j] = (BWD ? i == L - 1 : i == 0) ? y0[1] : 0.0;
        const f64x2 o0 = {bk[j], lb[j]};
        *(f64x2*)(op + 128 * j) = o0;
    }
    constexpr int CH = 8;  // steps per prefetch chunk; chunk c covers steps 1 + c*CH .. c*CH + CH
    // Prefetch depth.  The recursion is a dependent chain of ~25 ns per frame; a lattice row comes from memory (written by the kernel before
    // this one) in ~0.7 us.  With two chunk buffers (round 1 - 4) the chunk after the current one was requested one chunk = 8 frames earlier
    // and every chunk ended in a wait for memory: 88 ns per frame.  NB buffers = NB - 1 chunks in flight; memory instructions per chunk =
    // CH loads + CH stores (16-byte accesses), so three chunks in flight stay inside the 6-bit vmcnt counter (stores count too on gfx950).
    constexpr int NB = NS == 1 ? 4 : 2;
    const int nsteps = Tb - 1;            // recursion steps 1 .. Tb-1
    struct Chunk { f64x2 v[CH][NS]; };
    Chunk q[NB];
    auto fetch = [&](Chunk& d, const double* p) {   // p = row of the chunk's first step
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int j = 0; j < NS; ++j) d.v[k][j] = *(const f64x2*)(p + k * DT + 128 * j);
    };
    auto fetch_clamped = [&](Chunk& d, int chunk) {   // rows past the last frame are clamped to it (never used)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int step = min(1 + chunk * CH + k, nsteps);
#pragma unroll
            for (int j = 0; j < NS; ++j) d.v[k][j] = *(const f64x2*)(yp + (ptrdiff_t)step * DT + 128 * j);
        }
    };
    double sc = 1.0;   // power-of-two scale waiting to be applied
    int e_pending = 0;
    auto one_step = [&](const f64x2 (&yv)[NS], double* o, int k) {
        double yb[NS], yl[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) { yb[j] = yv[j][0]; yl[j] = yv[j][1]; }
        if ((k & 3) == 1) {   // apply the scale found two steps ago: off the chain, it multiplies y
#pragma unroll
            for (int j = 0; j < NS; ++j) { yb[j] *= sc; yl[j] *= sc; }
            esum += e_pending;
        }
        double nb[NS], nl[NS];
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                double sh = wave_shr1(lb[j]);          // Lb[i-1]
                if (j > 0) sh = lane == 0 ? lane_bcast(lb[j > 0 ? j - 1 : 0], 63) : sh;
                nb[j] = yb[j] * (bk[j] + sh);
                nl[j] = yl[j] * fma(skip[j], sh, lb[j] + bk[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                double sb = wave_shl1(bk[j]);          // Bk[i+1]
                double sl = wave_shl1(lb[j]);          // Lb[i+1]
                if (j + 1 < NS) {
                    sb = lane == 63 ? lane_bcast(bk[j + 1 < NS ? j + 1 : j], 0) : sb;
                    sl = lane == 63 ? lane_bcast(lb[j + 1 < NS ? j + 1 : j], 0) : sl;
                }
                nb[j] = yb[j] * (bk[j] + lb[j]);
                nl[j] = yl[j] * fma(skip[j], sl, lb[j] + sb);
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) { bk[j] = nb[j]; lb[j] = nl[j]; }
        if ((k & 3) == 3) {
            // Every 4th step: binary exponent e of the column maximum (integer DPP max over the
            // doubles' high words); 2^-e is applied two steps later.  No division, no log: the
            // loss only needs the SUM of the exponents.  Six steps shrink the maximum by y^6, far
            // inside the fp64 range for any y a softmax can produce.  Branch-free.
            unsigned hi = 0u;
#pragma unroll
            for (int j = 0; j < NS; ++j) hi = max(hi, max((unsigned)__double2hiint(bk[j]), (unsigned)__double2hiint(lb[j])));
            hi = wave_max_u32<(W < 64 ? W : 64) - 1>(hi);
            const bool zero = (hi >> 20) == 0u;            // all states 0 (or denormal: treated as dead)
            e_pending = zero ? 0 : (int)(hi >> 20) - 1023;
            sc = __hiloint2double((1023 - e_pending) << 20, 0);
            dead |= zero;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const f64x2 ov = {bk[j], lb[j]};
            *(f64x2*)(o + 128 * j) = ov;
        }
    };
    const int nfull = nsteps / CH;
    int c = 0;
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) fetch_clamped(q[i], i);      // q[i] = chunk c + i for i < NB - 1
    const double* yq = yp + DT;   // row of step 1 + c*CH
    double* oq = op + DT;
    // steady state: the chunks requested in an iteration (c + NB - 1 .. c + 2 NB - 2) are full, every access unconditional
    for (; c + 2 * NB - 1 <= nfull; c += NB) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            fetch(q[(i + NB - 1) % NB], yq + (i + NB - 1) * CH * DT);
#pragma unroll
            for (int k = 0; k < CH; ++k) one_step(q[i].v[k], oq + (i * CH + k) * DT, k);
        }
        yq += NB * CH * DT;
        oq += NB * CH * DT;
    }
    // the last chunks (fewer than 2 NB, the last one possibly partial); q[0 .. NB-2] hold chunks c .. c + NB - 2
    for (; c * CH < nsteps; ++c) {
        fetch_clamped(q[NB - 1], c + NB - 1);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (1 + c * CH + k > nsteps) break;
            one_step(q[0].v[k], oq + k * DT, k);
        }
        oq += CH * DT;
#pragma unroll
        for (int i = 0; i + 1 < NB; ++i) q[i] = q[i + 1];
    }
    return dead ? -INFINITY : (double)esum * 0.6931471805599453;
}

template <int W>
__global__ __launch_bounds__(128) void ctc_alpha_beta_kernel(const double* __restrict__ lp, double* __restrict__ alpha, double* __restrict__ beta,
                                                             const int32_t* __restrict__ in_len, const int32_t* __restrict__ labels,
                                                             const int32_t* __restrict__ lab_len, float* __restrict__ nll_out, float* __restrict__ nll_raw,
                                                             int T_, int Lmax, int blank, int zero_infinity) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: each wave runs only its own direction
    const int Tb = in_len[b], L = lab_len[b];
    constexpr size_t RW = 2 * (size_t)W;
    const double* lpb = lp + (size_t)b * T_ * RW;
    double* ab = alpha + (size_t)b * T_ * RW;
    double* bb = beta + (size_t)b * T_ * RW;
    const int32_t* lab = labels + (size_t)b * Lmax;
    __shared__ double s_logc;
    if (Tb > 0 && (W >= 64 || lane < W)) {   // W = 32: half a wave
        if (w == 0) {
            const double logc = ctc_recursion<W, false>(lpb, ab, lab, Tb, L, lane);
            if (lane == 0) s_logc = logc;
        } else {
            ctc_recursion<W, true>(lpb, bb, lab, Tb, L, lane);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        float nll;
        if (Tb <= 0) {
            nll = (L == 0) ? 0.f : INFINITY;
        } else {
            const double* last = ab + (size_t)(Tb - 1) * RW;
            const double tail = last[2 * L] + (L > 0 ? last[2 * (L - 1) + 1] : 0.0);      // interleaved rows: (blank, label) pairs
            nll = (tail > 0.0 && s_logc != -INFINITY) ? (float)(-(s_logc + log(tail))) : INFINITY;
        }
        nll_raw[b] = nll;
        nll_out[b] = (nll == INFINITY && zero_infinity) ? 0.f : nll;
    }
}

// ---------------------------------------------------------------------------------- kernel 3
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ logits, T* __restrict__ dlogits, const double* __restrict__ lp,
                                                       const double* __restrict__ alpha, const double* __restrict__ beta,
                                                       const float* __restrict__ lse_in, const int32_t* __restrict__ in_len,
                                                       const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                       const float* __restrict__ nll_raw, int B, int T_, int V, int ld, int Lmax, int W, int blank,
                                                       float scale_in, int det, const float* __restrict__ scale_div) {
    const float scale = scale_div ? scale_in / *scale_div : scale_in;
    extern __shared__ __attribute__((aligned(16))) float occ[];  // V floats: posterior mass per label
    constexpr int N = Vec<T>::N;
    const int rows = B * T_;
    for (int i = threadIdx.x; i < V; i += 256) occ[i] = 0.f;
    __syncthreads();
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / T_, t = row - b * T_;
        T* dl = dlogits + (size_t)row * ld;
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
        const T* x = logits + (size_t)row * ld;
        const int L = lab_len[b];
        const float lse = lse_in[row];
        if (threadIdx.x < 64) {
            // posterior of a state at this frame, up to a per-frame constant: alpha * beta / y
            // (both recursions include y_t).  sum_s alpha_t(s) beta_t(s) / y_t(l'_s) = p(l|x) for
            // EVERY t, so normalising over s is exact and needs neither nll nor the scale factors.
            const size_t o = (size_t)row * 2 * W;
            double sum_b = 0.0, sum_l = 0.0;
            for (int i = threadIdx.x; i <= L; i += 64) {
                const double yb = lp[o + 2 * i], yl = lp[o + 2 * i + 1];
                sum_b += yb > 0.0 ? alpha[o + 2 * i] * beta[o + 2 * i] / yb : 0.0;
                sum_l += (i < L && yl > 0.0) ? alpha[o + 2 * i + 1] * beta[o + 2 * i + 1] / yl : 0.0;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sum_b += __shfl_xor(sum_b, off, 64);
                sum_l += __shfl_xor(sum_l, off, 64);
            }
            const double sum = sum_b + sum_l;
            if (sum > 0.0) {
                if (threadIdx.x == 0) occ[blank] = (float)(sum_b / sum);      // all blank states share one class
                if (det) {   // repeated labels share a class: add their states in label order (fixed order)
                    if (threadIdx.x == 0)
                        for (int i = 0; i < L; ++i) {
                            const double yl = lp[o + 2 * i + 1];
                            if (yl > 0.0) occ[labels[(size_t)b * Lmax + i]] += (float)(alpha[o + 2 * i + 1] * beta[o + 2 * i + 1] / yl / sum);
                        }
                } else
                for (int i = threadIdx.x; i < L; i += 64) {
                    const double yl = lp[o + 2 * i + 1];
                    if (yl > 0.0) atomicAdd(&occ[labels[(size_t)b * Lmax + i]], (float)(alpha[o + 2 * i + 1] * beta[o + 2 * i + 1] / yl / sum));
                }
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
        if (threadIdx.x == 0) occ[blank] = 0.f;
        for (int i = threadIdx.x; i < L; i += 256) occ[labels[(size_t)b * Lmax + i]] = 0.f;
        __syncthreads();
    }
}

// Second half of the split gradient: the row already holds scale * softmax (ctc_lse_gather_rows_kernel<.., true>);
// one wave per frame subtracts the posterior mass of the classes that occur in the label sequence by
// overwriting those few elements, and zeroes the rows of infeasible utterances (nll = +inf).
__global__ __launch_bounds__(256) void ctc_label_fix_kernel(bf16_t* __restrict__ dlogits, const double* __restrict__ lp, const double* __restrict__ alpha,
                                                            const double* __restrict__ beta, const int32_t* __restrict__ in_len,
                                                            const int32_t* __restrict__ labels, const int32_t* __restrict__ lab_len,
                                                            const float* __restrict__ nll_raw, int B, int T_, int V, int ld, int Lmax, int W, int blank, float scale_in, const float* __restrict__ scale_div) {
    const float scale = scale_div ? scale_in / *scale_div : scale_in;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = B * T_, nvec = V >> 3;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const int b = row / T_, t = row - b * T_;
        if (t >= in_len[b]) continue;           // already zero
        bf16_t* dl = dlogits + (size_t)row * ld;
        if (nll_raw[b] == INFINITY) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = lane; i < nvec; i += 64) *(u32x4*)(dl + (size_t)i * 8) = z;
            continue;
        }
        const int L = lab_len[b];
        // posterior of a state at this frame, up to a per-frame constant: alpha * beta / y; normalising over the
        // states is exact (see ctc_grad_kernel: the sum over s of alpha beta / y is p(l | x) at EVERY t, so it needs neither nll nor the scale factors)
        const size_t o = (size_t)row * 2 * W;
        double pb = 0.0, pl[4], yl[4];
        int lab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            pl[j] = 0.0;
            yl[j] = 0.0;
            lab[j] = -1;
            if (i <= L && i < W) {
                // entry i = (blank, label) pair: one 16-byte load per array
                const f64x2 yv = *(const f64x2*)(lp + o + 2 * i), av = *(const f64x2*)(alpha + o + 2 * i), bv = *(const f64x2*)(beta + o + 2 * i);
                pb += yv[0] > 0.0 ? av[0] * bv[0] / yv[0] : 0.0;
                if (i < L) {
                    yl[j] = yv[1];
                    pl[j] = yl[j] > 0.0 ? av[1] * bv[1] / yl[j] : 0.0;
                    lab[j] = labels[(size_t)b * Lmax + i];
                }
            }
        }
        double sum = pb + pl[0] + pl[1] + pl[2] + pl[3];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum += __shfl_xor(sum, off, 64);
            pb += __shfl_xor(pb, off, 64);
        }
        const double inv = sum > 0.0 ? 1.0 / sum : 0.0;
        float occ[4] = {0.f, 0.f, 0.f, 0.f};   // occupancy of each label's class = sum over the positions that carry the same label
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (64 * jj >= L) break;
            const int cnt = min(64, L - 64 * jj);
            const float pj_all = (float)(pl[jj] * inv);
            for (int q = 0; q < cnt; ++q) {
                const int lq = __shfl(lab[jj], q, 64);
                const float pq = __shfl(pj_all, q, 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) occ[j] += (lab[j] == lq) ? pq : 0.f;
            }
        }
        if (lane == 0) dl[blank] = (bf16_t)(scale * ((float)lp[o] - (float)(pb * inv)));   // every blank state has the same y
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lab[j] >= 0) dl[lab[j]] = (bf16_t)(scale * ((float)yl[j] - occ[j]));       // positions with equal labels write equal values
    }
}

// ---------------------------------------------------------------------------------- xent
// (value, index) of the row maximum over the workgroup, first index on ties (torch.argmax): every thread brings the first maximum of its own
// elements (ascending indices, strict >); `redi` = 8 ints of LDS
__device__ __forceinline__ int block_argmax(float v, int i, float* red, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(v, o, 64);
        const int i2 = __shfl_xor(i, o, 64);
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { red[w] = v; redi[w] = i; }
    __syncthreads();
    float bv = red[0];
    int bi = redi[0];
    for (int k = 1; k < nw; ++k)
        if (red[k] > bv || (red[k] == bv && redi[k] < bi)) { bv = red[k]; bi = redi[k]; }
    return bi;
}

// ARGMAX: also write the row's greedy class (first index of the maximum) to argmax_out[row] - for EVERY row, ignored ones included: the
// reference's per-step CER takes pred.topk(1) of all rows (transformer_official.py:87-91), and the gradient overwrites the logits in place.
template <typename T, bool ARGMAX>
__global__ __launch_bounds__(256) void xent_kernel(const T* __restrict__ logits, const int32_t* __restrict__ gold, const float* __restrict__ n_valid,
                                                   float* __restrict__ row_nll, T* __restrict__ dlogits, int M, int V, int ignore_index,
                                                   float smoothing, float grad_scale, int32_t* __restrict__ argmax_out) {
    __shared__ float red[16];
    __shared__ int redi[8];
    constexpr int N = Vec<T>::N;
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * V;
    const int g = gold[row];
    const bool ignored = (g == ignore_index) || g < 0 || g >= V;
    T* dl = dlogits ? dlogits + (size_t)row * V : nullptr;
    float best = NEG_INF;
    int bi = 0x7fffffff;
    if (ignored) {
        if (ARGMAX) {
            for (int i = threadIdx.x; i < V; i += 256) {
                const float v = to_f32<T>(x[i]);
                if (v > best) { best = v; bi = i; }
            }
            const int a = block_argmax(best, bi, red, redi);      // contains barriers: every thread of the row's workgroup is here
            if (threadIdx.x == 0) argmax_out[row] = a;
        }
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
            if (ARGMAX) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (v[j] > best) { best = v[j]; bi = i * N + j; }
            }
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
            if (ARGMAX && v > best) { best = v; bi = i; }
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
            sx += v;
        }
    }
    if (ARGMAX) {
        const int a = block_argmax(best, bi, red, redi);
        if (threadIdx.x == 0) argmax_out[row] = a;
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

// lattice row stride: whole waves (see ctc_recursion)
// half-width W of a lattice row ([blanks | labels], 2W doubles): L+1 blanks must fit
static inline int width_of(int Lmax) { return Lmax < 32 ? 32 : Lmax < 64 ? 64 : Lmax < 128 ? 128 : 256; }
static inline int smax_of(int Lmax) { return 2 * width_of(Lmax); }

}  // namespace

extern "C" size_t asr_ctc_workspace_bytes(int B, int T, int Lmax) {
    return (size_t)3 * B * T * smax_of(Lmax) * sizeof(double) + ((size_t)B * T + (size_t)B) * sizeof(float);
}

extern "C" int asr_ctc_fwd_bwd(const void* logits, void* dlogits, const int32_t* in_len, const int32_t* labels, const int32_t* lab_len,
                               float* nll, int B, int T, int V, int ld, int Lmax, int blank, float grad_scale, const float* grad_scale_div,
                               int zero_infinity, int32_t* best_path, void* ws, size_t ws_bytes, int dtype, void* stream) {
    if (!logits || !in_len || !labels || !lab_len || !nll || !ws) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: null pointer");
    if (B <= 0 || T <= 0 || V <= 1 || Lmax <= 0 || blank < 0 || blank >= V) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: bad shape B=%d T=%d V=%d Lmax=%d blank=%d", B, T, V, Lmax, blank);
    if (ld < V || (ld != V && ld % 8)) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: row stride ld=%d (V=%d): V, or a multiple of 8 above it", ld, V);
    if (Lmax > 255) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: Lmax = %d: label sequences longer than 255 are not supported", Lmax);
    if (ws_bytes < asr_ctc_workspace_bytes(B, T, Lmax)) ASR_FAIL(ASR_EWORKSPACE, "asr_ctc_fwd_bwd: workspace %zu < %zu", ws_bytes, asr_ctc_workspace_bytes(B, T, Lmax));
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_ctc_fwd_bwd: dtype %d", dtype);
    const size_t lds = (size_t)((V + 3) & ~3) * sizeof(float);
    if (dlogits && lds > 160 * 1024 && !(dtype == ASR_BF16 && V % 8 == 0 && V <= 8192)) ASR_FAIL(ASR_EINVAL, "asr_ctc_fwd_bwd: V=%d does not fit the LDS posterior table", V);
    hipStream_t st = (hipStream_t)stream;
    const int W = width_of(Lmax), Smax = 2 * W;
    double* lp = (double*)ws;
    double* alpha = lp + (size_t)B * T * Smax;
    double* beta = alpha + (size_t)B * T * Smax;
    float* lse = (float*)(beta + (size_t)B * T * Smax);
    float* nll_raw = lse + (size_t)B * T;
    const int rows = B * T;
    int g1 = ceil_div(rows, 4);
    if (g1 > 4096) g1 = 4096;
    // tuning option "cu_limit" (> 0 while the CTC branch of the joint model runs beside the decoder's chain of small kernels): the row kernels are
    // grid-stride loops - size them for that many CUs (8 four-wave workgroups each) instead of flooding every CU with 4096 short workgroups
    const int cu_lim = asr_option(ASR_OPT_CU_LIMIT);
    if (cu_lim > 0 && g1 > 8 * cu_lim) g1 = 8 * cu_lim;
    // bf16 rows of whole 16-byte vectors that fit a wave's registers take the row-in-registers kernels
    const int need = ceil_div(V / 8, 64);
    const bool rows_path = dtype == ASR_BF16 && V % 8 == 0 && ld % 8 == 0 && need <= 16 && ((uintptr_t)logits % 16) == 0 && (!dlogits || ((uintptr_t)dlogits % 16) == 0);
#define ROWS_DISPATCH(CALL)           \
    do {                              \
        if (need <= 2) { CALL(2); }   \
        else if (need <= 4) { CALL(4); }   \
        else if (need <= 6) { CALL(6); }   \
        else if (need <= 9) { CALL(9); }   \
        else if (need <= 12) { CALL(12); } \
        else { CALL(16); }            \
    } while (0)
    if (best_path && !rows_path) {      // fp32 / odd shapes: the decoding kernel, in front of the gradient kernel that may overwrite the logits in place
        const int rc = asr_ctc_frame_argmax(logits, in_len, best_path, B, T, V, ld, blank, dtype, stream);
        if (rc != ASR_OK) return rc;
    }
    if (rows_path) {
        // with a gradient: the softmax part of it is written by the same wave that reduces the row (one pass over the logits less than
        // with a separate row-in-registers gradient kernel: 110 vs 127 us; that kernel, ctc_grad_rows_kernel, is in the git history)
        if (dlogits && best_path) {
#define K1(NV) ctc_lse_gather_rows_kernel<NV, true, true><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank, (bf16_t*)dlogits, grad_scale, grad_scale_div, best_path)
            ROWS_DISPATCH(K1);
#undef K1
        } else if (dlogits) {
#define K1(NV) ctc_lse_gather_rows_kernel<NV, true, false><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank, (bf16_t*)dlogits, grad_scale, grad_scale_div, nullptr)
            ROWS_DISPATCH(K1);
#undef K1
        } else if (best_path) {
#define K1(NV) ctc_lse_gather_rows_kernel<NV, false, true><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank, nullptr, 0.f, nullptr, best_path)
            ROWS_DISPATCH(K1);
#undef K1
        } else {
#define K1(NV) ctc_lse_gather_rows_kernel<NV, false, false><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank, nullptr, 0.f, nullptr, nullptr)
            ROWS_DISPATCH(K1);
#undef K1
        }
    } else if (dtype == ASR_F32) ctc_lse_gather_kernel<float><<<g1, 256, 0, st>>>((const float*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank);
    else ctc_lse_gather_kernel<bf16_t><<<g1, 256, 0, st>>>((const bf16_t*)logits, in_len, labels, lab_len, lp, lse, B, T, V, ld, Lmax, W, blank);
#define AB(N) ctc_alpha_beta_kernel<N><<<B, 128, 0, st>>>(lp, alpha, beta, in_len, labels, lab_len, nll, nll_raw, T, Lmax, blank, zero_infinity)
    if (W == 32) AB(32);
    else if (W == 64) AB(64);
    else if (W == 128) AB(128);
    else AB(256);
#undef AB
    if (dlogits) {
        int g3 = rows < 2048 ? rows : 2048;
        if (rows_path) {
            asr_launch_armed(ctc_label_fix_kernel, dim3(g1), dim3(256), 0, st, (bf16_t*)dlogits, lp, alpha, beta, in_len, labels, lab_len, nll_raw, B, T, V, ld, Lmax, W, blank, grad_scale, grad_scale_div);      // last kernel: may carry an armed completion event
        } else if (dtype == ASR_F32) ctc_grad_kernel<float><<<g3, 256, lds, st>>>((const float*)logits, (float*)dlogits, lp, alpha, beta, lse, in_len, labels, lab_len, nll_raw, B, T, V, ld, Lmax, W, blank, grad_scale, asr_deterministic(), grad_scale_div);
        else ctc_grad_kernel<bf16_t><<<g3, 256, lds, st>>>((const bf16_t*)logits, (bf16_t*)dlogits, lp, alpha, beta, lse, in_len, labels, lab_len, nll_raw, B, T, V, ld, Lmax, W, blank, grad_scale, asr_deterministic(), grad_scale_div);
    }
    ASR_CHECK_LAUNCH("asr_ctc_fwd_bwd");
    return ASR_OK;
}

extern "C" int asr_xent_fwd_bwd(const void* logits, const int32_t* gold, const float* n_valid, float* row_nll, void* dlogits, int M,
                                int V, int ignore_index, float smoothing, float grad_scale, int32_t* argmax_out, int dtype, void* stream) {
    if (!logits || !gold || !row_nll || (dlogits && !n_valid)) ASR_FAIL(ASR_EINVAL, "asr_xent_fwd_bwd: null pointer");
    if (M <= 0 || V <= 1) ASR_FAIL(ASR_EINVAL, "asr_xent_fwd_bwd: bad shape M=%d V=%d", M, V);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_xent_fwd_bwd: dtype %d", dtype);
#define XENT(T_, A_) xent_kernel<T_, A_><<<M, 256, 0, st>>>((const T_*)logits, gold, n_valid, row_nll, (T_*)dlogits, M, V, ignore_index, smoothing, grad_scale, argmax_out)
    if (dtype == ASR_F32) { if (argmax_out) XENT(float, true); else XENT(float, false); }
    else { if (argmax_out) XENT(bf16_t, true); else XENT(bf16_t, false); }
#undef XENT
    ASR_CHECK_LAUNCH("asr_xent_fwd_bwd");
    return ASR_OK;
}
