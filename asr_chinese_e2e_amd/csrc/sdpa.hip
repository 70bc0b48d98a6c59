// Masked scaled-dot-product attention, flash style, forward and backward.
//
// bf16 path (MFMA 32x32x16, head dim 64), all three kernels share one tiling idea: the tensor the
// kernel OWNS lives in registers as MFMA B-operand fragments, the tensor it STREAMS goes through
// 64-row LDS tiles, and the first product is oriented so that the softmax axis bookkeeping is
// lane-local and its accumulator is directly the operand of the second product
// (guide section 3, "An accumulator tile as the next MFMA's operand"):
//
//   fwd  (workgroup = 128 queries, 4 waves x 32):   S^T = K Q^T  -> P^T (keys in regs, query on lane)
//                                                   O^T += V^T P^T   (V^T via ds_read_b64_tr_b16)
//   dQ   (workgroup = 128 queries):                 S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta)
//                                                   dQ^T += K^T dS^T (K^T via transposed LDS read)
//   dKV  (workgroup = 128 keys):                    S = Q K^T, dP = dO V^T  (queries in regs, key on lane)
//                                                   dV^T += dO^T P ; dK^T += Q^T dS (transposed LDS reads)
//
// Scores never touch memory (the reference materialises (H*B, T, T) fp32 = 256 MB per layer,
// attention.py:76-84).  Masks come from k_len / causal / window, never from a mask tensor.
// Algorithmic HBM bytes per (b,h): fwd = Q,K,V read + O written = 4*T*dk*2 B.
//
// f32 path: exact-fp32 VALU kernels with the same decomposition (used for fp32 parity mode and
// for head sizes other than 64); one wave per query (fwd, dQ) or per key (dKV).
#include "asr_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float M_INIT = -1.0e30f;  // finite "minus infinity" for running maxima

__device__ __forceinline__ bool visible(int qi, int kj, int klen, int causal, int window) {
    bool ok = kj < klen;
    if (causal) ok = ok && (kj <= qi);
    if (window >= 0) ok = ok && (kj - qi <= window) && (qi - kj <= window);
    return ok;
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA path
// ------------------------------------------------------------------------------------------
constexpr int DK = 64;        // head dim
constexpr int TS = 72;        // LDS tile row stride in elements (144 B: ds_read_b128 rows conflict-free)
constexpr int TILE = 64;      // rows per LDS tile
constexpr int TILE_ELEMS = TILE * TS;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// A/B operand fragment from a row-major LDS tile: lane (r = l&31, hh = l>>5) takes
// tile[row0 + r][16*ks + 8*hh .. +8)
__device__ __forceinline__ bf16x8 frag_row(const bf16_t* tile, int row0, int ks, int lane) {
    return *(const bf16x8*)(tile + (row0 + (lane & 31)) * TS + 16 * ks + 8 * (lane >> 5));
}
// Transposed operand fragment (hardware transpose read): lane (r, hh) gets, for j = 0..7,
// tile[row0 + 16*s + 8*(j>>2) + 4*hh + (j&3)][col0 + r]  - the k-order an accumulator tile
// converted in place presents (guide section 3).
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* tile, int row0, int col0, int s, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const bf16_t* p = tile + (row0 + 16 * s + 4 * (G >> 1) + (i >> 2)) * TS + col0 + 16 * (G & 1) + 4 * (i & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 8 * TS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// accumulator (rows in regs, col on lane) -> bf16 operand fragment of k-step s (regs 8s..8s+7)
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& x, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
    return f;
}
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// 64 x 64 tile: global rows [row0, row0+64) of a (rows, ld) matrix starting at column col0;
// rows >= row_limit are zero-filled.  Each thread moves 2 x 16 B.
struct TileRegs { u32x4 v[2]; };
__device__ __forceinline__ void tile_load(TileRegs& tr, const bf16_t* __restrict__ base, size_t ld, int row0, int row_limit, int tid) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        u32x4 z = {0u, 0u, 0u, 0u};
        tr.v[c] = (row0 + row < row_limit) ? *(const u32x4*)(base + (size_t)(row0 + row) * ld + ch * 8) : z;
    }
}
__device__ __forceinline__ void tile_store(const TileRegs& tr, bf16_t* tile, int tid) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        *(u32x4*)(tile + row * TS + ch * 8) = tr.v[c];
    }
}
// per-lane operand fragments straight from global: rows row0 + (l&31) (clamped), 4 k-steps
__device__ __forceinline__ void frags_from_global(bf16x8 (&f)[4], const bf16_t* __restrict__ base, size_t ld, int row0, int row_limit, int lane) {
    int row = row0 + (lane & 31);
    const bool ok = row < row_limit;
    row = ok ? row : row_limit - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 v = *(const bf16x8*)(base + (size_t)row * ld + 16 * ks + 8 * (lane >> 5));
        if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
        }
        f[ks] = v;
    }
}
// store a transposed accumulator pair (rows = d in regs, col = row index on lane) as rows of a
// (rows, ld) bf16 matrix: 8-byte pieces of 4 consecutive d
__device__ __forceinline__ void store_rows_T(const f32x16 (&acc)[2], float mul, bf16_t* __restrict__ base, size_t ld, int row0, int row_limit, int lane) {
    const int row = row0 + (lane & 31);
    if (row >= row_limit) return;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[db][4 * g4 + e] * mul);
            *(bf16x4*)(base + (size_t)row * ld + 32 * db + 8 * g4 + 4 * (lane >> 5)) = o;
        }
}

// ---------------------------------------------------------------- forward
template <bool DROP>
__global__ __launch_bounds__(256, 2) void sdpa_fwd_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ k_len, int H,
                                                            int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale,
                                                            uint32_t dseed, uint32_t dthr, float dscale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * TILE_ELEMS];
    bf16_t* Kt = smem;
    bf16_t* Vt = smem + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // 1-D grid, XCD-aware: the query blocks of one (b, h) - which all stream the same K/V - get
    // consecutive virtual ids and so share one XCD's L2 (K/V otherwise re-fetched per query block)
    const int nqb = (Tq + 127) >> 7, vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    const int bh = vid / nqb, b = bh / H, h = bh - b * H, qblk = (vid - bh * nqb) * 128, q0 = qblk + 32 * w;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* kb = k + (size_t)b * Tk * ldk + h * DK;
    const bf16_t* vb = v + (size_t)b * Tk * ldv + h * DK;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    int kend = klen, kbeg = 0;
    if (causal) kend = min(kend, min(qblk + 128, Tq));
    if (window >= 0) { kend = min(kend, min(qblk + 128, Tq) + window); kbeg = max(0, qblk - window) & ~63; }
    bf16x8 qf[4];
    frags_from_global(qf, qb, ldq, q0, Tq, lane);
    const int qi = q0 + (lane & 31);
    const float sc2 = scale * LOG2E;
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float m = M_INIT, l = 0.f;
    TileRegs kr, vr;
    if (kbeg < kend) { tile_load(kr, kb, ldk, kbeg, klen, tid); tile_load(vr, vb, ldv, kbeg, klen, tid); }
    for (int k0 = kbeg; k0 < kend; k0 += TILE) {
        __syncthreads();
        tile_store(kr, Kt, tid);
        tile_store(vr, Vt, tid);
        __syncthreads();
        if (k0 + TILE < kend) { tile_load(kr, kb, ldk, k0 + TILE, klen, tid); tile_load(vr, vb, ldv, k0 + TILE, klen, tid); }
        f32x16 st[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[sub][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Kt, 32 * sub, ks, lane), qf[ks], st[sub], 0, 0, 0);
        }
        // The kernel is VALU-bound (dk = 64: 16 MFMAs vs ~32 softmax elements per lane and tile), so
        // the per-element work is kept to max / fma / exp2 / add: tiles that lie wholly inside every
        // mask skip the visibility test (wave-uniform branch), the 1/sqrt(dk)*log2(e) scale is folded
        // into one fma with the running maximum, and exp2 is the bare v_exp_f32.
        const bool need_mask = (k0 + TILE > klen) || (causal && k0 + TILE - 1 > q0) ||
                               (window >= 0 && (k0 + TILE - 1 - q0 > window || q0 + 31 - k0 > window));
        if (need_mask) {
            if (!causal && window < 0) {   // key-length mask only (encoder self-attention): one compare per element
                const int lim = klen - k0 - 4 * (lane >> 5);      // key (32*sub + (i&3) + 8*(i>>2)) + 4*hh + k0 < klen
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (32 * sub + (i & 3) + 8 * (i >> 2) >= lim) st[sub][i] = -INFINITY;
            } else {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int kj = k0 + 32 * sub + acc_row(i, lane);
                        if (!visible(qi, kj, klen, causal, window)) st[sub][i] = -INFINITY;
                    }
            }
        }
        float tmax = M_INIT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, st[sub][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);                       // running maximum in raw-score units
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * sc2);
        m = mn;
        const float mc = mn * sc2;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[sub][i], sc2, -mc));
                st[sub][i] = p;
                psum += p;
            }
        l = l * alpha + psum;
        if constexpr (DROP) {   // attention.py:83: dropout on the probabilities (the normaliser l stays undropped)
            const uint32_t rowbase = (((uint32_t)(b * H + h)) * Tq + min(qi, Tq - 1)) * ((Tk + 1) & ~1);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const uint32_t hsh = drop_hash((rowbase + k0 + 32 * sub + acc_row(i, lane)) >> 1, dseed);
                    st[sub][i] = drop_keep(hsh, 0, dthr) ? st[sub][i] * dscale : 0.f;
                    st[sub][i + 1] = drop_keep(hsh, 1, dthr) ? st[sub][i + 1] * dscale : 0.f;
                }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = acc_to_frag(st[sub], s);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vt, 32 * sub, 32 * db, s, lane), pf, oacc[db], 0, 0, 0);
            }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    store_rows_T(oacc, inv, o + (size_t)b * Tq * ldo + h * DK, ldo, q0, Tq, lane);
    if (lane < 32 && qi < Tq) lse[((size_t)b * H + h) * Tq + qi] = l > 0.f ? (m * sc2 + log2f(l)) * LN2 : -INFINITY;
}

// delta[b,h,q] = sum_d dO * O   (one wave per 8 rows x 8 lanes... simple: one thread-group of 8 lanes per (row, head))
template <typename T>
__global__ __launch_bounds__(256) void sdpa_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ delta, int B, int H, int Tq,
                                                         int dk, int ldo) {
    // 8 lanes per (b, t, h); each lane strides over dk
    const int gid = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    const int total = B * Tq * H;
    float s = 0.f;
    int b = 0, t = 0, h = 0;
    if (gid < total) {
        h = gid % H;
        const int bt = gid / H;
        b = bt / Tq; t = bt - b * Tq;
        const size_t off = (size_t)bt * ldo + (size_t)h * dk;
        for (int c = sub; c < dk; c += 8) s += to_f32<T>(o[off + c]) * to_f32<T>(d_o[off + c]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (gid < total && sub == 0) delta[((size_t)b * H + h) * Tq + t] = s;
}

// ---------------------------------------------------------------- backward: dQ
template <bool DROP>
__global__ __launch_bounds__(256, 3) void sdpa_bwd_dq_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16_t* __restrict__ dq, const int32_t* __restrict__ k_len, int H, int Tq, int Tk, int ldq,
                                                               int ldk, int ldv, int ldo, int causal, int window, float scale, uint32_t dseed, uint32_t dthr,
                                                               float dscale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * TILE_ELEMS];
    bf16_t* Kt = smem;
    bf16_t* Vt = smem + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nqb = (Tq + 127) >> 7, vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    const int bh = vid / nqb, b = bh / H, h = bh - b * H, qblk = (vid - bh * nqb) * 128, q0 = qblk + 32 * w;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* dob = d_o + (size_t)b * Tq * ldo + h * DK;
    const bf16_t* kb = k + (size_t)b * Tk * ldk + h * DK;
    const bf16_t* vb = v + (size_t)b * Tk * ldv + h * DK;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    int kend = klen, kbeg = 0;
    if (causal) kend = min(kend, min(qblk + 128, Tq));
    if (window >= 0) { kend = min(kend, min(qblk + 128, Tq) + window); kbeg = max(0, qblk - window) & ~63; }
    bf16x8 qf[4], dof[4];
    frags_from_global(qf, qb, ldq, q0, Tq, lane);
    frags_from_global(dof, dob, ldo, q0, Tq, lane);
    const int qi = q0 + (lane & 31);
    const size_t stat = ((size_t)b * H + h) * Tq + min(qi, Tq - 1);
    // a query with no admissible key at all (a padded frame beyond the long-form band) has lse = -inf: its probabilities are 0
    const float lse_raw = lse[stat];
    const float lse2 = lse_raw == -INFINITY ? 1.0e30f : lse_raw * LOG2E;
    // delta = rowsum(dO o O): this wave already holds its 32 dO rows as fragments, so the separate
    // delta pass (one more read of O and dO, one more launch) is folded in here; the result is also
    // written out for the dK/dV kernel that runs next on the stream.
    float dl = 0.f;
    {
        bf16x8 of[4];
        frags_from_global(of, o + (size_t)b * Tq * ldo + h * DK, ldo, q0, Tq, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[ks][j];
        dl += __shfl_xor(dl, 32, 64);
        if (lane < 32 && qi < Tq) delta[stat] = dl;
    }
    const float sc2 = scale * LOG2E;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    TileRegs kr, vr;
    if (kbeg < kend) { tile_load(kr, kb, ldk, kbeg, klen, tid); tile_load(vr, vb, ldv, kbeg, klen, tid); }
    for (int k0 = kbeg; k0 < kend; k0 += TILE) {
        __syncthreads();
        tile_store(kr, Kt, tid);
        tile_store(vr, Vt, tid);
        __syncthreads();
        if (k0 + TILE < kend) { tile_load(kr, kb, ldk, k0 + TILE, klen, tid); tile_load(vr, vb, ldv, k0 + TILE, klen, tid); }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Kt, 32 * sub, ks, lane), qf[ks], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Vt, 32 * sub, ks, lane), dof[ks], dp, 0, 0, 0);
            }
            const int ks0 = k0 + 32 * sub;
            const bool need_mask = (ks0 + 32 > klen) || (causal && ks0 + 31 > q0) || (window >= 0 && (ks0 + 31 - q0 > window || q0 + 31 - ks0 > window));
            if (need_mask) {
                if (!causal && window < 0) {   // key-length mask only
                    const int lim = klen - ks0 - 4 * (lane >> 5);
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((i & 3) + 8 * (i >> 2) >= lim) st[i] = -INFINITY;
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (!visible(qi, ks0 + acc_row(i, lane), klen, causal, window)) st[i] = -INFINITY;
                }
            }
            if constexpr (DROP) {   // dP = (dO V^T) o keep / (1-p)
                const uint32_t rowbase = (((uint32_t)(b * H + h)) * Tq + min(qi, Tq - 1)) * ((Tk + 1) & ~1);
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const uint32_t hsh = drop_hash((rowbase + ks0 + acc_row(i, lane)) >> 1, dseed);
                    dp[i] = drop_keep(hsh, 0, dthr) ? dp[i] * dscale : 0.f;
                    dp[i + 1] = drop_keep(hsh, 1, dthr) ? dp[i + 1] * dscale : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[i], sc2, -lse2));
                st[i] = p * (dp[i] - dl);  // dS^T / scale: the factor is applied once, when dQ is stored
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 df = acc_to_frag(st, s);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Kt, 32 * sub, 32 * db, s, lane), df, acc[db], 0, 0, 0);
            }
        }
    }
    store_rows_T(acc, scale, dq + (size_t)b * Tq * ldq + h * DK, ldq, q0, Tq, lane);
}

// ---------------------------------------------------------------- backward: dK, dV
// launch bound 2 waves/SIMD: without it the kernel takes 176 arch + 96 accumulator registers = 272
// of the unified 512-entry file, i.e. ONE workgroup per CU (measured: 134 us, 2.8 waves/CU average)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void sdpa_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dk_, bf16_t* __restrict__ dv, const int32_t* __restrict__ k_len, int H,
                                                                int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale,
                                                                uint32_t dseed, uint32_t dthr, float dscale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * TILE_ELEMS];
    __shared__ __attribute__((aligned(16))) float stats[2 * TILE];
    bf16_t* Qt = smem;
    bf16_t* Dt = smem + TILE_ELEMS;
    float* s_lse = stats;
    float* s_del = stats + TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nkb = (Tk + 127) >> 7, vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    const int bh = vid / nkb, b = bh / H, h = bh - b * H, kblk = (vid - bh * nkb) * 128, kk0 = kblk + 32 * w;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* dob = d_o + (size_t)b * Tq * ldo + h * DK;
    const bf16_t* kb = k + (size_t)b * Tk * ldk + h * DK;
    const bf16_t* vb = v + (size_t)b * Tk * ldv + h * DK;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    bf16x8 kf[4], vf[4];
    frags_from_global(kf, kb, ldk, kk0, Tk, lane);
    frags_from_global(vf, vb, ldv, kk0, Tk, lane);
    const int kj = kk0 + (lane & 31);
    const float sc2 = scale * LOG2E;
    f32x16 dka[2], dva[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dka[0][i] = dka[1][i] = dva[0][i] = dva[1][i] = 0.f; }
    // query range that can see this key block
    int qbeg = 0, qend = Tq;
    if (causal) qbeg = kblk & ~63;
    if (window >= 0) { qbeg = max(qbeg, (kblk - window) & ~63); qbeg = max(qbeg, 0); qend = min(Tq, kblk + 128 + window); }
    if (kblk >= klen) qend = qbeg;  // whole key block is padding: gradients are zero
    const float* lseb = lse + ((size_t)b * H + h) * Tq;
    const float* delb = delta + ((size_t)b * H + h) * Tq;
    TileRegs qr, dr;
    float st_l = 0.f, st_d = 0.f;
    auto stat_load = [&](int q0) {
        if (tid < TILE) {
            const int qi = q0 + tid;
            const float lr = qi < Tq ? lseb[qi] : -INFINITY;
            st_l = lr == -INFINITY ? 1.0e30f : lr * LOG2E;  // exp2(s - 1e30) = 0 for rows past the end and for queries that see no key (lse = -inf)
            st_d = qi < Tq ? delb[qi] : 0.f;
        }
    };
    if (qbeg < qend) { tile_load(qr, qb, ldq, qbeg, Tq, tid); tile_load(dr, dob, ldo, qbeg, Tq, tid); stat_load(qbeg); }
    for (int q0 = qbeg; q0 < qend; q0 += TILE) {
        __syncthreads();
        tile_store(qr, Qt, tid);
        tile_store(dr, Dt, tid);
        if (tid < TILE) { s_lse[tid] = st_l; s_del[tid] = st_d; }
        __syncthreads();
        if (q0 + TILE < qend) { tile_load(qr, qb, ldq, q0 + TILE, Tq, tid); tile_load(dr, dob, ldo, q0 + TILE, Tq, tid); stat_load(q0 + TILE); }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Qt, 32 * sub, ks, lane), kf[ks], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Dt, 32 * sub, ks, lane), vf[ks], dp, 0, 0, 0);
            }
            f32x16 ds;
            const int qs0 = q0 + 32 * sub;   // wave-uniform: does this 32x32 sub-tile touch any mask edge?
            const bool need_mask = (kk0 + 32 > klen) || (causal && kk0 + 31 > qs0) || (window >= 0 && (kk0 + 31 - qs0 > window || qs0 + 31 - kk0 > window));
            if (need_mask) {
                if (!causal && window < 0) {   // key-length mask only: the key is this lane's, one test per tile
                    if (kj >= klen) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) st[i] = -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (!visible(q0 + 32 * sub + acc_row(i, lane), kj, klen, causal, window)) st[i] = -INFINITY;
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int r0 = 32 * sub + 8 * g4 + 4 * (lane >> 5);
                const f32x4 l4 = *(const f32x4*)(s_lse + r0);
                const f32x4 d4 = *(const f32x4*)(s_del + r0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    const float sv = st[i];
                    const float p = __builtin_amdgcn_exp2f(fmaf(sv, sc2, -l4[e]));
                    float keepf = 1.f;
                    if constexpr (DROP) {
                        const uint32_t el = (((uint32_t)(b * H + h)) * Tq + min(q0 + r0 + e, Tq - 1)) * ((Tk + 1) & ~1) + min(kj, Tk - 1);
                        keepf = drop_keep_at(el, dseed, dthr) ? dscale : 0.f;
                    }
                    st[i] = p * keepf;                               // dropped probabilities feed dV
                    ds[i] = p * (dp[i] * keepf - d4[e]);             // dS / scale: applied once, when dK is stored
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = acc_to_frag(st, s);
                const bf16x8 df = acc_to_frag(ds, s);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dva[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Dt, 32 * sub, 32 * db, s, lane), pf, dva[db], 0, 0, 0);
                    dka[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qt, 32 * sub, 32 * db, s, lane), df, dka[db], 0, 0, 0);
                }
            }
        }
    }
    store_rows_T(dka, scale, dk_ + (size_t)b * Tk * ldk + h * DK, ldk, kk0, Tk, lane);
    store_rows_T(dva, 1.f, dv + (size_t)b * Tk * ldv + h * DK, ldv, kk0, Tk, lane);
}

// ------------------------------------------------------------------------------------------
// exact fp32 VALU path (any dk <= 128)
// ------------------------------------------------------------------------------------------
// one wave per (b, h, query): scores -> LDS, softmax, then lanes own output columns
template <typename T>
__global__ __launch_bounds__(64) void sdpa_fwd_generic_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ o,
                                                              float* __restrict__ lse, const int32_t* __restrict__ k_len, int H, int Tq, int Tk, int dk,
                                                              int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale, uint32_t dseed,
                                                              uint32_t dthr, float dscale) {
    extern __shared__ float sc[];  // Tk scores, then dk floats of the query row
    float* qrow = sc + Tk;
    const int lane = threadIdx.x;
    const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    const T* qp = q + ((size_t)b * Tq + qi) * ldq + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) qrow[c] = to_f32<T>(qp[c]);
    __syncthreads();
    float m = -INFINITY;
    for (int j = lane; j < Tk; j += 64) {
        float s = -INFINITY;
        if (visible(qi, j, klen, causal, window)) {
            const T* kp = k + ((size_t)b * Tk + j) * ldk + (size_t)h * dk;
            float a = 0.f;
            for (int c = 0; c < dk; ++c) a += qrow[c] * to_f32<T>(kp[c]);
            s = a * scale;
        }
        sc[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Tk; j += 64) {
        const float p = (m == -INFINITY) ? 0.f : expf(sc[j] - m);
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    __syncthreads();
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (dthr) {   // dropout on the probabilities; l (the normaliser) is already summed
        const uint32_t rowbase = (((uint32_t)(b * H + h)) * Tq + qi) * ((Tk + 1) & ~1);
        for (int j = lane; j < Tk; j += 64) sc[j] = drop_keep_at(rowbase + j, dseed, dthr) ? sc[j] * dscale : 0.f;
        __syncthreads();
    }
    T* op = o + ((size_t)b * Tq + qi) * ldo + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) {
        float a = 0.f;
        for (int j = 0; j < Tk; ++j) {
            const float p = sc[j];
            if (p != 0.f) a += p * to_f32<T>(v[((size_t)b * Tk + j) * ldv + (size_t)h * dk + c]);
        }
        op[c] = from_f32<T>(a * inv);
    }
    if (lane == 0) lse[((size_t)b * H + h) * Tq + qi] = l > 0.f ? m + logf(l) : -INFINITY;
}

// dQ: one wave per (b,h,query)
template <typename T>
__global__ __launch_bounds__(64) void sdpa_bwd_dq_generic_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                 const T* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 T* __restrict__ dq, const int32_t* __restrict__ k_len, int H, int Tq, int Tk, int dk, int ldq,
                                                                 int ldk, int ldv, int ldo, int causal, int window, float scale, uint32_t dseed, uint32_t dthr,
                                                                 float dscale) {
    extern __shared__ float sc[];  // Tk dS values, then q row (dk) and dO row (dk)
    float* qrow = sc + Tk;
    float* dorow = qrow + dk;
    const int lane = threadIdx.x;
    const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    const size_t st = ((size_t)b * H + h) * Tq + qi;
    const float L = lse[st], dl = delta[st];
    const T* qp = q + ((size_t)b * Tq + qi) * ldq + (size_t)h * dk;
    const T* dop = d_o + ((size_t)b * Tq + qi) * ldo + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) { qrow[c] = to_f32<T>(qp[c]); dorow[c] = to_f32<T>(dop[c]); }
    __syncthreads();
    for (int j = lane; j < Tk; j += 64) {
        float ds = 0.f;
        if (visible(qi, j, klen, causal, window)) {
            const T* kp = k + ((size_t)b * Tk + j) * ldk + (size_t)h * dk;
            const T* vp = v + ((size_t)b * Tk + j) * ldv + (size_t)h * dk;
            float a = 0.f, dp = 0.f;
            for (int c = 0; c < dk; ++c) { a += qrow[c] * to_f32<T>(kp[c]); dp += dorow[c] * to_f32<T>(vp[c]); }
            if (dthr) dp = drop_keep_at((((uint32_t)(b * H + h)) * Tq + qi) * ((Tk + 1) & ~1) + j, dseed, dthr) ? dp * dscale : 0.f;
            ds = expf(a * scale - L) * (dp - dl) * scale;
        }
        sc[j] = ds;
    }
    __syncthreads();
    T* dqp = dq + ((size_t)b * Tq + qi) * ldq + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) {
        float a = 0.f;
        for (int j = 0; j < Tk; ++j) {
            const float ds = sc[j];
            if (ds != 0.f) a += ds * to_f32<T>(k[((size_t)b * Tk + j) * ldk + (size_t)h * dk + c]);
        }
        dqp[c] = from_f32<T>(a);
    }
}

// dK, dV: one wave per (b,h,key)
template <typename T>
__global__ __launch_bounds__(64) void sdpa_bwd_dkv_generic_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                  const T* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  T* __restrict__ dk_, T* __restrict__ dv, const int32_t* __restrict__ k_len, int H, int Tq, int Tk,
                                                                  int dk, int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale, uint32_t dseed,
                                                                  uint32_t dthr, float dscale) {
    extern __shared__ float sc[];  // Tq p values, Tq dS values, k row (dk), v row (dk)
    float* ps = sc;
    float* dss = sc + Tq;
    float* krow = dss + Tq;
    float* vrow = krow + dk;
    const int lane = threadIdx.x;
    const int kj = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    const T* kp = k + ((size_t)b * Tk + kj) * ldk + (size_t)h * dk;
    const T* vp = v + ((size_t)b * Tk + kj) * ldv + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) { krow[c] = to_f32<T>(kp[c]); vrow[c] = to_f32<T>(vp[c]); }
    __syncthreads();
    for (int i = lane; i < Tq; i += 64) {
        float p = 0.f, ds = 0.f;
        if (visible(i, kj, klen, causal, window)) {
            const T* qp = q + ((size_t)b * Tq + i) * ldq + (size_t)h * dk;
            const T* dop = d_o + ((size_t)b * Tq + i) * ldo + (size_t)h * dk;
            float a = 0.f, dp = 0.f;
            for (int c = 0; c < dk; ++c) { a += to_f32<T>(qp[c]) * krow[c]; dp += to_f32<T>(dop[c]) * vrow[c]; }
            const size_t st = ((size_t)b * H + h) * Tq + i;
            p = expf(a * scale - lse[st]);
            float keepf = 1.f;
            if (dthr) keepf = drop_keep_at((((uint32_t)(b * H + h)) * Tq + i) * ((Tk + 1) & ~1) + kj, dseed, dthr) ? dscale : 0.f;
            ds = p * (dp * keepf - delta[st]) * scale;
            p *= keepf;
        }
        ps[i] = p;
        dss[i] = ds;
    }
    __syncthreads();
    T* dkp = dk_ + ((size_t)b * Tk + kj) * ldk + (size_t)h * dk;
    T* dvp = dv + ((size_t)b * Tk + kj) * ldv + (size_t)h * dk;
    for (int c = lane; c < dk; c += 64) {
        float a = 0.f, e = 0.f;
        for (int i = 0; i < Tq; ++i) {
            const float p = ps[i], ds = dss[i];
            if (p != 0.f || ds != 0.f) {
                a += ds * to_f32<T>(q[((size_t)b * Tq + i) * ldq + (size_t)h * dk + c]);
                e += p * to_f32<T>(d_o[((size_t)b * Tq + i) * ldo + (size_t)h * dk + c]);
            }
        }
        dkp[c] = from_f32<T>(a);
        dvp[c] = from_f32<T>(e);
    }
}

static bool mfma_ok(int dk, int ldq, int ldk, int ldv, int ldo, const void* a, const void* b, const void* c, const void* d) {
    return dk == DK && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 &&
           (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) % 16) == 0;
}

static int check_common(const char* name, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk, int ldv, int ldo) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || dk <= 0 || dk > 128) ASR_FAIL(ASR_EINVAL, "%s: bad shape B=%d H=%d Tq=%d Tk=%d dk=%d", name, B, H, Tq, Tk, dk);
    if (ldq < H * dk || ldk < H * dk || ldv < H * dk || ldo < H * dk) ASR_FAIL(ASR_EINVAL, "%s: row stride smaller than H*dk", name);
    if (B > 65535 || H > 65535) ASR_FAIL(ASR_EINVAL, "%s: B or H exceeds grid limits", name);
    return ASR_OK;
}

}  // namespace

extern "C" int asr_sdpa_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* k_len, int B, int H, int Tq, int Tk, int dk,
                            int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale, float drop_p, uint32_t dseed, int dtype,
                            void* stream) {
    if (!q || !k || !v || !o || !lse) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: null pointer");
    if (int rc = check_common("asr_sdpa_fwd", B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo)) return rc;
    if (drop_p < 0.f || drop_p >= 1.f) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: bad dropout p=%f", drop_p);
    if ((double)B * H * Tq * (Tk + 1) >= 4294967296.0) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: B*H*Tq*Tk exceeds the 32-bit dropout counter");
    const uint32_t dthr = drop_thr16(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_BF16 && mfma_ok(dk, ldq, ldk, ldv, ldo, q, k, v, o)) {
        const int grid = ceil_div(Tq, 128) * H * B;
        if (dthr) sdpa_fwd_bf16_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        else sdpa_fwd_bf16_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
    } else {
        dim3 grid(Tq, H, B);
        const size_t lds = (size_t)(Tk + dk) * sizeof(float);
        if (lds > 64 * 1024) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: generic path needs Tk+dk <= 16384");
        if (dtype == ASR_F32) sdpa_fwd_generic_kernel<float><<<grid, 64, lds, st>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, lse, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        else if (dtype == ASR_BF16) sdpa_fwd_generic_kernel<bf16_t><<<grid, 64, lds, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        else ASR_FAIL(ASR_EDTYPE, "asr_sdpa_fwd: dtype %d", dtype);
    }
    ASR_CHECK_LAUNCH("asr_sdpa_fwd");
    return ASR_OK;
}

extern "C" int asr_sdpa_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta, void* dq, void* dk_,
                            void* dv, const int32_t* k_len, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk, int ldv, int ldo, int causal, int window,
                            float scale, float drop_p, uint32_t dseed, int dtype, void* stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk_ || !dv) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: null pointer");
    if (int rc = check_common("asr_sdpa_bwd", B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo)) return rc;
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_sdpa_bwd: dtype %d", dtype);
    if (drop_p < 0.f || drop_p >= 1.f) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: bad dropout p=%f", drop_p);
    const uint32_t dthr = drop_thr16(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = B * Tq * H;
    const bool mfma = dtype == ASR_BF16 && mfma_ok(dk, ldq, ldk, ldv, ldo, q, k, v, d_o) && mfma_ok(dk, ldq, ldk, ldv, ldo, dq, dk_, dv, o);
    if (!mfma) {   // the MFMA dQ kernel computes delta itself
        if (dtype == ASR_F32) sdpa_delta_kernel<float><<<ceil_div(ngroups, 32), 256, 0, st>>>((const float*)o, (const float*)d_o, delta, B, H, Tq, dk, ldo);
        else sdpa_delta_kernel<bf16_t><<<ceil_div(ngroups, 32), 256, 0, st>>>((const bf16_t*)o, (const bf16_t*)d_o, delta, B, H, Tq, dk, ldo);
    }
    if (mfma) {
        const int gq = ceil_div(Tq, 128) * H * B, gk = ceil_div(Tk, 128) * H * B;
#define SDPA_BWD(D)                                                                                                                                        \
    do {                                                                                                                                                   \
        sdpa_bwd_dq_bf16_kernel<D><<<gq, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, (const bf16_t*)o, lse, delta, (bf16_t*)dq, \
                                                        k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);                  \
        sdpa_bwd_dkv_bf16_kernel<D><<<gk, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk_, \
                                                         (bf16_t*)dv, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);    \
    } while (0)
        if (dthr) SDPA_BWD(true);
        else SDPA_BWD(false);
#undef SDPA_BWD
    } else {
        dim3 gq(Tq, H, B), gk(Tk, H, B);
        const size_t l1 = (size_t)(Tk + 2 * dk) * sizeof(float), l2 = (size_t)(2 * Tq + 2 * dk) * sizeof(float);
        if (l1 > 64 * 1024 || l2 > 64 * 1024) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: generic path sequence too long for LDS");
        if (dtype == ASR_F32) {
            sdpa_bwd_dq_generic_kernel<float><<<gq, 64, l1, st>>>((const float*)q, (const float*)k, (const float*)v, (const float*)d_o, lse, delta, (float*)dq, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
            sdpa_bwd_dkv_generic_kernel<float><<<gk, 64, l2, st>>>((const float*)q, (const float*)k, (const float*)v, (const float*)d_o, lse, delta, (float*)dk_, (float*)dv, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        } else {
            sdpa_bwd_dq_generic_kernel<bf16_t><<<gq, 64, l1, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
            sdpa_bwd_dkv_generic_kernel<bf16_t><<<gk, 64, l2, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk_, (bf16_t*)dv, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        }
    }
    ASR_CHECK_LAUNCH("asr_sdpa_bwd");
    return ASR_OK;
}
