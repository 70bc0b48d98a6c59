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
#include <stdlib.h>

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
// store a transposed accumulator pair (rows = d in regs, col = row index on lane) as rows of a (rows, ld) bf16 matrix.
// A lane holds 4 consecutive d per register group (8 g4 + 4 (lane >> 5) + e); the two lanes of a row (l, l + 32) trade one 8-byte piece
// through v_permlane32_swap so that each ends up with 8 consecutive d = ONE 16-byte store per pair of groups: the epilogue is
// store-issue bound (a wave instruction of 64 separate 8-byte pieces costs as much as one of 16-byte pieces), so half the
// instructions is half its time.  Same bytes in memory as the 8-byte form.
__device__ __forceinline__ void store_rows_T(const f32x16 (&acc)[2], float mul, bf16_t* __restrict__ base, size_t ld, int row0, int row_limit, int lane) {
    const int row = row0 + (lane & 31);
    const bool ok = row < row_limit;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {      // groups g4 = 2 gp (d 16 gp .. + 7) and 2 gp + 1 (d 16 gp + 8 .. + 15)
            bf16x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p0[e] = (bf16_t)(acc[db][8 * gp + e] * mul);
                p1[e] = (bf16_t)(acc[db][8 * gp + 4 + e] * mul);
            }
            const u32x2 a = __builtin_bit_cast(u32x2, p0), b = __builtin_bit_cast(u32x2, p1);
            // swap(x, y): the upper 32 lanes of x trade places with the lower 32 lanes of y
            const auto s0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
            // lanes < 32: own d 0..3 | partner's d 4..7;  lanes >= 32: partner's d 8..11 | own d 12..15   (relative to 32 db + 16 gp)
            const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) *(u32x4*)(base + (size_t)row * ld + 32 * db + 16 * gp + 8 * (lane >> 5)) = v;
        }
}

// The low-order part of the same rows: bf16(x - float(bf16(x))) of x = acc * mul, the other 8 bits of mantissa the bf16 O drops.  Only the
// backward pass reads it, for delta = rowsum(dO o O): with O rounded to bf16 that sum carries 2^-9 |O| of error per coordinate, and where the
// rows of K and of V have a component in common (a bias behind a LayerNorm is enough) dQ = sum_j p_j (dP_j - delta) K_j turns an error eps of
// delta into eps * (the mean key) while the true dQ only sees the keys' DEVIATIONS from that mean: 1 - cos(dQ) 5e-2 (mean = 3 sigma) or 7e-4
// (1 sigma) with O alone against 2e-4 / 2e-5 with the two pieces, the floor set by rounding Q, K, V, dO being 3e-5 / 7e-6
// (tools/sdpa_delta_forms.py; at full size the encoder's top-layer Q / K projection gradients measured 0.989 against the oracle without it).
__device__ __forceinline__ void store_rows_T_lo(const f32x16 (&acc)[2], float mul, bf16_t* __restrict__ base, size_t ld, int row0, int row_limit, int lane) {
    const int row = row0 + (lane & 31);
    const bool ok = row < row_limit;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            bf16x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = acc[db][8 * gp + e] * mul, x1 = acc[db][8 * gp + 4 + e] * mul;
                p0[e] = (bf16_t)(x0 - (float)(bf16_t)x0);
                p1[e] = (bf16_t)(x1 - (float)(bf16_t)x1);
            }
            const u32x2 a = __builtin_bit_cast(u32x2, p0), b = __builtin_bit_cast(u32x2, p1);
            const auto s0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
            const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) *(u32x4*)(base + (size_t)row * ld + 32 * db + 16 * gp + 8 * (lane >> 5)) = v;
        }
}

// ---------------------------------------------------------------- forward
template <bool DROP>
__global__ __launch_bounds__(256, 2) void sdpa_fwd_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ k_len, int H,
                                                            int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale,
                                                            uint32_t dseed, uint32_t dthr, float dscale, bf16_t* __restrict__ o_lo) {
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
    if (o_lo) store_rows_T_lo(oacc, inv, o_lo + (size_t)b * Tq * ldo + h * DK, ldo, q0, Tq, lane);
    if (lane < 32 && qi < Tq) lse[((size_t)b * H + h) * Tq + qi] = l > 0.f ? (m * sc2 + log2f(l)) * LN2 : -INFINITY;
}

// delta[b,h,q] = sum_d dO * O   (one wave per 8 rows x 8 lanes... simple: one thread-group of 8 lanes per (row, head))
template <typename T>
__global__ __launch_bounds__(256) void sdpa_delta_kernel(const T* __restrict__ o, const T* __restrict__ o_lo, const T* __restrict__ d_o, float* __restrict__ delta, int B, int H, int Tq,
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
        for (int c = sub; c < dk; c += 8) s += (to_f32<T>(o[off + c]) + (o_lo ? to_f32<T>(o_lo[off + c]) : 0.f)) * to_f32<T>(d_o[off + c]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (gid < total && sub == 0) delta[((size_t)b * H + h) * Tq + t] = s;
}

// ---------------------------------------------------------------- backward: dQ
template <bool DROP>
__global__ __launch_bounds__(256, 3) void sdpa_bwd_dq_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ o, const bf16_t* __restrict__ o_lo, const float* __restrict__ lse,
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
        if (o_lo) {      // the low-order piece of O (see store_rows_T_lo)
            frags_from_global(of, o_lo + (size_t)b * Tq * ldo + h * DK, ldo, q0, Tq, lane);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[ks][j];
        }
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

// ---------------------------------------------------------------- forward: K and V resident in LDS (Tk <= 512)
// The kernel above stages every 64-key K / V tile through registers with two barriers per tile and each 128-query workgroup
// streams the whole K / V of its head again.  When the keys of a head fit LDS (Tk <= 512: K and V images of 64 KiB each)
// ONE workgroup of 8 waves serves a whole (b, h) pair:
//   * K and V arrive ONCE, by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 key rows per wave instruction), all issued at
//     kernel entry in tile order; the first pass over the tiles waits per tile (counted vmcnt + one barrier), so the
//     softmax of tile 0 runs while tiles 1..7 are still in flight (a register-staged prologue cost ~5 us: the 32 MB of
//     all heads' K / V at the memory rate, with every CU idle);
//   * rows are 128 B unpadded; 16-byte chunk c of row r sits at position c ^ f((r >> 1) & 7), f(h) = ((h & 1) << 2) | (h >> 1)
//     (applied to the SOURCE address of the DMA): the ds_read_b128 row fragments of K and the ds_read_b64_tr_b16
//     transposed fragments of V are both bank-conflict free;
//   * every wave owns 64 queries (two 32-query blocks, one after the other; Q fragments in registers) and walks the key tiles
//     with no barrier after the first pass; fragment reads are issued a phase ahead of their MFMAs;
//   * same arithmetic as sdpa_fwd_bf16_kernel (S^T = K Q^T so that the softmax statistics are lane-local and P^T is already
//     the operand of O^T += V^T P^T; exp2 domain; masks from lengths), except that the row sums of P come out of the
//     matrix pipe (an all-ones A operand: 4 MFMAs instead of 32 VALU adds per tile - the loop is VALU-issue bound) and the
//     accumulator is only rescaled when the running maximum moved by more than 2^8.
constexpr int FF_KEYS = 512, FF_THREADS = 512;
constexpr int FF_IMG = FF_KEYS * 128;          // 65536 B per image
constexpr int FF_LDS = 2 * FF_IMG;
__device__ __forceinline__ int ff_swz(int row) {
    const int h = (row >> 1) & 7;
    return ((h & 1) << 2) | (h >> 1);
}
__device__ __forceinline__ void ff_wait_tiles(int younger) {      // at most `younger` tiles (2 DMAs each) may still be in flight
    switch (younger) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    }
}

template <bool DROP, bool MASKED>
__global__ __launch_bounds__(FF_THREADS, 2) void sdpa_fwd_fused_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                         bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ k_len, int H,
                                                                         int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale,
                                                                         uint32_t dseed, uint32_t dthr, float dscale) {
    extern __shared__ __attribute__((aligned(1024))) char smem_ff[];
    const char* Kimg = smem_ff;
    const char* Vimg = smem_ff + FF_IMG;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* kb = k + (size_t)b * Tk * ldk + h * DK;
    const bf16_t* vb = v + (size_t)b * Tk * ldv + h * DK;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    const int ntile = (klen + TILE - 1) / TILE;      // 64-key tiles that hold a valid key
    const float sc2 = scale * LOG2E;
    bf16_t* ob = o + (size_t)b * Tq * ldo + h * DK;

    // The K / V stream: tile t = two DMAs (K, V) of every wave; rows past klen repeat the last key (finite; masked).
    // The DMA is inline asm, invisible to the compiler's vmcnt bookkeeping, so the order is: tile 0, then the Q fragments of the
    // wave's first block as ordinary loads that are CONSUMED (the compiler's vmcnt(0) for them covers tile 0, which is needed
    // first anyway), then tiles 1..; from there on only the counted waits of ff_wait_tiles() touch vmcnt.
    bf16x8 qf[4];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_ff;
    const int r8 = 8 * w + (lane >> 3), c8 = 8 * ((lane & 7) ^ ff_swz(r8));
    auto dma_tile = [&](int t) {
        const int rc = min(TILE * t + r8, klen - 1);
        const bf16_t* sk = kb + (size_t)rc * ldk + c8;
        const bf16_t* sv = vb + (size_t)rc * ldv + c8;
        const unsigned dk_ = __builtin_amdgcn_readfirstlane(lds0 + (TILE * t + 8 * w) * 128), dv_ = dk_ + FF_IMG;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sk), "s"(dk_) : "memory");      // m0 has no compiler-generated user in this kernel
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sv), "s"(dv_) : "memory");
    };
    // Only FF_AHEAD tiles are ever in flight: with all of them requested at entry (every CU at once: 48 MB) a wave's FIRST tile
    // queues behind everybody's later ones and the first softmax started ~10 us into the kernel.
    // The rendezvous (counted wait + barrier) is per PAIR of tiles: a barrier per tile locked the two waves of a SIMD into the
    // same phase (all MFMA, then all VALU) and cost ~40 % per tile against the barrier-free second pass.
    constexpr int FF_AHEAD = 3;
    if (ntile > 0) dma_tile(0);
    frags_from_global(qf, qb, ldq, 64 * w, Tq, lane);
    asm volatile("; Q fragments landed" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])::"memory");
    for (int t = 1; t < min(ntile, MASKED ? FF_KEYS / TILE : FF_AHEAD + 1); ++t) dma_tile(t);      // masked variants: everything (they wait for all of it)
    // lane bases of the fragment reads (bytes); tile, sub-block and k-step are immediates / one add
    int kofs[4], vofs[2][2];
    {
        const int r = lane & 31, hh = lane >> 5, f = ff_swz(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kofs[ks] = r * 128 + (((2 * ks + hh) ^ f) << 4);
        const int G = lane >> 4, i = lane & 15, rv = 4 * (G >> 1) + (i >> 2);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int chunk = 4 * db + 2 * (G & 1) + ((i & 3) >> 1);
                vofs[db][hi] = FF_IMG + (rv + 8 * hi) * 128 + ((chunk ^ ff_swz(rv + 8 * hi)) << 4) + 8 * (i & 1);      // V image: the rest fits the 16-bit offset field
            }
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    bool streamed = false;      // the first pass over the tiles consumes the stream; after it everything is resident
    for (int qc = 0; qc < Tq; qc += 64 * 8) {         // 512 queries per pass (one pass at T = 500)
        const int q0w = qc + 64 * w;
        bf16x8 qnext[4];
#pragma unroll 1
        for (int qbk = 0; qbk < 2; ++qbk) {
            const int q0 = q0w + 32 * qbk;
            const bool stream = !streamed && !MASKED;
            if (!streamed && MASKED) {      // masked variants walk different tile ranges per wave: no per-tile rendezvous, wait for everything
                ff_wait_tiles(0);
                __syncthreads();
            }
            if (streamed) {
                if (q0 >= Tq) break;
                if (qbk == 1 && !MASKED) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qf[ks] = qnext[ks];      // fetched under the first block's last tile
                } else {
                    frags_from_global(qf, qb, ldq, q0, Tq, lane);
                }
            }
            const bool active = q0 < Tq;       // an idle wave of the first pass still keeps the rendezvous
            const int qi = q0 + (lane & 31);
            f32x16 oacc[2], lacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; lacc[i] = 0.f; }
            float m = M_INIT, l = 0.f;
            int t0 = 0, t1 = ntile;
            if (MASKED) {
                if (causal) t1 = min(t1, (min(q0 + 32, Tq) + TILE - 1) / TILE);
                if (window >= 0) { t1 = min(t1, (min(q0 + 32, Tq) + window + TILE - 1) / TILE); t0 = max(0, q0 - window) / TILE; }
                if (!active) t1 = t0;
            }
            bf16x8 kf[2][4];
            constexpr bool PREFETCH = !MASKED;      // the masked variants have no registers to spare for it
#define FF_LOAD_K(T_)                                                                                       \
    {                                                                                                       \
        const char* kpt = Kimg + (T_) * (TILE * 128);                                                       \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub)                                                 \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) kf[sub][ks] = *(const bf16x8*)(kpt + kofs[ks] + sub * 4096); \
    }
            if (PREFETCH && !stream && t0 < t1) FF_LOAD_K(t0);
            for (int t = t0; t < t1; ++t) {
                const int k0 = t * TILE;
                if (stream) {
                    if ((t & 1) == 0) {      // tiles t and t+1 must have landed; t+2 and t+3 may be in flight; t+4, t+5 go out
                        ff_wait_tiles(min(2, max(0, ntile - 2 - t)));
                        __syncthreads();
                        if (t + 4 < ntile) dma_tile(t + 4);
                        if (t + 5 < ntile) dma_tile(t + 5);
                    }
                    if (!active) continue;
#ifdef SDPA_FWD_SKIP
                    if (SDPA_FWD_SKIP & 1) continue;      // diagnostic builds: the first pass only streams (WRONG results, timing only)
#endif
                    FF_LOAD_K(t);
                } else if (!PREFETCH) {
                    FF_LOAD_K(t);
                }
#ifdef SDPA_FWD_SKIP
                if ((SDPA_FWD_SKIP & 2) && !stream) continue;      // diagnostic builds: no work in the second (resident) pass
#endif
                if (!MASKED && qbk == 0 && t == t1 - 1 && q0 + 32 < Tq) frags_from_global(qnext, qb, ldq, q0 + 32, Tq, lane);
                f32x16 st[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[sub][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[ks], st[sub], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 vf[2][2][2];
                {
                    const char* vpt = Kimg + k0 * 128;      // vofs carries the V image offset
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int db = 0; db < 2; ++db) {
                                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vpt + vofs[db][0] + (32 * sub + 16 * s2) * 128));
                                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vpt + vofs[db][1] + (32 * sub + 16 * s2) * 128));
                                vf[sub][s2][db] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                            }
                }
                if (PREFETCH && !stream && t + 1 < t1) FF_LOAD_K(t + 1);
                __builtin_amdgcn_sched_barrier(0);
                const bool edge = (k0 + TILE > klen) || (MASKED && ((causal && k0 + TILE - 1 > q0) || (window >= 0 && (k0 + TILE - 1 - q0 > window || q0 + 31 - k0 > window))));
                if (edge) {
                    if (!MASKED) {      // key-length mask only: one compare per element
                        const int lim = klen - k0 - 4 * (lane >> 5);
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                            for (int i = 0; i < 16; ++i)
                                if (32 * sub + (i & 3) + 8 * (i >> 2) >= lim) st[sub][i] = -INFINITY;
                    } else {
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                            for (int i = 0; i < 16; ++i)
                                if (!visible(qi, k0 + 32 * sub + acc_row(i, lane), klen, causal, window)) st[sub][i] = -INFINITY;
                    }
                }
                // The loop is VALU-issue bound (dk = 64: 16 MFMAs against 32 softmax elements per lane), so the element work is
                // max3 / fma / exp2 (scalar f32: the packed v_pk_* forms issue slower than the two ops they replace), and the
                // accumulators are only rescaled when the running maximum moved by more than 2^8 in the exp2 domain (P stays
                // <= 256: exact in fp32, and l / lse stay consistent with m).
                float tmax = M_INIT;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, st[sub][i]);
                {      // the other half-wave's maximum: one VALU lane swap instead of an LDS round trip
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                if (__builtin_amdgcn_ballot_w64((tmax - m) * sc2 > 8.f) != 0) {      // wave-uniform
                    asm volatile("; rescale" ::: "memory");
                    const float mn = fmaxf(m, tmax);
                    const float alpha = __builtin_amdgcn_exp2f((m - mn) * sc2);
                    m = mn;
                    if constexpr (DROP) l *= alpha;
                    else lacc[0] *= alpha;      // every register of lacc holds the same sum; only [0] is read
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
                }
                const float mc = m * sc2;
                float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const float p0 = __builtin_amdgcn_exp2f(fmaf(st[sub][i], sc2, -mc));
                        const float p1 = __builtin_amdgcn_exp2f(fmaf(st[sub][i + 1], sc2, -mc));
                        st[sub][i] = p0;
                        st[sub][i + 1] = p1;
                        if constexpr (DROP) { ps0 += p0; ps1 += p1; }
                    }
                if constexpr (DROP) {
                    l += ps0 + ps1;      // the sum is over the un-dropped P: VALU adds here
                    const uint32_t rowbase = (((uint32_t)(b * H + h)) * Tq + min(qi, Tq - 1)) * ((Tk + 1) & ~1);
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const uint32_t hsh = drop_hash((rowbase + k0 + 32 * sub + acc_row(i, lane)) >> 1, dseed);
                            const bool ka = drop_keep(hsh, 0, dthr), kb = drop_keep(hsh, 1, dthr);
                            st[sub][i] = ka ? st[sub][i] * dscale : 0.f;
                            st[sub][i + 1] = kb ? st[sub][i + 1] * dscale : 0.f;
                        }
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 pf = acc_to_frag(st[sub], s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[sub][s2][db], pf, oacc[db], 0, 0, 0);
                        if constexpr (!DROP) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
                    }
            }
#undef FF_LOAD_K
            streamed = true;
            if (!active) break;
            if constexpr (DROP) l += __shfl_xor(l, 32, 64);
            else l = lacc[0];      // the contraction ran over all 64 keys of every tile: no cross-half add
            const float inv = l > 0.f ? 1.f / l : 0.f;
            store_rows_T(oacc, inv, ob, ldo, q0, Tq, lane);
            if (lane < 32 && qi < Tq) lse[((size_t)b * H + h) * Tq + qi] = l > 0.f ? (m * sc2 + log2f(l)) * LN2 : -INFINITY;
        }
    }
}

// ---------------------------------------------------------------- forward, both 32-query blocks of a wave in ONE pass over the key tiles (round 5)
// sdpa_fwd_fused_bf16_kernel walks the key tiles twice per wave: its first 32-query block consumes the K / V stream (and waits for it), its
// second block re-reads every K and V fragment from LDS.  Here a wave takes a tile's K fragments ONCE for the scores of both blocks and its V
// fragments once for both P V products: half the LDS fragment reads per score element (the loop is vector-issue bound, and an LDS read costs an
// issue slot like any vector instruction), and the stream hides behind twice the arithmetic per tile.  Same image layout, same DMA ring, same
// arithmetic per element (exp2 domain, row sums on the matrix pipe, rescale only when the maximum moved by > 2^8) - the two kernels agree to
// rounding of the order of accumulation only.  Key-length masking only (no causal / band, no dropout: those take the kernel above).
// Registers: Q fragments 32, O accumulators 64, row sums 32, scores 64, V fragments of one 32-key half 16, K fragments 32 (dead before the
// softmax): 247 at the peak, no scratch - K fragments are NOT prefetched a tile ahead here (no room).
// MEASURED (tools/sdpa_pair_ab.py, one process): bit-identical outputs and lse, and SLOWER - 33.5 vs 31.6 us at (32, 8, 500, 500), 30.0 vs 29.4 ragged,
// 12.1 vs 10.9 at the decoder's cross-attention shape, 20.5 vs 20.9 at (8, 8, 333, 470): what the shared fragment reads save is lost to a loop body
// whose 64 exponentials sit between two blocks of 16 - 24 MFMAs as the compiler schedules it (the two-pass kernel's shorter phases interleave
// better across the two waves of a SIMD).  Opt-in (tuning option "sdpa_pair"); a hand-placed schedule of this body is the open lever.
__global__ __launch_bounds__(FF_THREADS, 2) void sdpa_fwd_pair_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                        bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ k_len, int H,
                                                                        int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, float scale) {
    extern __shared__ __attribute__((aligned(1024))) char smem_ff[];
    const char* Kimg = smem_ff;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* kb = k + (size_t)b * Tk * ldk + h * DK;
    const bf16_t* vb = v + (size_t)b * Tk * ldv + h * DK;
    const int klen = min(k_len ? k_len[b] : Tk, Tk);
    const int ntile = (klen + TILE - 1) / TILE;
    const float sc2 = scale * LOG2E;
    bf16_t* ob = o + (size_t)b * Tq * ldo + h * DK;

    bf16x8 qf[2][4];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_ff;
    const int r8 = 8 * w + (lane >> 3), c8 = 8 * ((lane & 7) ^ ff_swz(r8));
    auto dma_tile = [&](int t) {
        const int rc = min(TILE * t + r8, klen - 1);
        const bf16_t* sk = kb + (size_t)rc * ldk + c8;
        const bf16_t* sv = vb + (size_t)rc * ldv + c8;
        const unsigned dk_ = __builtin_amdgcn_readfirstlane(lds0 + (TILE * t + 8 * w) * 128), dv_ = dk_ + FF_IMG;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sk), "s"(dk_) : "memory");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sv), "s"(dv_) : "memory");
    };
    constexpr int FF_AHEAD = 3;
    if (ntile > 0) dma_tile(0);
    frags_from_global(qf[0], qb, ldq, 64 * w, Tq, lane);
    frags_from_global(qf[1], qb, ldq, 64 * w + 32, Tq, lane);
    asm volatile("; Q fragments landed" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3])::"memory");
    for (int t = 1; t < min(ntile, FF_AHEAD + 1); ++t) dma_tile(t);
    int kofs[4], vofs[2][2];
    {
        const int r = lane & 31, hh = lane >> 5, f = ff_swz(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kofs[ks] = r * 128 + (((2 * ks + hh) ^ f) << 4);
        const int G = lane >> 4, i = lane & 15, rv = 4 * (G >> 1) + (i >> 2);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int chunk = 4 * db + 2 * (G & 1) + ((i & 3) >> 1);
                vofs[db][hi] = FF_IMG + (rv + 8 * hi) * 128 + ((chunk ^ ff_swz(rv + 8 * hi)) << 4) + 8 * (i & 1);
            }
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    bool streamed = false;
    for (int qc = 0; qc < Tq; qc += 64 * 8) {
        const int q0 = qc + 64 * w;                     // block 0: q0 .. q0 + 31, block 1: q0 + 32 .. q0 + 63
        const bool stream = !streamed;
        if (streamed) {
            if (q0 >= Tq) break;
            frags_from_global(qf[0], qb, ldq, q0, Tq, lane);
            frags_from_global(qf[1], qb, ldq, q0 + 32, Tq, lane);
        }
        const bool act0 = q0 < Tq, act1 = q0 + 32 < Tq;      // wave-uniform; an idle wave of the first pass still keeps the rendezvous
        f32x16 oacc[2][2], lacc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][0][i] = 0.f; oacc[0][1][i] = 0.f; oacc[1][0][i] = 0.f; oacc[1][1][i] = 0.f; lacc[0][i] = 0.f; lacc[1][i] = 0.f; }
        float m[2] = {M_INIT, M_INIT};
        for (int t = 0; t < ntile; ++t) {
            const int k0 = t * TILE;
            if (stream) {
                if ((t & 1) == 0) {      // tiles t and t+1 must have landed; t+2 and t+3 may be in flight; t+4, t+5 go out
                    ff_wait_tiles(min(2, max(0, ntile - 2 - t)));
                    __syncthreads();
                    if (t + 4 < ntile) dma_tile(t + 4);
                    if (t + 5 < ntile) dma_tile(t + 5);
                }
                if (!act0) continue;
            }
            // ---- scores of both blocks from ONE read of the tile's K fragments
            f32x16 st[2][2];      // [block][32-key half]
            {
                bf16x8 kf[2][4];
                const char* kpt = Kimg + t * (TILE * 128);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) kf[sub][ks] = *(const bf16x8*)(kpt + kofs[ks] + sub * 4096);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { st[0][sub][i] = 0.f; st[1][sub][i] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        st[0][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[0][ks], st[0][sub], 0, 0, 0);
                        if (act1) st[1][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[1][ks], st[1][sub], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool edge = k0 + TILE > klen;
            if (edge) {      // key-length mask: one compare per element
                const int lim = klen - k0 - 4 * (lane >> 5);
#pragma unroll
                for (int bk = 0; bk < 2; ++bk)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (32 * sub + (i & 3) + 8 * (i >> 2) >= lim) st[bk][sub][i] = -INFINITY;
            }
            // ---- softmax of each block (lane = query): running maximum, deferred rescale, exp2
#pragma unroll
            for (int bk = 0; bk < 2; ++bk) {
                if (bk == 1 && !act1) break;
                float tmax = M_INIT;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, st[bk][sub][i]);
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                if (__builtin_amdgcn_ballot_w64((tmax - m[bk]) * sc2 > 8.f) != 0) {      // wave-uniform
                    const float mn = fmaxf(m[bk], tmax);
                    const float alpha = __builtin_amdgcn_exp2f((m[bk] - mn) * sc2);
                    m[bk] = mn;
                    lacc[bk][0] *= alpha;      // every register of lacc holds the same sum; only [0] is read
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[bk][0][i] *= alpha; oacc[bk][1][i] *= alpha; }
                }
                const float mc = m[bk] * sc2;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[bk][sub][i] = __builtin_amdgcn_exp2f(fmaf(st[bk][sub][i], sc2, -mc));
            }
            // ---- O^T += V^T P^T for both blocks from ONE read of the tile's V fragments (a 32-key half at a time)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                bf16x8 vf[2][2];      // [16-key step][d half]
                const char* vpt = Kimg + k0 * 128;      // vofs carries the V image offset
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vpt + vofs[db][0] + (32 * sub + 16 * s2) * 128));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vpt + vofs[db][1] + (32 * sub + 16 * s2) * 128));
                        vf[s2][db] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 p0 = acc_to_frag(st[0][sub], s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db) oacc[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][db], p0, oacc[0][db], 0, 0, 0);
                    lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, p0, lacc[0], 0, 0, 0);
                    if (act1) {
                        const bf16x8 p1 = acc_to_frag(st[1][sub], s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db) oacc[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][db], p1, oacc[1][db], 0, 0, 0);
                        lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, p1, lacc[1], 0, 0, 0);
                    }
                }
            }
        }
        streamed = true;
        if (!act0) break;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
            if (bk == 1 && !act1) break;
            const int qs = q0 + 32 * bk, qi = qs + (lane & 31);
            const float l = lacc[bk][0];
            const float inv = l > 0.f ? 1.f / l : 0.f;
            store_rows_T(oacc[bk], inv, ob, ldo, qs, Tq, lane);
            if (lane < 32 && qi < Tq) lse[((size_t)b * H + h) * Tq + qi] = l > 0.f ? (m[bk] * sc2 + log2f(l)) * LN2 : -INFINITY;
        }
    }
}

// ---------------------------------------------------------------- backward: ONE kernel per (b, h)
// The dQ + dK/dV pair above reads Q, K, V, dO twice and computes S and dP twice (7 products, 2 exp passes; PMC traffic
// 1.52 x the algorithmic bytes).  When all keys of a head fit one workgroup (Tk <= 512: every encoder / decoder shape of
// the T = 500 configurations) the whole backward of a (b, h) pair is ONE workgroup of 8 waves:
//   * wave w owns keys [64 w, 64 w + 64): their dK^T and dV^T live in its accumulators for the whole kernel (no sum
//     across workgroups, no atomics), their V fragments in registers; K of all 512 keys sits in LDS once (72 KiB),
//     pre-multiplied by scale * log2(e) so that the scores leave the MFMA in the exp2 domain;
//   * the queries stream through in 32-row tiles (Q, dO: 4.5 KiB each, double buffered, prefetched more than a tile
//     ahead in registers together with -lse log2(e) and -delta = -rowsum(dO o O), computed here from the O rows);
//   * S and dP are computed ONCE, key on the lane (guide: "Key on the lane"), their accumulators START at the row
//     constants (-lse log2(e), -delta; lanes of padded keys read a row of -1e30 instead), so p = exp2(S') and
//     dS = p dP' need no scale, no subtraction, no row maximum and no key mask; P and dS feed dV^T += dO^T P and
//     dK^T += Q^T dS straight from the accumulators;
//   * only dS crosses LDS, once: each wave writes its 64 keys into a [key][query] image (8-byte pieces, 64-byte rows,
//     chunk c of row r at c ^ ((r >> 1) & 7): writes and transposed reads conflict-free), double buffered, so ONE
//     barrier per tile is enough; after it wave w computes ONE 16 x 16 block of dQ^T = K^T dS^T over all keys with
//     MFMA 16x16x32 (both operands by transposed LDS reads) and stores it - dQ is reduced inside the workgroup, in a
//     fixed order, while other waves already work on the next tile.
// 5 products, 1 exp pass; HBM traffic = Q, K, V, O, dO read once + dQ, dK, dV written once.  Deterministic.
// (A 4-wave form with the whole 512-entry register file per wave - operand sets shared by four key blocks, a third of
// the LDS traffic - measured 102 us against 82 us for this form: one wave per SIMD exposes every LDS and MFMA latency.)
constexpr int FB_WAVES = 8, FB_KEYS = 64 * FB_WAVES, FB_QT = 32, FB_THREADS = 64 * FB_WAVES;
// K image and Q / dO tiles: rows of 128 B, unpadded; the 16-byte chunk c of row r sits at chunk c ^ ff_swz(r) (the forward kernel's layout):
// the ds_read_b128 row fragments AND the transposed ds_read_b64_tr_b16 fragments (32x32x16 and 16x16x32 forms) are bank-conflict
// free.  (Rows padded to 144 B served the row fragments only: the transposed reads of a half-wave - rows r, r + 1, r + 2, r + 3 and
// r + 8 .. - landed two deep on the banks: 31 % of the LDS cycles of round 2's kernel were conflicts.)
constexpr int FBS = 64;                                       // row stride in elements
constexpr int FB_K_BYTES = FB_KEYS * FBS * 2;                 // 65536
constexpr int FB_DS_BYTES = FB_KEYS * FB_QT * 2;              // 32768 per buffer
constexpr int FB_TILE_ELEMS = FB_QT * FBS;
constexpr int FB_STATS = (2 * 64 + 32 + 64) * 4;              // [2 buffers][-lse log2e + q . mean key (32) | -delta (32)] + a row of -1e30 + the mean key (64)
constexpr int FB_LDS = FB_K_BYTES + 2 * FB_DS_BYTES + 4 * FB_TILE_ELEMS * 2 + FB_STATS;   // 148352 B
constexpr int FB_VIMG_BYTES = 64 * FBS * 2;                   // + V of <= 64 keys for the short causal heads' delta pass
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// Fragments out of the swizzled images.  ff_swz(r) depends on (r >> 1) & 7 only, so for rows that differ by a multiple of 16 a lane's
// swizzled chunk is the same: every fragment address is one of a few per-lane byte offsets (FbOffsets, computed once) plus an
// IMMEDIATE (image, 16- or 32-row block) - as cheap to address as a padded layout.
struct FbOffsets {
    int row[4];      // row fragment, k-step ks: (r, hh) -> row r, chunk (2 ks + hh) ^ swz(r)
    int tr[2][2];    // 32x32x16 transposed fragment, [column block db][lo / hi half]: rows 4 (G >> 1) + (i >> 2) (+ 8)
    int tr16[2];     // 16x16x32 transposed fragment of columns d0 .., [lo / hi]: rows 8 g + (i >> 2) (+ 4)
};
__device__ __forceinline__ FbOffsets fb_offsets(int lane, int d0) {
    FbOffsets o;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o.row[ks] = r * 128 + (((2 * ks + hh) ^ ff_swz(r)) << 4);
    const int G = lane >> 4, i = lane & 15, sub = 8 * (i & 1);
    const int rt = 4 * (G >> 1) + (i >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int chunk = 4 * db + 2 * (G & 1) + ((i & 3) >> 1);
        o.tr[db][0] = rt * 128 + ((chunk ^ ff_swz(rt)) << 4) + sub;
        o.tr[db][1] = (rt + 8) * 128 + ((chunk ^ ff_swz(rt + 8)) << 4) + sub;
    }
    const int r16 = 8 * G + (i >> 2), c16 = (d0 >> 3) + ((i & 3) >> 1);
    o.tr16[0] = r16 * 128 + ((c16 ^ ff_swz(r16)) << 4) + sub;
    o.tr16[1] = (r16 + 4) * 128 + ((c16 ^ ff_swz(r16 + 4)) << 4) + sub;
    return o;
}
// img: image start; blk: byte offset of the 16- / 32-row block (a multiple of 2048: compile-time where the caller's is)
__device__ __forceinline__ bf16x8 fbs_row(const bf16_t* img, int blk, int off) { return *(const bf16x8*)((const char*)img + blk + off); }
__device__ __forceinline__ bf16x8 fbs_tr(const bf16_t* img, int blk, const int (&off)[2]) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)((const char*)img + blk + off[0]));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)((const char*)img + blk + off[1]));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// 16x16x32 operand by transposed LDS reads: lane (i = l & 15, g = l >> 4) gets img[row0 + 8 g + j][col0 + i], j = 0..7
template <int LD>
__device__ __forceinline__ bf16x8 frag_tr16(const bf16_t* img, int row0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const bf16_t* p = img + (row0 + 8 * g + (i >> 2)) * LD + col0 + 4 * (i & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ uint32_t scale_sub_bf16_pair(uint32_t x, float f, float sub_lo, float sub_hi) {      // (x * f - sub) per half
    const bf16_t lo = (bf16_t)(__uint_as_float(x << 16) * f - sub_lo), hi = (bf16_t)(__uint_as_float(x & 0xffff0000u) * f - sub_hi);
    return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}

// sum over the 16 lanes of a DPP row, in every lane of the row: four DPP adds, no LDS
__device__ __forceinline__ float row_sum_dpp(float v) {
    v += wave_dpp<0xB1>(v);             // quad_perm [1,0,3,2]
    v += wave_dpp<0x4E>(v);             // quad_perm [2,3,0,1]
    v += wave_dpp<0x141>(v);            // row_half_mirror
    v += wave_dpp<0x140>(v);            // row_mirror
    return v;
}

template <bool DROP, bool MASKED, bool BAND = false>
__global__ __launch_bounds__(FB_THREADS, 2) void sdpa_bwd_fused_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                         const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ o, const bf16_t* __restrict__ o_lo, const float* __restrict__ lse,
                                                                         bf16_t* __restrict__ dq, bf16_t* __restrict__ dk_, bf16_t* __restrict__ dv,
                                                                         const int32_t* __restrict__ k_len, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                                                                         int causal, int window, float scale, uint32_t dseed, uint32_t dthr, float dscale,
                                                                         float* __restrict__ halo = nullptr, int halo_slots = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_fb[];
    // BAND (self-attention inside a +-window band over MORE keys than one workgroup holds: the long-form configuration, T = 2000):
    // blockIdx.y = key block of FB_KEYS keys.  The workgroup owns the dK / dV of its keys and walks only the query tiles whose band
    // touches them; per tile only the waves whose 64 keys lie in the band work, and the dQ product runs over those waves' keys.
    // A query tile near a block boundary gets dQ contributions from TWO workgroups: each writes its fp32 partial to a slab of
    // `halo` and sdpa_band_halo_kernel adds the two (fixed order, no atomics); tiles of one block store bf16 dQ directly.
    const int kblk0 = BAND ? (int)blockIdx.y * FB_KEYS : 0;      // first key of this workgroup's block
    const int Tkg = Tk;                                          // keys of the head; Tk below = keys of this block
    if (BAND) Tk = min(FB_KEYS, Tkg - kblk0);
    bf16_t* Kimg = (bf16_t*)smem_fb;
    bf16_t* dSimg = (bf16_t*)(smem_fb + FB_K_BYTES);                               // two buffers of FB_KEYS x 32
    bf16_t* tiles = (bf16_t*)(smem_fb + FB_K_BYTES + 2 * FB_DS_BYTES);
    float* stats = (float*)(smem_fb + FB_K_BYTES + 2 * FB_DS_BYTES + 4 * FB_TILE_ELEMS * 2);
    float* s_neg = stats + 128;
    float* kbar = stats + 160;      // the head's mean key times scale * log2(e)
    // Short causal heads without dropout take delta from their own p and dP (phase_keys): exact, so neither remedy below is needed for them.
    const bool own_delta = !DROP && MASKED && !BAND && Tk <= 64;
    const bool centred = !BAND && !own_delta;      // the K image holds the keys minus their mean (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const bf16_t* qb = q + (size_t)b * Tq * ldq + h * DK;
    const bf16_t* dob = d_o + (size_t)b * Tq * ldo + h * DK;
    const bf16_t* ob = o + (size_t)b * Tq * ldo + h * DK;
    // BAND only: the low-order piece of O (store_rows_T_lo) for delta - the remedies of the one-block form below (centred keys, dK's mean over the
    // keys taken out) need ONE mean for all the keys of a head, which no workgroup of the band form holds.  Without a piece the load is unused.
    const bf16_t* olb = ((BAND && o_lo) ? o_lo : o) + (size_t)b * Tq * ldo + h * DK;
    const bool has_lo = BAND && o_lo != nullptr;
    const bf16_t* kb = k + ((size_t)b * Tkg + kblk0) * ldk + h * DK;
    const bf16_t* vb = v + ((size_t)b * Tkg + kblk0) * ldv + h * DK;
    const float* lseb = lse + ((size_t)b * H + h) * Tq;
    const int klen_g = min(k_len ? k_len[b] : Tkg, Tkg);       // valid keys of the head
    const int klen = BAND ? max(0, min(klen_g - kblk0, Tk)) : klen_g;      // ... of this block (block-local index)
    const int kk0 = 64 * w;
    const bool active = kk0 < Tk;                 // wave-uniform: this wave owns keys of the head
    const int nks = (Tk + 31) >> 5;               // 32-key steps of the dQ product
    const float sc2 = scale * LOG2E;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // ---- K * scale * log2(e) of every key -> LDS (rows >= klen are zeros: padded keys never contribute); dS images start
    //      as zeros (rows of waves without keys are never written and are read by the unrolled dQ loop)
    // all eight 16-byte pieces of a thread are requested before the first is used (as a rolled loop every piece waited for its own
    // round trip to memory: eight in a row, with every workgroup of the chip in this prologue at once)
    constexpr int KPIECES = FB_KEYS * 8 / FB_THREADS;
    u32x4 kx[KPIECES];
#pragma unroll
    for (int i = 0; i < KPIECES; ++i) {
        const int c = tid + i * FB_THREADS, row = c >> 3, ch = c & 7;
        kx[i] = *(const u32x4*)(kb + (size_t)min(row, max(klen - 1, 0)) * ldk + ch * 8);      // unpredicated (a valid row of this block); rows >= klen are zeroed below
    }
    for (int c = tid; c < 2 * FB_DS_BYTES / 16; c += FB_THREADS) *(u32x4*)((char*)dSimg + c * 16) = zero4;      // under the loads' latency
    if (tid < 32) s_neg[tid] = -1.0e30f;
    bf16_t* Vimg = (bf16_t*)(smem_fb + FB_LDS);      // short causal heads only (FB_VIMG_BYTES more LDS in that launch): V of the 64 keys, laid out as K's image
    if constexpr (!DROP && MASKED && !BAND) {
        if (Tk <= 64) {
            const int row = tid >> 3, ch = tid & 7;
            const u32x4 raw = *(const u32x4*)(vb + (size_t)min(row, max(klen - 1, 0)) * ldv + ch * 8);
            *(u32x4*)(Vimg + row * FBS + ((ch ^ ff_swz(row)) << 3)) = row < klen ? raw : zero4;
        }
    }
    // ---- this wave's V rows as B-operand fragments (key on the lane)
    bf16x8 vf[2][4];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
        const int row = kk0 + 32 * kbk + (lane & 31);
        const bool ok = row < klen;
        const bf16_t* src = vb + (size_t)(ok ? row : 0) * ldv + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 x = *(const bf16x8*)(src + 16 * ks);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (bf16_t)0.f;
            }
            vf[kbk][ks] = x;
        }
    }
    // ---- query tile prefetch (registers): threads 0..255 one 16-byte piece of Q, 256..511 of dO (+ the O piece for delta)
    const int t2 = tid & 255, prow = t2 >> 3, pch = t2 & 7;
    const bool is_do = tid >= 256;
    const bf16_t* srcA = is_do ? dob : qb;      // one code path for both halves of the workgroup
    const size_t ldA = is_do ? (size_t)ldo : (size_t)ldq;
    u32x4 pa = zero4, po = zero4, po2 = zero4;      // po2: BAND only
    float pl = 0.f;
    // FB_PREFETCH only ISSUES the loads of a tile (rows clamped into the matrix, no use of the loaded values); FB_COMMIT, a tile later,
    // zeroes what lies past the end and stores the tile.  With the selects on the loaded values written next to the loads the compiler
    // waited for the memory round trip right there - once per tile, in every wave, between the barrier and the dQ block.
#define FB_PREFETCH(Q0)                                                                                   \
    do {                                                                                                  \
        const size_t row_ = (size_t)min((Q0) + prow, Tq - 1);                                             \
        pa = *(const u32x4*)(srcA + row_ * ldA + pch * 8);                                                \
        po = *(const u32x4*)(ob + row_ * ldo + pch * 8);                                                  \
        if (has_lo) po2 = *(const u32x4*)(olb + row_ * ldo + pch * 8);                                    \
        pl = lseb[min((Q0) + prow, Tq - 1)];                                                              \
    } while (0)
    // rows past the end / rows that saw no key get -1e30: p = 0.  (Every thread loads one lse value: no divergent path.)
#define FB_COMMIT(BUF, Q0)                                                                                \
    do {                                                                                                  \
        if ((Q0) + prow >= Tq) { pa = zero4; po = zero4; po2 = zero4; }                                   \
        const float pl_ = ((Q0) + prow >= Tq || pl == -INFINITY) ? -1.0e30f : -pl * LOG2E;                \
        bf16_t* T_ = tiles + (BUF) * 2 * FB_TILE_ELEMS + (is_do ? FB_TILE_ELEMS : 0);                     \
        *(u32x4*)(T_ + prow * FBS + ((pch ^ ff_swz(prow)) << 3)) = pa;                                                         \
        float* st_ = stats + (BUF) * 64;                                                                  \
        /* 8 lanes per row.  dO half: delta = rowsum(dO o O).  Q half: q . (mean key), the shift that the centred K image takes out of the scores */ \
        float x_[8];                                                                                      \
        if (is_do) {                                                                                      \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                            \
                x_[2 * e_] = __uint_as_float(po[e_] << 16) + (BAND ? __uint_as_float(po2[e_] << 16) : 0.f);             \
                x_[2 * e_ + 1] = __uint_as_float(po[e_] & 0xffff0000u) + (BAND ? __uint_as_float(po2[e_] & 0xffff0000u) : 0.f); \
            }                                                                                             \
        } else if (centred) {                                                                             \
            const f32x4_t k0_ = *(const f32x4_t*)(kbar + 8 * pch), k1_ = *(const f32x4_t*)(kbar + 8 * pch + 4); \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { x_[e_] = k0_[e_]; x_[4 + e_] = k1_[e_]; }   \
        } else {                                                                                          \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) x_[e_] = 0.f;                                \
        }                                                                                                 \
        float d_ = 0.f;                                                                                   \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                \
            d_ += __uint_as_float(pa[e_] << 16) * x_[2 * e_];                                             \
            d_ += __uint_as_float(pa[e_] & 0xffff0000u) * x_[2 * e_ + 1];                                 \
        }                                                                                                 \
        d_ += __shfl_xor(d_, 1, 64);                                                                      \
        d_ += __shfl_xor(d_, 2, 64);                                                                      \
        d_ += __shfl_xor(d_, 4, 64);                                                                      \
        if (pch == 0) {                                                                                   \
            if (is_do) st_[32 + prow] = -d_;                                                              \
            else st_[prow] = pl_ + d_;      /* -lse log2(e) + q . mean key: p = exp2(S' + this), S' from the centred keys */ \
        }                                                                                                 \
    } while (0)

    f32x16 dka[2][2], dva[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dka[a][c][i] = 0.f; dva[a][c][i] = 0.f; }
    const int ntiles = (Tq + FB_QT - 1) / FB_QT;
    const int d0 = 16 * (w >> 1), q0l = 16 * (w & 1);     // this wave's 16 x 16 block of dQ^T
    const FbOffsets fo = fb_offsets(lane, d0);
    bf16_t* dqb = dq + (size_t)b * Tq * ldq + h * DK;
    // dS image addressing: 8-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7)
    const int kjl = lane & 31;                              // key inside a 32-key block (f(key) does not depend on the block: 32 rows = 16 pairs)
    const int fw = (kjl >> 1) & 7;
    int ds_wr[4];                                           // element offsets of this lane's four pieces (queries 8 g + 4 hh ..) in its key row
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) ds_wr[g4] = kjl * FB_QT + 4 * ((2 * g4 + hh) ^ fw);
    int ds_rd[2];                                           // transposed-read offsets (rows 8 g + (i >> 2) and + 4), columns q0l + 4 (i & 3)
    {
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int pl4 = 0; pl4 < 2; ++pl4) {
            const int row = 8 * g + (i >> 2) + 4 * pl4;
            ds_rd[pl4] = row * FB_QT + 4 * (((q0l >> 2) + (i & 3)) ^ ((row >> 1) & 7));
        }
    }
    // ---- the two halves of a tile's work
    auto phase_keys = [&](int t) {      // S, dP, P, dS, dV^T, dK^T of this wave's keys against query tile t; dS -> image t & 1
        const int buf = t & 1, qt0 = t * FB_QT;
        if (BAND && (kblk0 + kk0 > qt0 + FB_QT - 1 + window || kblk0 + kk0 + 63 < qt0 - window)) return;      // none of this wave's keys is in the tile's band
        const bf16_t* Qt = tiles + buf * 2 * FB_TILE_ELEMS;
        const bf16_t* Dt = Qt + FB_TILE_ELEMS;
        const float* s_l = stats + buf * 64;
        const float* s_d = s_l + 32;
        bf16_t* dSw = dSimg + buf * (FB_DS_BYTES / 2);
        if constexpr (!DROP && MASKED && !BAND) {
            // Short causal heads (the decoder's self-attention, Tk <= 64: this wave holds EVERY key of the head) take delta as
            // sum_j p_j dP_j / sum_j p_j from the p and dP of THIS kernel - what a softmax backward computes - instead of
            // rowsum(dO o O) from the rounded O.  The two are the same number in exact arithmetic; in bf16, where the rows of V are
            // close to one another, dP_j - delta cancels and the delta from the stored O costs ~3 x the error floor the inputs'
            // rounding sets (1 - cos(dQ) 4.9e-4 vs 1.6e-4 on the kernel: tools/sdpa_delta_ab.py; the forms in fp64:
            // tools/sdpa_delta_forms.py).  A first look at S and dP in 16 x 16 blocks (MFMA 16x16x32, query rows x key lanes: a
            // rolled loop of a few registers - the same pass in the 32 x 32 form of the code below cost 40 - 90 spilled registers),
            // V from an image staged beside K's; sums over the keys by four DPP adds; the result replaces the tile's -delta row in
            // LDS, read back below by this wave only (no other wave has keys).
            if (Tk <= 64) {
                const int li = lane & 15, lg = lane >> 4;
                float* s_dw = stats + buf * 64 + 32;
#pragma nounroll
                for (int qb = 0; qb < 2; ++qb) {      // 16 queries at a time: register r = query 16 qb + 4 lg + r
                    const int row = 16 * qb + li;
                    const int qo0 = row * FBS + ((lg ^ ff_swz(row)) << 3), qo1 = row * FBS + (((4 + lg) ^ ff_swz(row)) << 3);
                    const f32x4_t l4 = *(const f32x4_t*)(s_l + 16 * qb + 4 * lg);
                    f32x4_t sp = {0.f, 0.f, 0.f, 0.f}, spd = sp;
#pragma nounroll
                    for (int kb = 0; kb < (Tk + 15) >> 4; ++kb) {
                        const int key = 16 * kb + li;
                        const int ko0 = key * FBS + ((lg ^ ff_swz(key)) << 3), ko1 = key * FBS + (((4 + lg) ^ ff_swz(key)) << 3);
                        f32x4_t sv = l4, dv4 = {0.f, 0.f, 0.f, 0.f};
                        sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Qt + qo0), *(const bf16x8*)(Kimg + ko0), sv, 0, 0, 0);
                        dv4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Dt + qo0), *(const bf16x8*)(Vimg + ko0), dv4, 0, 0, 0);
                        sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Qt + qo1), *(const bf16x8*)(Kimg + ko1), sv, 0, 0, 0);
                        dv4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Dt + qo1), *(const bf16x8*)(Vimg + ko1), dv4, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = visible(qt0 + 16 * qb + 4 * lg + r, key, klen_g, causal, window) ? __builtin_amdgcn_exp2f(sv[r]) : 0.f;
                            sp[r] += p;
                            spd[r] += p * dv4[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = row_sum_dpp(sp[r]), c = row_sum_dpp(spd[r]);
                        if (li == 0) s_dw[16 * qb + 4 * lg + r] = a > 0.f ? -c / a : 0.f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            const int key0 = kk0 + 32 * kbk, kj = key0 + kjl;
            const float* s_lk = kj < klen ? s_l : s_neg;      // padded keys: p = 0 without a test per element
            f32x16 st, dp;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t l4 = *(const f32x4_t*)(s_lk + 8 * g4 + 4 * hh);
                const f32x4_t d4 = *(const f32x4_t*)(s_d + 8 * g4 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[4 * g4 + e] = l4[e];
                    dp[4 * g4 + e] = DROP ? 0.f : d4[e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbs_row(Qt, 0, fo.row[ks]), fbs_row(Kimg, key0 * 128, fo.row[ks]), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbs_row(Dt, 0, fo.row[ks]), vf[kbk][ks], dp, 0, 0, 0);
            }
            if (MASKED) {
                const int gk0 = kblk0 + key0;
                const bool need = (causal && gk0 + 31 > qt0) || (window >= 0 && (gk0 + 31 - qt0 > window || qt0 + 31 - gk0 > window));
                if (need) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (!visible(qt0 + acc_row(i, lane), kblk0 + kj, klen_g, causal, window)) st[i] = -1.0e30f;
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4_t d4 = {0.f, 0.f, 0.f, 0.f};
                if (DROP) d4 = *(const f32x4_t*)(s_d + 8 * g4 + 4 * hh);
                // dropout mask: one hash serves the element PAIR (key 2m, key 2m + 1) of a query, and the two keys of a pair sit in
                // neighbouring lanes here - each lane hashes for one of two consecutive queries and takes the other from its
                // neighbour (8 hashes + 8 lane swaps per 16 elements instead of 16 hashes: 3 integer multiplies each)
                uint32_t hsh[4] = {0u, 0u, 0u, 0u};
                if (DROP) {
                    const uint32_t kcl = (uint32_t)min(kblk0 + kj, Tkg - 1), tkp = (uint32_t)((Tkg + 1) & ~1), bhq = ((uint32_t)(b * H + h)) * Tq;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int em = e + (lane & 1);
                        const uint32_t el_m = (bhq + min(qt0 + 8 * g4 + 4 * hh + em, Tq - 1)) * tkp + kcl;
                        const uint32_t h_m = drop_hash(el_m >> 1, dseed);
                        const uint32_t h_o = (uint32_t)__shfl_xor((int)h_m, 1, 64);      // the neighbour hashed the other query of the two for this key pair
                        hsh[e] = (lane & 1) ? h_o : h_m;
                        hsh[e + 1] = (lane & 1) ? h_m : h_o;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    const float p = __builtin_amdgcn_exp2f(st[i]);
                    if (DROP) {
                        const float keepf = drop_keep(hsh[e], min(kblk0 + kj, Tkg - 1) & 1, dthr) ? dscale : 0.f;
                        dp[i] = p * (dp[i] * keepf + d4[e]);     // dS / scale (d4 = -delta)
                        st[i] = p * keepf;                       // dropped probabilities feed dV
                    } else {
                        dp[i] = p * dp[i];                       // dS / scale: the factor is applied once, when dQ / dK are stored
                        st[i] = p;
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pfr = acc_to_frag(st, s2);
                const bf16x8 dfr = acc_to_frag(dp, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dva[kbk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbs_tr(Dt, s2 * 2048, fo.tr[db]), pfr, dva[kbk][db], 0, 0, 0);
                    dka[kbk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbs_tr(Qt, s2 * 2048, fo.tr[db]), dfr, dka[kbk][db], 0, 0, 0);
                }
                // dS -> [key][query] image: the fragment of k-step s2 holds registers 8 s2 .. 8 s2 + 7 = the pieces g4 = 2 s2, 2 s2 + 1
                const u32x4 dw = __builtin_bit_cast(u32x4, dfr);
                const u32x2_t lo2 = {dw[0], dw[1]}, hi2 = {dw[2], dw[3]};
                *(u32x2_t*)(dSw + (size_t)(key0 * FB_QT) + ds_wr[2 * s2]) = lo2;
                *(u32x2_t*)(dSw + (size_t)(key0 * FB_QT) + ds_wr[2 * s2 + 1]) = hi2;
            }
        }
    };
    auto phase_dq = [&](int t) {        // this wave's 16 x 16 block of dQ^T for query tile t, over every key
        const int buf = t & 1, qt0 = t * FB_QT;
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* dSr = dSimg + buf * (FB_DS_BYTES / 2);
        if (BAND) {      // only the waves that ran the key phase of this tile wrote dS rows: the product runs over their keys
            const int wv_lo = max(0, qt0 - window - kblk0) >> 6, wv_hi = min(min(qt0 + FB_QT - 1 + window, kblk0 + Tk - 1) - kblk0, FB_KEYS - 1) >> 6;
            for (int wv = wv_lo; wv <= wv_hi; ++wv) {
                bf16x8 ka[2], da[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ka[u] = fbs_tr(Kimg, 4096 * (2 * wv + u), fo.tr16);
                    const bf16_t* pr = dSr + 32 * (2 * wv + u) * FB_QT;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pr + ds_rd[0]));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pr + ds_rd[1]));
                    da[u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0], da[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1], da[1], acc1, 0, 0, 0);
            }
        } else
        for (int s4 = 0; s4 < nks; s4 += 4) {     // rows past the last key: K rows and dS rows are zeros
            bf16x8 ka[4], da[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ka[u] = fbs_tr(Kimg, 4096 * (s4 + u), fo.tr16);
                const bf16_t* pr = dSr + 32 * (s4 + u) * FB_QT;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pr + ds_rd[0]));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pr + ds_rd[1]));
                da[u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0], da[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1], da[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[2], da[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[3], da[3], acc1, 0, 0, 0);
        }
        const int qi = qt0 + q0l + (lane & 15);     // accumulator: column = query, rows = d0 + 4 (l >> 4) + r
        if (BAND) {
            // key blocks whose keys the tile's band touches: one -> this workgroup has the whole sum; two -> fp32 partial to the slab
            const int jl = max(qt0 - window, 0) / FB_KEYS, jh = min(qt0 + FB_QT - 1 + window, Tkg - 1) / FB_KEYS;
            if (jl != jh) {
                const int nbnd = (int)gridDim.y - 1, side = (int)blockIdx.y == jh ? 1 : 0;
                const int t_first = max(0, (FB_KEYS * (jl + 1) - (FB_QT - 1) - window + FB_QT - 1) / FB_QT);
                float* dst = halo + ((((size_t)bh * nbnd + jl) * 2 + side) * halo_slots + (t - t_first)) * (FB_QT * DK);
                f32x4_t pv;
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[e] = (acc0[e] + acc1[e]) * LN2;
                *(f32x4_t*)(dst + (q0l + (lane & 15)) * DK + d0 + 4 * (lane >> 4)) = pv;
                return;
            }
        }
        if (qi < Tq) {
            bf16x4 ov;       // K was staged times scale * log2(e): dQ = scale dS K = dS K' / log2(e)
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)((acc0[e] + acc1[e]) * LN2);
            *(bf16x4*)(dqb + (size_t)qi * ldq + d0 + 4 * (lane >> 4)) = ov;
        }
    };

    // BAND: the query tiles whose band [qt0 - window, qt0 + 31 + window] touches this block's keys
    const int t_lo = BAND ? max(0, (kblk0 - (FB_QT - 1) - window + FB_QT - 1) / FB_QT) : 0;
    const int t_hi = BAND ? min(ntiles, (kblk0 + Tk + window + FB_QT - 1) / FB_QT) : ntiles;
    FB_PREFETCH(t_lo * FB_QT);
    // ---- the K image holds the keys MINUS their mean over the head (times scale * log2(e)).  Why: dQ = sum_j dS_j K_j with sum_j dS_j = 0
    // in exact arithmetic, so a component common to all keys contributes nothing - but dS_j = p_j (dP_j - delta) carries delta's error eps
    // (delta = rowsum(dO o O) from the bf16-ROUNDED O: 2^-9 |O| per coordinate) into every key alike, and the product with the uncentred
    // keys puts eps * (mean key) into dQ: where the rows of K and of V share a component (a bias behind a LayerNorm) that term was the
    // whole error - gradient cosine 0.989 of the top encoder layer's Q / K projections at the full-size configuration, 0.90 for the
    // decoder's cross attention.  With centred keys eps only meets (p-weighted mean key - mean key).  1 - cos(dQ), fp64 emulation at 500
    // keys, common component 1 / 3 sigma: 7e-4 / 5e-2 uncentred, 1e-5 / 3e-5 centred, floor from rounding the inputs 7e-6 / 3e-5
    // (tools/sdpa_delta_forms_500.py) - better than a delta from an fp32 O (2e-5 / 2e-4), at no memory traffic.  The scores lose the
    // per-query constant q . mean key, which FB_COMMIT adds to the row's -lse log2(e); the centred image also rounds smaller numbers.
    // BAND: no centring (the two key blocks a tile's band touches would need one common mean).
    float kb8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // this thread's chunk (tid & 7) of the mean key
    if (centred) {
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPIECES; ++i) {
            const int row = (tid + i * FB_THREADS) >> 3;
            if (row < klen) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    part[2 * e] += __uint_as_float(kx[i][e] << 16);
                    part[2 * e + 1] += __uint_as_float(kx[i][e] & 0xffff0000u);
                }
            }
        }
        // lanes l, l + 8, .., l + 56 of a wave hold the same chunk: three exchanges, then the eight waves' sums through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            part[e] += __shfl_xor(part[e], 8, 64);
            part[e] += __shfl_xor(part[e], 16, 64);
            part[e] += __shfl_xor(part[e], 32, 64);
        }
        float* red = (float*)tiles;      // [FB_WAVES][64] floats in the query tiles' space, which is written only after the barriers below
        if (lane < 8) {
            *(f32x4_t*)(red + w * 64 + 8 * lane) = f32x4_t{part[0], part[1], part[2], part[3]};
            *(f32x4_t*)(red + w * 64 + 8 * lane + 4) = f32x4_t{part[4], part[5], part[6], part[7]};
        }
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < FB_WAVES; ++wv) sum += red[wv * 64 + tid];
            kbar[tid] = klen > 0 ? sum / (float)klen * sc2 : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) kb8[e] = kbar[8 * (tid & 7) + e];
    }
#pragma unroll
    for (int i = 0; i < KPIECES; ++i) {      // (K - mean) * scale * log2(e) -> LDS (rows >= klen stay zero)
        const int c = tid + i * FB_THREADS, row = c >> 3, ch = c & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) kx[i][e] = row < klen ? scale_sub_bf16_pair(kx[i][e], sc2, kb8[2 * e], kb8[2 * e + 1]) : 0u;
        *(u32x4*)(Kimg + row * FBS + ((ch ^ ff_swz(row)) << 3)) = kx[i];
    }
    FB_COMMIT(t_lo & 1, t_lo * FB_QT);
    if (t_lo + 1 < t_hi) FB_PREFETCH((t_lo + 1) * FB_QT);
    __syncthreads();         // K image, zeroed dS images, first tile
    // One barrier per tile.  Before barrier t: key phase of tile t (writes dS image t & 1, last read by the dQ block of
    // tile t - 2, i.e. before barrier t - 1) and the store of tile t + 1 into tile buffer (t + 1) & 1 (last read by the key
    // phase of tile t - 1).  After it: the dQ block of tile t, while other waves already run the key phase of tile t + 1.
    // (Giving the two waves of a SIMD opposite phase orders through a wave-uniform switch was 10 us SLOWER: 92 vs 82.)
#ifndef SDPA_SKIP
#define SDPA_SKIP 0      // diagnostic builds only (make sdpa-skip SDPA_SKIP=n): 1 = no dQ phase, 2 = no key phase, 4 = neither; WRONG results, timing only
#endif
    for (int t = t_lo; t < t_hi; ++t) {
        if (active && !(SDPA_SKIP & 6)) phase_keys(t);
        if (t + 1 < t_hi) FB_COMMIT((t + 1) & 1, (t + 1) * FB_QT);
        __syncthreads();
        if (t + 2 < t_hi) FB_PREFETCH((t + 2) * FB_QT);
        if (!(SDPA_SKIP & 5)) phase_dq(t);
    }
#undef FB_PREFETCH
#undef FB_COMMIT
    // dV leaves first: its stores run under the reduction below (with both behind the two barriers the whole chip stored at once: +4 us)
    if (active) {
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) store_rows_T(dva[kbk], 1.f, dv + ((size_t)b * Tkg + kblk0) * ldv + h * DK, ldv, kk0 + 32 * kbk, Tk, lane);
    }
    // ---- the rows of dK sum to zero over the keys of a head (sum_j dK_j = sum_q Q_q sum_j dS_qj, and every query's dS sums to zero over
    // the keys it sees, whatever the mask): what the accumulators hold beyond that is delta's error again - dS_qj = p_qj (dP_qj - delta_q)
    // adds -eps_q p_qj Q_q to dK_j, for near-uniform p the SAME vector for every key, and the K projection's weight gradient
    // dW_k = sum_j dK_j x_j^T multiplies that vector by T times the mean input row: the top encoder layer's w_ks measured 0.988 (cosine
    // against the oracle at full size) with the keys centred for dQ and 0.9998 once this mean is taken out.  fp64 emulation at 500 keys,
    // common components of 1 - 3 sigma in Q, K, V and the layer input: 1 - cos(dW_k) 7e-4 .. 4e-2 as computed, 7e-6 .. 2e-5 with the
    // mean over the keys removed - the floor of the inputs' rounding, below a delta from the kernel's own p and dP (tools/sdpa_dk_mean.py).
    // Once per kernel: 32 half-wave sums by DPP, 2.5 KiB through LDS, two barriers: +0.5 us at the encoder's shape, +2.5 us for the
    // decoder's cross attention (A/B in one process against the kernel without it).  BAND: the sum is over ALL key blocks - not done.
    if (centred) {
        float* red = (float*)tiles;      // the query tiles are dead: every wave's last key phase lies in front of the loop's last barrier
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = dka[0][db][i] + dka[1][db][i];      // this wave's 64 keys (rows of padded keys are zeros); lanes = the 32 keys of a block
                v += wave_dpp<0xB1>(v);             // quad_perm [1,0,3,2]
                v += wave_dpp<0x4E>(v);             // quad_perm [2,3,0,1]
                v += wave_dpp<0x141>(v);            // row_half_mirror
                v += wave_dpp<0x140>(v);            // row_mirror: every lane of a 16-lane row holds the row's sum
                v += wave_dpp<0x142, 0xA>(v);       // row_bcast15 into rows 1 and 3: lanes 16 - 31 / 48 - 63 hold the half-wave's sum
                if ((lane & 31) == 16) red[w * 64 + 32 * db + acc_row(i, lane)] = v;
            }
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < FB_WAVES; ++wv) sum += red[wv * 64 + tid];
            red[FB_WAVES * 64 + tid] = klen > 0 ? sum / (float)klen : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m = red[FB_WAVES * 64 + 32 * db + acc_row(i, lane)];
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
                    if (kk0 + 32 * kbk + (lane & 31) < klen) dka[kbk][db][i] -= m;
            }
    }
    if (active) {
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) store_rows_T(dka[kbk], scale, dk_ + ((size_t)b * Tkg + kblk0) * ldk + h * DK, ldk, kk0 + 32 * kbk, Tk, lane);
    }
}

// Sum of the two fp32 dQ partials of the query tiles that straddle a key-block boundary of sdpa_bwd_fused_bf16_kernel<.., BAND>
// (side 0 = the lower block's keys, then side 1: a fixed order) -> bf16 dQ rows.  grid (slots, boundaries, B * H), 256 threads.
__global__ __launch_bounds__(256) void sdpa_band_halo_kernel(const float* __restrict__ halo, bf16_t* __restrict__ dq, int H, int Tq, int Tk, int ldq, int window,
                                                             int halo_slots) {
    const int slot = blockIdx.x, bnd = blockIdx.y, bh = blockIdx.z, nbnd = gridDim.y;
    const int t_first = max(0, (FB_KEYS * (bnd + 1) - (FB_QT - 1) - window + FB_QT - 1) / FB_QT);
    const int t = t_first + slot, qt0 = t * FB_QT;
    if (qt0 >= Tq) return;
    const int jl = max(qt0 - window, 0) / FB_KEYS, jh = min(qt0 + FB_QT - 1 + window, Tk - 1) / FB_KEYS;
    if (jl != bnd || jh != bnd + 1) return;      // not a tile shared across this boundary
    const float* s0 = halo + ((((size_t)bh * nbnd + bnd) * 2 + 0) * halo_slots + slot) * (FB_QT * DK);
    const float* s1 = s0 + (size_t)halo_slots * (FB_QT * DK);
    const int b = bh / H, h = bh - b * H;
    const int row = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
    const int qi = qt0 + row;
    if (qi >= Tq) return;
    float x[8], y[8];
    load8<float>(s0 + row * DK + c8, x);
    load8<float>(s1 + row * DK + c8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    store8<bf16_t>(dq + ((size_t)b * Tq + qi) * ldq + h * DK + c8, x);
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
                            int ldq, int ldk, int ldv, int ldo, int causal, int window, float scale, float drop_p, uint32_t dseed, void* o_lo,
                            int dtype, void* stream) {
    if (!q || !k || !v || !o || !lse) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: null pointer");
    if (o_lo && (dtype != ASR_BF16 || ((uintptr_t)o_lo % 16) != 0)) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: o_lo is the low-order piece of a bf16 output (16-byte aligned, the layout of o)");
    if (int rc = check_common("asr_sdpa_fwd", B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo)) return rc;
    if (drop_p < 0.f || drop_p >= 1.f) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: bad dropout p=%f", drop_p);
    if ((double)B * H * Tq * (Tk + 1) >= 4294967296.0) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: B*H*Tq*Tk exceeds the 32-bit dropout counter");
    const uint32_t dthr = drop_thr16(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    const bool ff_fused = dtype == ASR_BF16 && mfma_ok(dk, ldq, ldk, ldv, ldo, q, k, v, o) && Tk <= FF_KEYS;      // else (more keys than LDS holds): the tiled kernel of round 1
    if (ff_fused) {   // K and V of a head fit LDS: one workgroup per (b, h)
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)sdpa_fwd_fused_bf16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_fwd_fused_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_fwd_fused_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_fwd_fused_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_fwd_pair_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
            attr = true;
        }
        // the backward pass of these shapes (every key of a head in one workgroup) does not read a low-order piece: a buffer given anyway is cleared
        if (o_lo && hipMemset2DAsync(o_lo, (size_t)ldo * 2, 0, (size_t)H * dk * 2, (size_t)B * Tq, st) != hipSuccess) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: clearing o_lo failed");
#define FF_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale
        const bool masked = causal || window >= 0;
        if (dthr && masked) sdpa_fwd_fused_bf16_kernel<true, true><<<B * H, FF_THREADS, FF_LDS, st>>>(FF_ARGS);
        else if (dthr) sdpa_fwd_fused_bf16_kernel<true, false><<<B * H, FF_THREADS, FF_LDS, st>>>(FF_ARGS);
        else if (masked) sdpa_fwd_fused_bf16_kernel<false, true><<<B * H, FF_THREADS, FF_LDS, st>>>(FF_ARGS);
        else if (asr_option(ASR_OPT_SDPA_PAIR)) sdpa_fwd_pair_bf16_kernel<<<B * H, FF_THREADS, FF_LDS, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, scale);
        else sdpa_fwd_fused_bf16_kernel<false, false><<<B * H, FF_THREADS, FF_LDS, st>>>(FF_ARGS);
#undef FF_ARGS
    } else if (dtype == ASR_BF16 && mfma_ok(dk, ldq, ldk, ldv, ldo, q, k, v, o)) {
        const int grid = ceil_div(Tq, 128) * H * B;
        if (dthr) sdpa_fwd_bf16_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale, (bf16_t*)o_lo);
        else sdpa_fwd_bf16_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale, (bf16_t*)o_lo);
    } else {
        dim3 grid(Tq, H, B);
        const size_t lds = (size_t)(Tk + dk) * sizeof(float);
        if (lds > 64 * 1024) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: generic path needs Tk+dk <= 16384");
        if (dtype == ASR_F32) sdpa_fwd_generic_kernel<float><<<grid, 64, lds, st>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, lse, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        else if (dtype == ASR_BF16) {
            // this path writes no low-order piece: zeros (the backward pass then takes delta from o alone)
            if (o_lo && hipMemset2DAsync(o_lo, (size_t)ldo * 2, 0, (size_t)H * dk * 2, (size_t)B * Tq, st) != hipSuccess) ASR_FAIL(ASR_EINVAL, "asr_sdpa_fwd: clearing o_lo failed");
            sdpa_fwd_generic_kernel<bf16_t><<<grid, 64, lds, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, k_len, H, Tq, Tk, dk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale);
        }
        else ASR_FAIL(ASR_EDTYPE, "asr_sdpa_fwd: dtype %d", dtype);
    }
    ASR_CHECK_LAUNCH("asr_sdpa_fwd");
    return ASR_OK;
}

// single-pass band kernel: bf16 MFMA shapes, self-attention (Tq == Tk) inside a +-window band, more keys than one workgroup holds,
// and the band of a 32-query tile touches at most two 512-key blocks
static bool sdpa_band_shape(int Tq, int Tk, int dk, int causal, int window, int dtype) {
    return dtype == ASR_BF16 && dk == DK && Tq == Tk && Tk > FB_KEYS && !causal && window >= 0 && 2 * window + FB_QT <= FB_KEYS;
}
static int sdpa_band_slots(int window) { return (2 * window + FB_QT - 1) / FB_QT + 2; }

extern "C" size_t asr_sdpa_bwd_workspace_bytes(int B, int H, int Tq, int Tk, int dk, int causal, int window, int dtype) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
    size_t need = (size_t)B * H * Tq * sizeof(float);
    if (sdpa_band_shape(Tq, Tk, dk, causal, window, dtype)) {
        const size_t halo = (size_t)B * H * (ceil_div(Tk, FB_KEYS) - 1) * 2 * sdpa_band_slots(window) * FB_QT * DK * sizeof(float);
        if (halo > need) need = halo;
    }
    return need;
}

extern "C" int asr_sdpa_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta, size_t delta_bytes,
                            void* dq, void* dk_, void* dv, const int32_t* k_len, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk, int ldv, int ldo,
                            int causal, int window, float scale, float drop_p, uint32_t dseed, const void* o_lo, int dtype, void* stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk_ || !dv) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: null pointer");
    if (o_lo && (dtype != ASR_BF16 || ((uintptr_t)o_lo % 16) != 0)) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: o_lo is the low-order piece of a bf16 output (16-byte aligned, the layout of o)");
    if (int rc = check_common("asr_sdpa_bwd", B, H, Tq, Tk, dk, ldq, ldk, ldv, ldo)) return rc;
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_sdpa_bwd: dtype %d", dtype);
    if (delta_bytes < (size_t)B * H * Tq * sizeof(float)) ASR_FAIL(ASR_EWORKSPACE, "asr_sdpa_bwd: scratch of %zu bytes, need at least B*H*Tq floats = %zu", delta_bytes, (size_t)B * H * Tq * sizeof(float));
    if (drop_p < 0.f || drop_p >= 1.f) ASR_FAIL(ASR_EINVAL, "asr_sdpa_bwd: bad dropout p=%f", drop_p);
    const uint32_t dthr = drop_thr16(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = B * Tq * H;
    const bool mfma = dtype == ASR_BF16 && mfma_ok(dk, ldq, ldk, ldv, ldo, q, k, v, d_o) && mfma_ok(dk, ldq, ldk, ldv, ldo, dq, dk_, dv, o);
    if (!mfma) {   // the MFMA dQ kernel computes delta itself
        if (dtype == ASR_F32) sdpa_delta_kernel<float><<<ceil_div(ngroups, 32), 256, 0, st>>>((const float*)o, (const float*)nullptr, (const float*)d_o, delta, B, H, Tq, dk, ldo);
        else sdpa_delta_kernel<bf16_t><<<ceil_div(ngroups, 32), 256, 0, st>>>((const bf16_t*)o, (const bf16_t*)o_lo, (const bf16_t*)d_o, delta, B, H, Tq, dk, ldo);
    }
    const bool fb_fused = mfma && Tk <= FB_KEYS;      // else: the band kernel (windowed attention over many keys) or the dQ + dK/dV pair of round 1
    if (fb_fused) {   // every key of a head fits one workgroup: single-pass backward
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS + FB_VIMG_BYTES);
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
            attr = true;
        }
#define FB_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, (const bf16_t*)o, (const bf16_t*)nullptr, lse, (bf16_t*)dq, (bf16_t*)dk_, (bf16_t*)dv, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale
        const bool masked = causal || window >= 0;
        // the only kernel of this path: it may carry an armed completion event (asr_stream_arm)
        if (dthr && masked) asr_launch_armed(sdpa_bwd_fused_bf16_kernel<true, true>, dim3(B * H), dim3(FB_THREADS), FB_LDS, st, FB_ARGS, (float*)nullptr, 0);
        else if (dthr) asr_launch_armed(sdpa_bwd_fused_bf16_kernel<true, false>, dim3(B * H), dim3(FB_THREADS), FB_LDS, st, FB_ARGS, (float*)nullptr, 0);
        else if (masked) asr_launch_armed(sdpa_bwd_fused_bf16_kernel<false, true>, dim3(B * H), dim3(FB_THREADS), FB_LDS + (Tk <= 64 ? FB_VIMG_BYTES : 0), st, FB_ARGS, (float*)nullptr, 0);
        else asr_launch_armed(sdpa_bwd_fused_bf16_kernel<false, false>, dim3(B * H), dim3(FB_THREADS), FB_LDS, st, FB_ARGS, (float*)nullptr, 0);
#undef FB_ARGS
    } else if (mfma && sdpa_band_shape(Tq, Tk, dk, causal, window, dtype) && delta_bytes >= asr_sdpa_bwd_workspace_bytes(B, H, Tq, Tk, dk, causal, window, dtype)
               && ((uintptr_t)delta % 16) == 0) {
        // long-form band: one workgroup per (b, h, 512-key block), single pass; dQ of the tiles on a block boundary through fp32 slabs
        static bool attr_b = false;
        if (!attr_b) {
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
            (void)hipFuncSetAttribute((const void*)sdpa_bwd_fused_bf16_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
            attr_b = true;
        }
        const int nblk = ceil_div(Tk, FB_KEYS), slots = sdpa_band_slots(window);
        const dim3 grid(B * H, nblk);
#define FBB_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, (const bf16_t*)o, (const bf16_t*)o_lo, lse, (bf16_t*)dq, (bf16_t*)dk_, (bf16_t*)dv, k_len, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, window, scale, dseed, dthr, dscale, delta, slots
        if (dthr) sdpa_bwd_fused_bf16_kernel<true, true, true><<<grid, FB_THREADS, FB_LDS, st>>>(FBB_ARGS);
        else sdpa_bwd_fused_bf16_kernel<false, true, true><<<grid, FB_THREADS, FB_LDS, st>>>(FBB_ARGS);
#undef FBB_ARGS
        sdpa_band_halo_kernel<<<dim3(slots, nblk - 1, B * H), 256, 0, st>>>(delta, (bf16_t*)dq, H, Tq, Tk, ldq, window, slots);
    } else if (mfma) {
        const int gq = ceil_div(Tq, 128) * H * B, gk = ceil_div(Tk, 128) * H * B;
#define SDPA_BWD(D)                                                                                                                                        \
    do {                                                                                                                                                   \
        sdpa_bwd_dq_bf16_kernel<D><<<gq, 256, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, (const bf16_t*)o, (const bf16_t*)o_lo, lse, delta, (bf16_t*)dq, \
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
