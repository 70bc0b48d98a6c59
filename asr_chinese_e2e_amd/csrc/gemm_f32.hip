// fp32 GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate - bitwise a chain of fmaf): the
// projections of the PARITY mode (dtype = fp32), which is compared with the reference's own CPU outputs at 1e-4 .. 1e-5
// (tests/test_model_gpu.py::test_fp32_matches_reference_golden, test_default_width_matches_reference_golden).
//
//   asr_gemm_f32 : C[m][n] (+)= act( sum_k opA(m, k) opB(k, n) + bias[n] )  (masked by `mask` for the ReLU backward)
//       trans_a = 0: A stored (M, K) row-major        trans_a = 1: A stored (K, M) row-major (its transpose is used)
//       trans_b = 0: B stored (K, N) row-major        trans_b = 1: B stored (N, K) row-major (nn.Linear's weight)
//     forward        y  = x W^T + b      : A = x  (0), B = W  (1)        attention.py:43-45,59, module.py:70-71
//     input gradient dx = dy W           : A = dy (0), B = W  (0)
//     weight gradient dW += dy^T x       : A = dy (1), B = x  (0), accumulate = 1
//   One workgroup of 4 waves (2 x 2) per 64 x 64 output tile, each wave one 32 x 32 accumulator block; k-tiles of 16
//   staged through LDS in reduction-major order ([k][m] and [k][n], so both fragments are conflict-free ds_read_b32 of 32
//   consecutive floats), register-prefetched one tile ahead.  Global loads run along the contiguous dimension of each
//   operand (16-byte loads when the leading dimension / pointer alignment allow it, scalar otherwise: the parity shapes
//   include V = 21, 30, 50).  The reduction over k is NOT split across workgroups: every output element is one fixed-order
//   sum, so the fp32 mode is deterministic without a scratch buffer.
//   The accumulator block has n on the lane (32 consecutive floats per register = 128-byte row segments), so the store
//   tail is coalesced as it stands; bias is a per-lane constant.
#include "asr_common.h"

namespace {

constexpr int FT = 64, FK = 16, FS = FT + 4;   // tile edge, k-tile, LDS row stride (floats)

__device__ __forceinline__ int acc_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// One operand tile: FT "outer" indices (m or n) x FK reduction indices -> 4 floats per thread.
// CONTIG_K: the reduction index is the contiguous one in memory (thread: outer = t / 4, k = 4 (t % 4) .. +3),
// otherwise the outer index is (thread: k = t / 16, outer = 4 (t % 16) .. +3).
template <bool CONTIG_K, bool VEC>
__device__ __forceinline__ f32x4 f32_tile_load(const float* __restrict__ p, int ld, int o0, int no, int k0, int nk, int tid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (CONTIG_K) {
        const int o = o0 + (tid >> 2), k = k0 + (tid & 3) * 4;
        if (o < no) {
            const float* q = p + (size_t)o * ld + k;
            if (VEC && k + 3 < nk) v = *(const f32x4*)q;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (k + e < nk) ? q[e] : 0.f;
            }
        }
    } else {
        const int k = k0 + (tid >> 4), o = o0 + (tid & 15) * 4;
        if (k < nk) {
            const float* q = p + (size_t)k * ld + o;
            if (VEC && o + 3 < no) v = *(const f32x4*)q;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (o + e < no) ? q[e] : 0.f;
            }
        }
    }
    return v;
}
template <bool CONTIG_K>
__device__ __forceinline__ void f32_tile_store(float* tile, f32x4 v, int tid) {   // tile[k][outer], row stride FS
    if (CONTIG_K) {
        const int o = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(k + e) * FS + o] = v[e];
    } else {
        const int k = tid >> 4, o = (tid & 15) * 4;
        *(f32x4*)(tile + k * FS + o) = v;
    }
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                       const float* __restrict__ mask, float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                       int ldc, int tiles_n, int act, int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[2][FK * FS], Bs[2][FK * FS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * FT, n0 = tn * FT;
    const int wm = w >> 1, wn = w & 1, r = lane & 31, hh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // A: outer = m; memory (M, K) has k contiguous unless transposed.  B: outer = n; memory (K, N) has n contiguous unless transposed.
    f32x4 sa = f32_tile_load<!TA, VEC>(A, lda, m0, M, 0, K, tid);
    f32x4 sb = f32_tile_load<TB, VEC>(B, ldb, n0, N, 0, K, tid);
    const int nk = (K + FK - 1) / FK;
    f32_tile_store<!TA>(As[0], sa, tid);
    f32_tile_store<TB>(Bs[0], sb, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa = f32_tile_load<!TA, VEC>(A, lda, m0, M, (kt + 1) * FK, K, tid);
            sb = f32_tile_load<TB, VEC>(B, ldb, n0, N, (kt + 1) * FK, K, tid);
        }
        const float* as = As[cur] + wm * 32 + r;
        const float* bs = Bs[cur] + wn * 32 + r;
#pragma unroll
        for (int s = 0; s < FK / 2; ++s) {      // operand of v_mfma_f32_32x32x2_f32: lane (r, hh) holds element [r][k = hh]
            const float a = as[(2 * s + hh) * FS], b = bs[(2 * s + hh) * FS];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (kt + 1 < nk) {      // the other buffer was last read in iteration kt - 1, before that iteration's barrier
            f32_tile_store<!TA>(As[cur ^ 1], sa, tid);
            f32_tile_store<TB>(Bs[cur ^ 1], sb, tid);
        }
        __syncthreads();
    }
    // accumulator block: column n = lane & 31, rows m = acc_row32(reg, lane)
    const int n = n0 + wn * 32 + r;
    if (n >= N) return;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 32 + acc_row32(i, lane);
        if (m >= M) continue;
        float x = acc[i] + bv;
        if (act == ASR_ACT_RELU) x = fmaxf(x, 0.f);
        const size_t at = (size_t)m * ldc + n;
        if (act == ASR_ACT_RELU_MASK) x = mask[at] > 0.f ? x : 0.f;
        C[at] = accumulate ? C[at] + x : x;
    }
}

}  // namespace

extern "C" int asr_gemm_f32(const float* A, const float* B, const float* bias, const float* mask, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int trans_a, int trans_b, int act, int accumulate, void* stream) {
    if (!A || !B || !C) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: bad shape M=%d N=%d K=%d", M, N, K);
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: leading dimension smaller than the row (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    if (act != ASR_ACT_NONE && act != ASR_ACT_RELU && act != ASR_ACT_RELU_MASK) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: unknown activation %d", act);
    if ((act == ASR_ACT_RELU_MASK) != (mask != nullptr)) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: ASR_ACT_RELU_MASK needs the activations in `mask` (and only it), laid out like C");
    if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 4) || (bias && (uintptr_t)bias % 4) || (mask && (uintptr_t)mask % 4)) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: misaligned pointer");
    const bool vec = (((uintptr_t)A | (uintptr_t)B) % 16) == 0 && lda % 4 == 0 && ldb % 4 == 0;
    const int tiles_n = ceil_div(N, FT);
    const long long grid = (long long)tiles_n * ceil_div(M, FT);
    if (grid > 0x7fffffffLL) ASR_FAIL(ASR_EINVAL, "asr_gemm_f32: too many tiles");
    hipStream_t st = (hipStream_t)stream;
#define F32_LAUNCH(TA_, TB_, V_) gemm_f32_kernel<TA_, TB_, V_><<<(int)grid, 256, 0, st>>>(A, B, bias, mask, C, M, N, K, lda, ldb, ldc, tiles_n, act, accumulate)
#define F32_PICK(TA_, TB_) do { if (vec) F32_LAUNCH(TA_, TB_, true); else F32_LAUNCH(TA_, TB_, false); } while (0)
    if (trans_a) { if (trans_b) F32_PICK(true, true); else F32_PICK(true, false); }
    else { if (trans_b) F32_PICK(false, true); else F32_PICK(false, false); }
#undef F32_PICK
#undef F32_LAUNCH
    ASR_CHECK_LAUNCH("asr_gemm_f32");
    return ASR_OK;
}
