// Shared device/host helpers for libasr_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "../../include/asr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ---- error reporting (host) -----------------------------------------------------------------
void asr_set_error(const char* fmt, ...);
int asr_deterministic(void);   // 1: fixed-order reductions everywhere (asr_set_deterministic / ASR_DETERMINISTIC=1)
// tuning options (asr_set_option, misc.hip): every value of every option gives correct results
enum { ASR_OPT_CU_LIMIT = 0, ASR_OPT_TN_MULTI = 1, ASR_OPT_SDPA_PAIR = 2, ASR_OPT_COUNT = 3 };
int asr_option(int key);
int asr_option_set(int key, int value);      // returns the previous value (library-internal callers: decoder_exec.hip)
#define ASR_FAIL(code, ...)        \
    do {                           \
        asr_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)
#define ASR_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- element load/store in fp32 -------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements <-> f32x4 (16-B loads for f32, 8-B for bf16)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
// Non-temporal ("nt") store for LARGE outputs that the next kernel will not find in L2 anyway: a plain store leaves
// the lines dirty in the XCD's L2, and the write-back of everything still dirty is serialised at the END of the kernel
// (the eight L2s are not coherent with each other, so a kernel's results must reach memory before the next kernel
// starts).  Used by the NT GEMM store tail (fc 17.0 -> 15.3 us; step -1.4 % on one box, -0.1 % on another, never
// worse).  NOT used for LayerNorm / CTC rows (faster alone, 12.6 -> 11.4 us, but the step gets 0.7 % slower on
// some boxes: their consumer then misses L2; the CTC softmax rows, 135 MB: 110 -> 104 us alone, step unchanged) and
// never for 8-byte pieces (attention outputs: 36 -> 49 us).
template <typename V> __device__ __forceinline__ void stream_store(V v, V* p) { __builtin_nontemporal_store(v, p); }
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4*)p = r;
}
// 8 consecutive elements (32-B f32 / 16-B bf16)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&r)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&r)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&r)[8]) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&r)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&r)[8]) {
    f32x4 a = {r[0], r[1], r[2], r[3]}, b = {r[4], r[5], r[6], r[7]};
    *(f32x4*)p = a;
    *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&r)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)r[i];
    *(bf16x8*)p = v;
}

// ---- wave-level reductions (64 lanes, DPP/shuffle, no LDS) ----------------------------------
// Wave-wide reductions on the vector ALU's lane crossbar (DPP), result in every lane: four butterfly steps inside each row of
// 16 lanes (quad swaps, half-row and row mirrors), two row broadcasts, one v_readlane.  As six __shfl_xor steps each of them was a
// ds_bpermute_b32 through the LDS crossbar followed by s_waitcnt lgkmcnt(0): ~100 cycles of exposed latency per step, and the
// one-wave-per-row kernels (LayerNorm) run two to four such reductions per row.
// PRECONDITION (wave_sum, wave_max and block_sum / block_max on top of them): the WHOLE wave executes the call - all 64 lanes active, wave-uniform
// control flow around it - because the result is read from lane 63 (v_readlane) and the row broadcasts feed from lanes 15 / 31.  The __shfl_xor form
// they replaced returned a correct value on every active lane of a partial wave; these do not.  Every launch in this library that reaches them uses a
// block size that is a multiple of 64 (checked at compile time where it is a constant: ASR_FULL_WAVES) and calls them outside divergent branches.
#define ASR_FULL_WAVES(threads) static_assert((threads) % 64 == 0, "wave_sum / wave_max need full 64-lane waves: block sizes must be multiples of 64")
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float wave_dpp(float v) {      // lanes outside ROW_MASK, and lanes without a source, get 0 / keep v's identity
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += wave_dpp<0xB1>(v);             // quad_perm [1,0,3,2]
    v += wave_dpp<0x4E>(v);             // quad_perm [2,3,0,1]
    v += wave_dpp<0x141>(v);            // row_half_mirror
    v += wave_dpp<0x140>(v);            // row_mirror: every lane of a row holds the row's sum
    v += wave_dpp<0x142, 0xA>(v);       // row_bcast15 into rows 1 and 3
    v += wave_dpp<0x143, 0xC>(v);       // row_bcast31 into rows 2 and 3: row 3 holds the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    // old = v for the masked steps: a lane outside the row mask must keep its own value under max (0 would be wrong for negatives)
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xf, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// block-wide sum for blocks of up to 1024 threads (a multiple of 64: see the precondition above); `red` is >= 16 floats of LDS.  All threads
// get the result.  Contains two barriers.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

__device__ __forceinline__ float log_add(float a, float b) {
    // log(exp(a)+exp(b)) with -inf handling
    float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(__expf(a - m) + __expf(b - m));
}

// Second stage of every column reduction: out_k[c] (+)= sum_p part[p*pstride + k*d + c] for the
// up-to-three vectors k packed side by side in a partial row.  32 columns x 32 partial groups per
// 1024-thread workgroup: each thread sums P/32 partials, so the pass is a few microseconds.
static __global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, int P, size_t pstride, int ncols, int d,
                                                                      float* __restrict__ out0, float* __restrict__ out1,
                                                                      float* __restrict__ out2, int accumulate) {
    __shared__ float red[32][33];
    const int cg = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cg;
    // gridDim.y > 1 (accumulating calls only): the partial rows are split between gridDim.y workgroups, each
    // adds its share with one atomic per column - one batch of loads per thread instead of four in series
    const int per = (P + gridDim.y - 1) / gridDim.y, pbeg = blockIdx.y * per, pend = min(P, pbeg + per);
    float s = 0.f;
    if (col < ncols) {
        int p = pbeg + pg;
        for (; p + 7 * 32 < pend; p += 8 * 32) {   // 8 independent loads in flight (the pass is latency-bound)
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
        float* out = k == 0 ? out0 : (k == 1 ? out1 : out2);
        if (gridDim.y > 1) atomicAdd(out + c, t);
        else out[c] = accumulate ? out[c] + t : t;
    }
}

// ---- dropout: counter-based keep mask ---------------------------------------------------------
// One 32-bit hash per PAIR of adjacent elements (low / high 16 bits), keyed by (pair index, seed):
// forward and backward regenerate identical masks, nothing is stored.  keep <=> 16 bits >= p*65536.
// Round 3: two 24-bit multiplies (v_mul_u32_u24, full rate) instead of three 32-bit ones (v_mul_lo_u32, quarter rate): 9 full-rate
// vector instructions per pair instead of ~19 issue slots; each xor-shift folds the bits the next multiply would drop (it reads the low
// 24) into the ones it keeps.  Checked on 4 M consecutive indices (numpy emulation): drop rate 0.10006 / 0.09997 at p = 0.1 for the
// two halves, |correlation| of the keep bits <= 2.3e-3 at lags 1, 250, 256, 1024, between the halves and between seeds s, s + 1;
// chi-square of the top byte 229 / 199 on 255 degrees of freedom.
__device__ __forceinline__ uint32_t drop_hash(uint32_t pair, uint32_t seed) {
    uint32_t h = pair ^ seed;
    h ^= h >> 15; h = __umul24(h, 0x2C1B3Du);
    h ^= h >> 13; h = __umul24(h, 0x297A2Du);
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ bool drop_keep(uint32_t h, int half, uint32_t thr16) { return ((h >> (16 * half)) & 0xffffu) >= thr16; }
__device__ __forceinline__ bool drop_keep_at(uint32_t elem, uint32_t seed, uint32_t thr16) {
    return drop_keep(drop_hash(elem >> 1, seed), elem & 1, thr16);
}
static inline uint32_t drop_thr16(float p) { return p <= 0.f ? 0u : (uint32_t)(p * 65536.f + 0.5f); }

// XCD-aware workgroup order (guide T1).  Workgroups are dealt round-robin over the 8 XCDs (each
// with a private L2), so consecutive linear ids land on different L2s.  This maps the hardware id
// to a virtual id such that each XCD owns a contiguous run of virtual ids: workgroups that share
// operand panels are given consecutive virtual ids and then hit one L2.  Bijective for any size.
__device__ __forceinline__ int xcd_virtual_id(int id, int nwg) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- armed launches (asr_stream_arm): the completion event of ONE kernel launch, without a marker packet in its queue ------------
// hipEventRecord puts a barrier packet behind the kernel it follows: the next kernel of that queue then starts ~3.5 us later than it
// would (step timeline, round 3: every weight-gradient hand-over to the side stream showed as a 6.3-us gap on the main stream instead
// of 2.8).  hipExtLaunchKernelGGL binds an event to the dispatch packet's own completion signal instead.  An entry point that supports
// arming launches its LAST kernel through asr_launch_armed(): when the caller armed this stream (asr_stream_arm) the kernel carries the
// event and the armed stream waits for it; otherwise it is an ordinary launch.
bool asr_arm_take(hipStream_t st, hipStream_t* to, hipEvent_t* ev);      // misc.hip: true once, if `st` is armed
template <typename... KArgs, typename... Args>
static inline void asr_launch_armed(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipStream_t to;
    hipEvent_t ev;
    if (asr_arm_take(st, &to, &ev)) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, nullptr, ev, 0, static_cast<KArgs>(args)...);
        (void)hipStreamWaitEvent(to, ev, 0);
    } else {
        kernel<<<grid, block, lds, st>>>(static_cast<KArgs>(args)...);
    }
}
