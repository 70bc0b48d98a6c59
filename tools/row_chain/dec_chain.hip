// Row-chain kernel of the decoder layers: several dependent projections (+ bias, ReLU, residual + LayerNorm + pad zeroing) of the
// SAME 32 rows in one launch, the rows staying in LDS between the stages.
//
// Between two attentions every operation of a decoder layer is row-local (transformer_official.py:446-458: out-projection, residual +
// LayerNorm (attention.py:56-62), the next Q projection; position-wise FFN with its LayerNorm (module.py:68-75), the next layer's fused
// Q|K|V projection).  On B * To ~ 550 rows each of them is a 5 - 12 us kernel whose time is launch, first-tile latency and drain (round-3
// trace: 11 kernels per layer and direction); as ONE workgroup per 32 rows walking through the stages, a layer's forward pass is 4 launches
// (Q|K|V of the first layer, self-attention, chain A, cross-attention, chain B) and the only floor left is the weight stream of each
// workgroup: every workgroup reads every weight of its stages once (L2 hits after the first workgroup), straight from global memory
// into MFMA fragments - a weight element is used by exactly one wave of the workgroup, so staging it in LDS would buy nothing.  The
// weights come from FRAGMENT-ORDERED copies (asr_frag_swizzle_batched_bf16, made once per optimizer step beside the transposed copies):
// out of the matrices as stored a fragment load touches 32 rows x 32 B, one tag lookup per lane, and the first version of this kernel
// streamed 21 - 31 GB/s per CU (chain A, 1 MB of weights: 48 us; chain B: 244 us - the joint step got 0.54 ms SLOWER).
//
//   chain A:  ctx_s -> fc_s (+ x_in, LayerNorm) -> y_s -> q_c
//   chain B:  ctx_c -> fc_c (+ y_s, LayerNorm) -> y_c -> w_1 (ReLU) -> h -> w_2 (+ y_c, LayerNorm) -> y_f [-> the next layer's Q|K|V]
//
// Every tensor the per-kernel sequence leaves for the backward pass (y, xhat, rstd, h, q_c, qkv) is written exactly as before, so
// asr_decoder_layer_bwd is unchanged.  Results differ from the per-kernel path only by summation order inside the MFMA chains (the rows pass
// between stages as bf16 either way).
//
// Workgroup: 512 threads = 8 waves; a stage's output is produced 512 columns at a time, wave w owning 64 of them for all 32 rows (C^T
// blocks: n in registers, m on the lane, as everywhere in gemm.hip).  LDS: one pool, the A rows of a stage at one end, its output rows at the other,
// roles swapping from stage to stage (32 x (K+8) + 32 x (N+8) bf16 <= 129 KiB covers (512 -> 1536) and (1024 -> 512)).
#include "asr_common.h"

namespace {

constexpr int CH_ROWS = 32, CH_THREADS = 512, CH_WAVES = CH_THREADS / WAVE, CH_PAD = 8;
constexpr int CH_MAX_STAGES = 4;
constexpr int CH_POOL = CH_ROWS * (512 + CH_PAD + 1536 + CH_PAD) * 2;      // bytes: the widest pair of the decoder (d -> 3 H dk)
constexpr int CH_LDS = CH_POOL + 2 * CH_WAVES * CH_ROWS * 4;                // + two exchange arrays of the LayerNorm statistics
ASR_FULL_WAVES(CH_THREADS);

enum { CH_PLAIN = 0, CH_RELU = 1, CH_LN = 2 };

struct ChainStage {
    const bf16_t* W;       // (N, K) in FRAGMENT order (asr_frag_swizzle_batched_bf16): 32-row slabs, each k-step's 512 elements in lane order
    const float* bias;     // (N) or null
    int N, K, mode;
    bf16_t* out;           // (M, N) row stride ldo: the stage's result (after ReLU / LayerNorm and pad zeroing)
    int ldo;
    // CH_LN: z = x W^T + bias + residual; xhat = (z - mean) rstd; out = xhat gamma + beta, rows t >= lens[b] zeroed
    const bf16_t* res;     // (M, N) row stride ldr, or null = the rows an earlier stage of the same width left at the output end of the pool
    int ldr;
    const float *gamma, *beta;
    bf16_t* xhat;          // (M, N) dense
    float* rstd;           // (M)
};
struct ChainArgs {
    const bf16_t* A0;      // (M, K of stage 0), row stride lda0
    int lda0, M, To, nstage;
    const int32_t* lens;   // (B) or null
    ChainStage st[CH_MAX_STAGES];
};

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ bf16_t* region(char* pool, int side, int width) {
    return (bf16_t*)(side ? pool + CH_POOL - CH_ROWS * (width + CH_PAD) * 2 : pool);
}

// rows [m0, m0 + 32) of a global (M, width) matrix <-> an LDS region with row stride width + CH_PAD, 16 B per thread and trip
__device__ __forceinline__ void rows_in(bf16_t* dst, const bf16_t* src, int ld, int width, int m0, int M, int tid) {
    const int cpr = width >> 3;
    for (int idx = tid; idx < CH_ROWS * cpr; idx += CH_THREADS) {
        const int row = idx / cpr, ch = idx - row * cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m0 + row < M) v = *(const u32x4*)(src + (size_t)(m0 + row) * ld + ch * 8);
        *(u32x4*)(dst + row * (width + CH_PAD) + ch * 8) = v;
    }
}
__device__ __forceinline__ void rows_out(bf16_t* dst, int ld, const bf16_t* src, int width, int m0, int M, int tid) {
    const int cpr = width >> 3;
    for (int idx = tid; idx < CH_ROWS * cpr; idx += CH_THREADS) {
        const int row = idx / cpr, ch = idx - row * cpr;
        if (m0 + row < M) *(u32x4*)(dst + (size_t)(m0 + row) * ld + ch * 8) = *(const u32x4*)(src + row * (width + CH_PAD) + ch * 8);
    }
}

// The MFMA chain of one 512-column block of a stage: wave w owns columns [n0, n0 + 64) = CH_NB blocks of 32.  The weight fragments come
// straight from global memory (fragment-ordered copy: 1 KiB of consecutive bytes per load instruction), CH_G groups of CH_KU k-steps in
// flight per wave - a group is re-issued as soon as its MFMAs are done: 32 KiB per wave, 256 KiB per CU.  The per-CU weight stream is bound
// by latency x bytes in flight (every workgroup streams every weight of the chain).
constexpr int CH_NB = 2, CH_KU = 4, CH_G = 4, CH_COLS = CH_WAVES * CH_NB * 32;      // 512 columns per pass of the workgroup
__device__ __forceinline__ void chain_mma(f32x16 (&acc)[CH_NB], const bf16_t* Wl, const bf16_t* Al, int K) {
#pragma unroll
    for (int nb = 0; nb < CH_NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
    const size_t blk = (size_t)32 * K;
    bf16x8 wf[CH_G][CH_KU][CH_NB];
    auto fetch = [&](bf16x8(&f)[CH_KU][CH_NB], int k0) {
#pragma unroll
        for (int ks = 0; ks < CH_KU; ++ks)
#pragma unroll
            for (int nb = 0; nb < CH_NB; ++nb) f[ks][nb] = *(const bf16x8*)(Wl + nb * blk + (size_t)((k0 >> 4) + ks) * 512);
    };
    auto mma = [&](bf16x8(&f)[CH_KU][CH_NB], int k0) {
#pragma unroll
        for (int ks = 0; ks < CH_KU; ++ks) {
            const bf16x8 af = *(const bf16x8*)(Al + k0 + 16 * ks);
#pragma unroll
            for (int nb = 0; nb < CH_NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][nb], af, acc[nb], 0, 0, 0);
        }
    };
    constexpr int KG = 16 * CH_KU;      // k per group; K is a multiple of CH_G * KG = 256 (host check)
#pragma unroll
    for (int g = 0; g < CH_G; ++g) fetch(wf[g], g * KG);
    for (int k0 = 0; k0 < K; k0 += CH_G * KG) {
#pragma unroll
        for (int g = 0; g < CH_G; ++g) {
            mma(wf[g], k0 + g * KG);
            if (k0 + (CH_G + g) * KG < K) fetch(wf[g], k0 + (CH_G + g) * KG);
        }
    }
}

// One stage: N / 512 passes of chain_mma, each followed by its store tail into the output end of the pool; a LayerNorm stage has N = 512
// (one pass: the row statistics are exchanged between the 8 waves through `red`).
__device__ __forceinline__ void chain_stage(const ChainStage& S, char* pool, float* red, int side, int m0, int M, int To, const int32_t* lens, int tid) {
    const int lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int N = S.N, K = S.K, ldA = K + CH_PAD, ldO = N + CH_PAD;
    const bf16_t* Ab = region(pool, side, K);
    bf16_t* Ob = region(pool, side ^ 1, N);
    const bf16_t* Al = Ab + r * ldA + 8 * hh;
    const int row = m0 + r;
    f32x16 acc[CH_NB];
    if (S.mode != CH_LN) {
        for (int c0 = 0; c0 < N; c0 += CH_COLS) {
            const int n0w = c0 + w * (CH_NB * 32);
            chain_mma(acc, S.W + (size_t)n0w * K + lane * 8, Al, K);
#pragma unroll
            for (int nb = 0; nb < CH_NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0w + nb * 32 + 8 * g + 4 * hh;
                    f32x4 v = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                    if (S.bias) v += *(const f32x4*)(S.bias + n);
                    if (S.mode == CH_RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    store4<bf16_t>(Ob + r * ldO + n, v);      // the output end of the pool: nobody reads it during this stage
                }
        }
        __syncthreads();
        rows_out(S.out, S.ldo, Ob, N, m0, M, tid);
    } else {
        const int n0w = w * (CH_NB * 32);
        chain_mma(acc, S.W + (size_t)n0w * K + lane * 8, Al, K);
        __syncthreads();      // every wave is done with the A rows (xhat is staged there); a preloaded residual is in place
        float* red2 = red + CH_WAVES * CH_ROWS;
        float s = 0.f;
#pragma unroll
        for (int nb = 0; nb < CH_NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0w + nb * 32 + 8 * g + 4 * hh;
                f32x4 v = load4<bf16_t>(Ob + r * ldO + n);      // the residual rows
                if (S.bias) v += *(const f32x4*)(S.bias + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[nb][4 * g + i] += v[i];
                    s += acc[nb][4 * g + i];
                }
            }
        s += __shfl_xor(s, 32);
        if (hh == 0) red[w * CH_ROWS + r] = s;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int i = 0; i < CH_WAVES; ++i) mean += red[i * CH_ROWS + r];
        const float inv_n = 1.f / (float)N;
        mean *= inv_n;
        float q = 0.f;
#pragma unroll
        for (int nb = 0; nb < CH_NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[nb][e] -= mean;
                q += acc[nb][e] * acc[nb][e];
            }
        q += __shfl_xor(q, 32);
        if (hh == 0) red2[w * CH_ROWS + r] = q;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < CH_WAVES; ++i) var += red2[i * CH_ROWS + r];
        const float rstd = rsqrtf(var * inv_n + 1e-5f);
        bool keep = true;
        if (lens && row < M) {
            const int b = row / To, t = row - b * To;
            keep = t < lens[b];
        }
        bf16_t* Xb = region(pool, side, N);      // xhat rows: staged where the A rows were
#pragma unroll
        for (int nb = 0; nb < CH_NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0w + nb * 32 + 8 * g + 4 * hh;
                const f32x4 ga = *(const f32x4*)(S.gamma + n), be = *(const f32x4*)(S.beta + n);
                f32x4 xh, y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[i] = acc[nb][4 * g + i] * rstd;
                    y[i] = keep ? xh[i] * ga[i] + be[i] : 0.f;
                }
                store4<bf16_t>(Xb + r * ldO + n, xh);
                store4<bf16_t>(Ob + r * ldO + n, y);
            }
        if (w == 0 && hh == 0 && row < M) S.rstd[row] = rstd;
        __syncthreads();
        rows_out(S.out, S.ldo, Ob, N, m0, M, tid);
        rows_out(S.xhat, N, Xb, N, m0, M, tid);
    }
    __syncthreads();      // the copies out of the pool are done before the next stage writes into it
}

__global__ __launch_bounds__(CH_THREADS) void dec_chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char pool[];
    float* red = (float*)(pool + CH_POOL);
    const int tid = threadIdx.x, m0 = blockIdx.x * CH_ROWS;
    rows_in(region(pool, 0, a.st[0].K), a.A0, a.lda0, a.st[0].K, m0, a.M, tid);
    int side = 0;
    for (int s = 0; s < a.nstage; ++s) {
        const ChainStage& S = a.st[s];
        if (S.mode == CH_LN && S.res) rows_in(region(pool, side ^ 1, S.N), S.res, S.ldr, S.N, m0, a.M, tid);
        __syncthreads();      // A rows (first stage) in place
        chain_stage(S, pool, red, side, m0, a.M, a.To, a.lens, tid);
        side ^= 1;
    }
}

bool stage_ok(const ChainStage& S) {
    if (S.N <= 0 || S.N % CH_COLS || (S.mode == CH_LN && S.N != CH_COLS)) return false;
    if (S.K % 256 || S.K <= 0 || S.ldo % 8 || S.ldo < S.N) return false;      // K: whole trips of every (KU, G) variant
    if (CH_ROWS * (S.K + CH_PAD + S.N + CH_PAD) * 2 > CH_POOL) return false;
    if (S.mode == CH_LN && CH_ROWS * 2 * (S.N + CH_PAD) * 2 > CH_POOL) return false;
    return true;
}

}      // namespace

// 1: the row-chain kernels cover a decoder layer of these widths (model width 512 = one pass of the workgroup, the other widths multiples
// of 512, reductions multiples of 256, each stage's rows within the LDS pool); the caller then sets asr_dec_layer_plan.chain.
extern "C" int asr_decoder_chain_supported(int d, int hd, int ff) {
    ChainStage t[5] = {};
    const int nk[5][2] = {{d, hd}, {hd, d}, {ff, d}, {d, ff}, {3 * hd, d}};
    for (int i = 0; i < 5; ++i) {
        t[i].N = nk[i][0]; t[i].K = nk[i][1]; t[i].ldo = nk[i][0];
        t[i].mode = (i == 0 || i == 3) ? CH_LN : CH_PLAIN;
        if (!stage_ok(t[i])) return 0;
    }
    return 1;
}

static int launch_chain(const ChainArgs& a, const char* who, hipStream_t st) {
    for (int s = 0; s < a.nstage; ++s) {
        const ChainStage& S = a.st[s];
        if (!stage_ok(S)) ASR_FAIL(ASR_EINVAL, "%s: stage %d (N=%d K=%d) is outside the row-chain kernel's shapes (asr_decoder_chain_supported)", who, s, S.N, S.K);
        if (((uintptr_t)S.W | (uintptr_t)S.out | (uintptr_t)S.bias | (uintptr_t)S.gamma | (uintptr_t)S.beta | (uintptr_t)S.res | (uintptr_t)S.xhat) % 16)
            ASR_FAIL(ASR_EINVAL, "%s: stage %d has a pointer off a 16-byte boundary", who, s);
        if (!S.W || !S.out || (S.mode == CH_LN && (!S.gamma || !S.beta || !S.xhat || !S.rstd))) ASR_FAIL(ASR_EINVAL, "%s: null pointer in stage %d", who, s);
        if (s && S.K != a.st[s - 1].N) ASR_FAIL(ASR_EINVAL, "%s: stage %d reduces over %d columns, stage %d produced %d", who, s, S.K, s - 1, a.st[s - 1].N);
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)dec_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
        attr = true;
    }
    asr_launch_armed(dec_chain_kernel, dim3(ceil_div(a.M, CH_ROWS)), dim3(CH_THREADS), CH_LDS, st, a);
    ASR_CHECK_LAUNCH(who);
    return ASR_OK;
}

static ChainStage plain(const void* W, const float* bias, int N, int K, void* out, int mode) {
    ChainStage S = {};
    S.W = (const bf16_t*)W; S.bias = bias; S.N = N; S.K = K; S.mode = mode; S.out = (bf16_t*)out; S.ldo = N;
    return S;
}
static ChainStage with_ln(const void* W, const float* bias, int N, int K, void* y, const void* res, const float* g, const float* be, void* xhat, float* rstd) {
    ChainStage S = plain(W, bias, N, K, y, CH_LN);
    S.res = (const bf16_t*)res; S.ldr = N; S.gamma = g; S.beta = be; S.xhat = (bf16_t*)xhat; S.rstd = rstd;
    return S;
}

// chain A of a layer: self-attention context -> out-projection, residual + LayerNorm -> Q projection of the encoder-decoder attention
int asr_dec_chain_a(const asr_dec_layer_plan* p, void* stream) {
    const int d = p->d, hd = p->H * p->dk;
    ChainArgs a = {};
    a.A0 = (const bf16_t*)p->ctx_s; a.lda0 = hd; a.M = p->B * p->To; a.To = p->To; a.lens = p->dec_len; a.nstage = 2;
    a.st[0] = with_ln(p->wf_fc_s, p->b_fc_s, d, hd, p->y_s, p->x_in, p->g_s, p->be_s, p->a_s, p->rstd_s);
    a.st[1] = plain(p->wf_q_c, p->b_q_c, hd, d, p->q_c, CH_PLAIN);
    return launch_chain(a, "asr_decoder_layer_fwd (chain A)", (hipStream_t)stream);
}

// chain B: cross-attention context -> out-projection, residual + LayerNorm -> w_1, ReLU -> w_2, residual + LayerNorm [-> Q|K|V of the next layer]
int asr_dec_chain_b(const asr_dec_layer_plan* p, void* stream) {
    const int d = p->d, hd = p->H * p->dk, ff = p->ff;
    ChainArgs a = {};
    a.A0 = (const bf16_t*)p->ctx_c; a.lda0 = hd; a.M = p->B * p->To; a.To = p->To; a.lens = p->dec_len; a.nstage = 3;
    a.st[0] = with_ln(p->wf_fc_c, p->b_fc_c, d, hd, p->y_c, p->y_s, p->g_c, p->be_c, p->a_c, p->rstd_c);
    a.st[1] = plain(p->wf_1, p->b_1, ff, d, p->h, CH_RELU);
    a.st[2] = with_ln(p->wf_2, p->b_2, d, ff, p->y_f, nullptr, p->g_f, p->be_f, p->o, p->rstd_f);      // residual y_c: still in the pool
    if (p->next) {
        const asr_dec_layer_plan* q = p->next;
        a.st[3] = plain(q->wf_qkv_s, q->b_qkv_s, 3 * hd, d, q->qkv_s, CH_PLAIN);
        a.nstage = 4;
    }
    return launch_chain(a, "asr_decoder_layer_fwd (chain B)", (hipStream_t)stream);
}
