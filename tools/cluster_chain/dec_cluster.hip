// Cluster kernels of the decoder layers: the row-local runs between two attentions - out-projection + residual + LayerNorm + the next
// projection(s) - as ONE launch each, every projection's N split across the 32 workgroups of a cluster, rows exchanged INSIDE the kernel.
//
// Reference: DecoderLayer.forward (transformer_official.py:446-458): MultiHeadAttention's fc + residual + LayerNorm + pad zeroing
// (attention.py:56-62), the next attention's Q projection; PositionwiseFeedForwardUseConv (module.py:68-75); the next layer's Q|K|V.
//   chain A:  ctx_s -> fc_s (+ x_in, LayerNorm) -> y_s -> q_c
//   chain B:  ctx_c -> fc_c (+ y_s, LayerNorm) -> y_c -> w_1 (ReLU) -> h -> w_2 (+ y_c, LayerNorm) -> y_f [-> the next layer's Q|K|V]
// On B * To ~ 550 rows each of these operations is a 5 - 12 us kernel, 9 of a layer's 11 forward launches; their cost is launch + drain.
//
// Two measurements shaped this kernel (tools/row_chain, tools/xcd_cluster; DESIGN.md section 4 "Tried"):
//  * a CU pulls ~40 GB/s out of L2: a workgroup that keeps rows in LDS and streams EVERY weight of its chain is slower than the separate
//    kernels (row-chain kernel: 25 / 123 us).  So N is split: workgroup j of a cluster owns N/32 output columns of every stage and reads
//    only those rows of W; what it needs from the other workgroups are the ROWS of the previous stage (68 x 512 bf16 = 70 KB per cluster).
//  * a barrier between 32 workgroups made of RELAXED agent-scope atomics costs 1.3 us; with release / acquire it costs 5.8 us - the L2
//    write-back and invalidate of the fences are what a kernel boundary pays too.  So exchanged data goes by sc1 stores (write-through to the
//    coherence point; complete when s_waitcnt vmcnt(0) returns) and sc1 loads (never served from this CU's L1 or a stale L2 line), the barrier
//    by relaxed atomics, and there is no fence anywhere.  Tensors that only LATER kernels read (xhat, rstd, the last stage) use plain stores.
//
// Geometry: 256 workgroups = 8 clusters x 32; cluster c = blockIdx % 8 (in practice the workgroups of one XCD - nothing depends on it),
// rows [c * rpc, (c + 1) * rpc), rpc = ceil(M / 8) <= 128; one wave per 16 rows (MFMA 16x16x32, C^T form: n in registers, m on the lane).
// LayerNorm: every workgroup leaves (sum, M2) of its 16 columns per row; after a barrier each merges the 32 partials (Chan's parallel
// variance - no E[x^2] - mean^2 cancellation), normalises its slice and publishes it; a second barrier makes the rows visible.
// All 256 workgroups must be resident at the same time (grid <= CUs, 1 workgroup per CU fits beside anything else that is running, or
// waits for it to finish - nothing running waits for this kernel); every spin is bounded and ends in an abort flag the host checks.
#include "asr_common.h"

namespace {

static_assert(true, "");
constexpr int CL_N = 8, CL_W = 8, CL_LN_NB = 4, CL_MAX_RPC = 128, CL_MAX_STAGES = 4, CL_KC = 512, CL_KS = CL_KC / 32, CL_SPIN_LIMIT = 1 << 20;
enum { CL_PLAIN = 0, CL_RELU = 1, CL_LN = 2 };

struct ClStage {
    const bf16_t* W;       // (N, K) as stored
    const float* bias;     // (N)
    int N, K, mode;
    int exchange;          // 1: a later stage of this launch reads `out` (sc1 stores + a barrier), 0: only later kernels do
    bf16_t* out;           // (M, N)
    // CL_LN (N == 512): z = x W^T + bias + residual
    const bf16_t* res;     // (M, N), or null = the previous LayerNorm stage's output (this lane still holds its slice)
    const float *gamma, *beta;
    bf16_t* xhat;          // (M, N)
    float* rstd;           // (M)
};
struct ClArgs {
    const bf16_t* A0;      // (M, K of stage 0)
    int M, rpc, To, nstage;
    const int32_t* lens;   // (B)
    ClStage st[CL_MAX_STAGES];
    f32x2* part;           // [CL_N][CL_MAX_RPC][CL_W]
    unsigned *ctr, *done;  // [CL_N][32] each (one 128-B line per cluster)
    unsigned* abort_flag;  // host-mapped
};

// The waits are part of the asm statements: outside them the compiler does not know a result is still in flight (it re-used a destination
// register as an address and the late-landing load overwrote it: tools/xcd_cluster/cluster_chain_probe.hip, first version, faulted on address 0).
__device__ __forceinline__ void load16x16_sc1(bf16x8 (&v)[16], const bf16_t* p) {      // 16 B at p + 64 B * i, i < 16; also drains every earlier load
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:64 sc1\n\tglobal_load_dwordx4 %2, %16, off offset:128 sc1\n\t"
        "global_load_dwordx4 %3, %16, off offset:192 sc1\n\tglobal_load_dwordx4 %4, %16, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %16, off offset:320 sc1\n\t"
        "global_load_dwordx4 %6, %16, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %16, off offset:448 sc1\n\tglobal_load_dwordx4 %8, %16, off offset:512 sc1\n\t"
        "global_load_dwordx4 %9, %16, off offset:576 sc1\n\tglobal_load_dwordx4 %10, %16, off offset:640 sc1\n\tglobal_load_dwordx4 %11, %16, off offset:704 sc1\n\t"
        "global_load_dwordx4 %12, %16, off offset:768 sc1\n\tglobal_load_dwordx4 %13, %16, off offset:832 sc1\n\tglobal_load_dwordx4 %14, %16, off offset:896 sc1\n\t"
        "global_load_dwordx4 %15, %16, off offset:960 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]),
          "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void load16fx4_sc1(f32x4 (&v)[4], const f32x2* p) {      // 16 B at p + 16 B * i, i < 4: the CL_W = 8 partials of a row
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(p)
                 : "memory");
}
__device__ __forceinline__ void store8_sc1(bf16_t* p, bf16x4 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store8f_sc1(f32x2* p, f32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// every thread: its own stores have completed; thread 0: arrive, then wait for `target` arrivals.  false = gave up (abort flag set)
__device__ __forceinline__ bool cluster_barrier(unsigned* ctr, unsigned target, unsigned* abort_flag, unsigned* s_dead) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > CL_SPIN_LIMIT) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *s_dead = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *s_dead == 0;
}

// The MFMA chain of one stage: the workgroup's (16 nbk = 64 NG)-column slice of W passes through LDS in tiles of 64 columns x 512 k (every
// wave needs every weight of the slice: read by each wave straight from global memory the slice went through the L1 once per wave, 6 x 64 KB
// per stage at the ~40 GB/s a CU gets - the first 8-wide version ran chain B in 150 us).  All 512 threads copy a tile (8 x 16 B each, coalesced
// rows), the next tile's loads are in flight while the waves that own rows multiply the current one.  The rows (A) of a 512-k chunk stay in
// registers: plain loads when an earlier kernel wrote them, sc1 loads when this launch did.
constexpr int CL_TN = 64, CL_TLD = CL_KC + 8, CL_TILE = CL_TN * CL_TLD;      // elements; two tiles = 130 KiB of LDS
template <int NG>
__device__ __forceinline__ void cluster_gemm(f32x4 (&acc)[12], bf16_t* lds, const bf16_t* arow, bool exchanged, bool has_rows, const bf16_t* wslice, int K) {
    const int tid = threadIdx.x, lane = tid & 63, mi = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nb = 0; nb < 4 * NG; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r0 = tid >> 6, ch = tid & 63;      // this thread copies rows r0 + 8 q (q < 8), 16-B chunk ch of every tile
    u32x4 st[8];
    auto tile_load = [&](int kc, int ng) {
        const bf16_t* src = wslice + (size_t)(CL_TN * ng + r0) * K + kc + 8 * ch;
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] = *(const u32x4*)(src + (size_t)8 * q * K);
    };
    auto tile_store = [&](bf16_t* buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) *(u32x4*)(buf + (r0 + 8 * q) * CL_TLD + 8 * ch) = st[q];
    };
    tile_load(0, 0);
    int t = 0;
    for (int kc = 0; kc < K; kc += CL_KC) {
        bf16x8 af[CL_KS];
        if (has_rows) {      // wave-uniform
            if (exchanged) {
                load16x16_sc1(af, arow + kc);
            } else {
#pragma unroll
                for (int ks = 0; ks < CL_KS; ++ks) af[ks] = *(const bf16x8*)(arow + kc + 32 * ks);
            }
        }
#pragma unroll
        for (int ng = 0; ng < NG; ++ng, ++t) {
            bf16_t* buf = lds + (t & 1) * CL_TILE;
            tile_store(buf);
            __syncthreads();      // tile t is in LDS; the other buffer is free (tile t - 1 was consumed before the previous barrier's successor)
            const bool more = ng + 1 < NG || kc + CL_KC < K;
            if (more) tile_load(ng + 1 < NG ? kc : kc + CL_KC, ng + 1 < NG ? ng + 1 : 0);
            if (has_rows) {
                const bf16_t* wl = buf + mi * CL_TLD + 8 * kq;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int ks = 0; ks < CL_KS; ++ks) {
                        const bf16x8 wf = *(const bf16x8*)(wl + 16 * nb * CL_TLD + 32 * ks);
                        acc[4 * ng + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[ks], acc[4 * ng + nb], 0, 0, 0);
                    }
            }
        }
    }
    __syncthreads();      // every wave is done with the last tile before the next stage's first tile_store
}

__global__ __launch_bounds__(512) void dec_cluster_kernel(const ClArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cl_smem[];
    bf16_t* lds = (bf16_t*)cl_smem;
    __shared__ unsigned s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    const int c = blockIdx.x & (CL_N - 1), j = blockIdx.x >> 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, mi = lane & 15, kq = lane >> 4;
    const int row_end = min(a.M, (c + 1) * a.rpc), lrow = 16 * w + mi, row = c * a.rpc + lrow;
    const bool valid = lrow < a.rpc && row < row_end;
    const size_t rowc = valid ? (size_t)row : 0;      // lanes without a row read row 0; their results are not stored
    unsigned* ctr = a.ctr + 32 * c;
    f32x2* part = a.part + ((size_t)c * CL_MAX_RPC + (lrow < CL_MAX_RPC ? lrow : 0)) * CL_W;
    bool keep = true;
    if (a.lens && valid) {
        const int b = row / a.To, t = row - b * a.To;
        keep = t < a.lens[b];
    }
    __syncthreads();
    unsigned bars = 0;
    bf16x4 ykeep[CL_LN_NB] = {};         // this lane's slice of the last LayerNorm output (residual of the next one)
    const bf16_t* A = a.A0;              // rows of the current stage's operand
    bool a_exchanged = false;            // written during this launch (sc1 loads) or by an earlier kernel (plain loads)
    for (int s = 0; s < a.nstage; ++s) {
        const ClStage& S = a.st[s];
        const int N = S.N, K = S.K, nbk = N / (16 * CL_W);      // 16-column blocks of this workgroup: columns [j * 16 nbk, (j + 1) * 16 nbk)
        f32x4 acc[12];
        {
            const bf16_t* arow = A + rowc * K + 8 * kq;
            const bf16_t* wslice = S.W + (size_t)(j * 16 * nbk) * K;
            const bool has_rows = 16 * w < a.rpc;
            if (nbk == 4) cluster_gemm<1>(acc, lds, arow, a_exchanged, has_rows, wslice, K);
            else if (nbk == 8) cluster_gemm<2>(acc, lds, arow, a_exchanged, has_rows, wslice, K);
            else cluster_gemm<3>(acc, lds, arow, a_exchanged, has_rows, wslice, K);
        }
        // D[n][m]: this lane holds row m = mi of its wave's 16 rows and columns 4 kq .. 4 kq + 3 of each 16-column block
        if (S.mode != CL_LN) {
#pragma unroll
            for (int nb = 0; nb < 12; ++nb) {
                if (nb < nbk) {
                    const int n = (j * nbk + nb) * 16 + 4 * kq;
                    const f32x4 b = *(const f32x4*)(S.bias + n);
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[nb][i] + b[i];
                        if (S.mode == CL_RELU) v = fmaxf(v, 0.f);
                        o[i] = (bf16_t)v;
                    }
                    if (valid) {
                        if (S.exchange) store8_sc1(S.out + (size_t)row * N + n, o);
                        else *(bf16x4*)(S.out + (size_t)row * N + n) = o;
                    }
                }
            }
        } else {
            // N == 16 CL_LN_NB CL_W: this workgroup's CL_LN_NB blocks; per row (sum, M2) over its 16 CL_LN_NB columns
            float z[CL_LN_NB][4];
            float sm = 0.f;
#pragma unroll
            for (int nb = 0; nb < CL_LN_NB; ++nb) {
                const int n = (j * CL_LN_NB + nb) * 16 + 4 * kq;
                const f32x4 b = *(const f32x4*)(S.bias + n);
                bf16x4 r = ykeep[nb];
                if (S.res) r = *(const bf16x4*)(S.res + rowc * N + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    z[nb][i] = acc[nb][i] + b[i] + (float)r[i];
                    sm += z[nb][i];
                }
            }
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            constexpr float inv_cnt = 1.f / (float)(16 * CL_LN_NB);
            const float lmean = sm * inv_cnt;
            float m2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < CL_LN_NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) m2 += (z[nb][i] - lmean) * (z[nb][i] - lmean);
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            if (kq == 0 && lrow < a.rpc) store8f_sc1(part + j, f32x2{sm, m2});
            if (!cluster_barrier(ctr, CL_W * ++bars, a.abort_flag, &s_dead)) return;
            // merge the CL_W partials of this row: mean = sum / N, M2 = sum M2_j + count sum (mean_j - mean)^2
            float mean, rstd;
            {
                f32x4 p[CL_W / 2];
                load16fx4_sc1(p, part);
                float ts = 0.f;
#pragma unroll
                for (int i = 0; i < CL_W / 2; ++i) ts += p[i][0] + p[i][2];
                mean = ts * (1.f / (float)(16 * CL_LN_NB * CL_W));
                float t2 = 0.f;
#pragma unroll
                for (int i = 0; i < CL_W / 2; ++i) {
                    const float d0 = p[i][0] * inv_cnt - mean, d1 = p[i][2] * inv_cnt - mean;
                    t2 += p[i][1] + p[i][3] + (float)(16 * CL_LN_NB) * (d0 * d0 + d1 * d1);
                }
                rstd = rsqrtf(t2 * (1.f / (float)(16 * CL_LN_NB * CL_W)) + 1e-5f);
            }
#pragma unroll
            for (int nb = 0; nb < CL_LN_NB; ++nb) {
                const int n = (j * CL_LN_NB + nb) * 16 + 4 * kq;
                const f32x4 g = *(const f32x4*)(S.gamma + n), be = *(const f32x4*)(S.beta + n);
                bf16x4 xh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = (z[nb][i] - mean) * rstd;
                    xh[i] = (bf16_t)x;
                    ykeep[nb][i] = (bf16_t)(keep ? x * g[i] + be[i] : 0.f);
                }
                if (valid) {
                    if (S.exchange) store8_sc1(S.out + (size_t)row * N + n, ykeep[nb]);
                    else *(bf16x4*)(S.out + (size_t)row * N + n) = ykeep[nb];
                    *(bf16x4*)(S.xhat + (size_t)row * N + n) = xh;
                }
            }
            if (valid && j == 0 && kq == 0) S.rstd[row] = rstd;
        }
        if (S.exchange && !cluster_barrier(ctr, CL_W * ++bars, a.abort_flag, &s_dead)) return;
        A = S.out;
        a_exchanged = S.exchange != 0;
    }
    // the last workgroup of the cluster to get here leaves the counters at zero for the next launch (everybody is past every barrier)
    if (threadIdx.x == 0) {
        unsigned* done = a.done + 32 * c;
        if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == CL_W - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct ClWorkspace {
    f32x2* part = nullptr;
    unsigned *ctr = nullptr, *done = nullptr;
    unsigned *abort_host = nullptr, *abort_dev = nullptr;
    int cus = 0;
};
ClWorkspace g_ws;

int cluster_workspace(const char* who) {
    if (g_ws.part) return ASR_OK;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: device query failed", who);
    g_ws.cus = prop.multiProcessorCount;
    char* base = nullptr;
    const size_t part_bytes = sizeof(f32x2) * CL_N * CL_MAX_RPC * CL_W, ctr_bytes = 2 * 4 * 32 * CL_N;
    if (hipMalloc((void**)&base, part_bytes + ctr_bytes) != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: hipMalloc of the cluster workspace failed", who);
    if (hipMemset(base, 0, part_bytes + ctr_bytes) != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: hipMemset failed", who);
    if (hipHostMalloc((void**)&g_ws.abort_host, 64, hipHostMallocMapped) != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: hipHostMalloc failed", who);
    *g_ws.abort_host = 0;
    if (hipHostGetDevicePointer((void**)&g_ws.abort_dev, g_ws.abort_host, 0) != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: hipHostGetDevicePointer failed", who);
    if (hipDeviceSynchronize() != hipSuccess) ASR_FAIL(ASR_EHIP, "%s: synchronize failed", who);
    g_ws.ctr = (unsigned*)(base + part_bytes);
    g_ws.done = g_ws.ctr + 32 * CL_N;
    g_ws.part = (f32x2*)base;
    return ASR_OK;
}

bool stage_shape_ok(int N, int K, int mode) {
    const int nbk = N / (16 * CL_W);
    if (N <= 0 || N % (16 * CL_W) || !(nbk == 4 || nbk == 8 || nbk == 12) || K <= 0 || K % CL_KC) return false;
    return mode != CL_LN || nbk == CL_LN_NB;
}

int launch_cluster(ClArgs& a, const char* who, hipStream_t st) {
    const int rc = cluster_workspace(who);
    if (rc != ASR_OK) return rc;
    if (*(volatile unsigned*)g_ws.abort_host)
        ASR_FAIL(ASR_EHIP, "%s: an earlier cluster kernel gave up at a barrier (its workgroups were not all resident within the spin limit): results since then are invalid", who);
    if (g_ws.cus < CL_N * CL_W) ASR_FAIL(ASR_EINVAL, "%s: the cluster kernels need %d CUs (device has %d)", who, CL_N * CL_W, g_ws.cus);
    a.rpc = ceil_div(a.M, CL_N);
    if (a.rpc > CL_MAX_RPC) ASR_FAIL(ASR_EINVAL, "%s: %d rows per cluster (limit %d)", who, a.rpc, CL_MAX_RPC);
    for (int s = 0; s < a.nstage; ++s) {
        const ClStage& S = a.st[s];
        if (!stage_shape_ok(S.N, S.K, S.mode)) ASR_FAIL(ASR_EINVAL, "%s: stage %d (N=%d K=%d) is outside the cluster kernel's shapes (asr_decoder_chain_supported)", who, s, S.N, S.K);
        if (!S.W || !S.out || !S.bias || (S.mode == CL_LN && (!S.gamma || !S.beta || !S.xhat || !S.rstd))) ASR_FAIL(ASR_EINVAL, "%s: null pointer in stage %d", who, s);
        if (((uintptr_t)S.W | (uintptr_t)S.out | (uintptr_t)S.bias | (uintptr_t)S.gamma | (uintptr_t)S.beta | (uintptr_t)S.res | (uintptr_t)S.xhat) % 16)
            ASR_FAIL(ASR_EINVAL, "%s: stage %d has a pointer off a 16-byte boundary", who, s);
        if (s && S.K != a.st[s - 1].N) ASR_FAIL(ASR_EINVAL, "%s: stage %d reduces over %d columns, stage %d produced %d", who, s, S.K, s - 1, a.st[s - 1].N);
        if (S.mode == CL_LN && !S.res && !(s >= 2 && a.st[s - 2].mode == CL_LN) && !(s >= 1 && a.st[s - 1].mode == CL_LN))
            ASR_FAIL(ASR_EINVAL, "%s: stage %d has no residual", who, s);
    }
    if (!a.A0 || (uintptr_t)a.A0 % 16) ASR_FAIL(ASR_EINVAL, "%s: bad input pointer", who);
    a.part = g_ws.part; a.ctr = g_ws.ctr; a.done = g_ws.done; a.abort_flag = g_ws.abort_dev;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)dec_cluster_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CL_TILE * 2);
        attr = true;
    }
    asr_launch_armed(dec_cluster_kernel, dim3(CL_N * CL_W), dim3(512), 2 * CL_TILE * 2, st, a);      // 8 waves copy tiles; those past the cluster's rows do nothing else
    ASR_CHECK_LAUNCH(who);
    return ASR_OK;
}

ClStage plain(const void* W, const float* bias, int N, int K, void* out, int mode, int exchange) {
    ClStage S = {};
    S.W = (const bf16_t*)W; S.bias = bias; S.N = N; S.K = K; S.mode = mode; S.out = (bf16_t*)out; S.exchange = exchange;
    return S;
}
ClStage with_ln(const void* W, const float* bias, int N, int K, void* y, const void* res, const float* g, const float* be, void* xhat, float* rstd, int exchange) {
    ClStage S = plain(W, bias, N, K, y, CL_LN, exchange);
    S.res = (const bf16_t*)res; S.gamma = g; S.beta = be; S.xhat = (bf16_t*)xhat; S.rstd = rstd;
    return S;
}

}      // namespace

// 1: the cluster kernels cover a decoder layer of these widths (model width 512 = one 16-column block per workgroup, the other widths
// multiples of 512 up to 1536) on M = B * To rows (<= 128 per cluster); the caller then may set asr_dec_layer_plan.chain.
extern "C" int asr_decoder_chain_supported(int d, int hd, int ff, int M) {
    return M > 0 && ceil_div(M, CL_N) <= CL_MAX_RPC && stage_shape_ok(d, hd, CL_LN) && stage_shape_ok(hd, d, CL_PLAIN) && stage_shape_ok(ff, d, CL_RELU) && stage_shape_ok(d, ff, CL_LN) &&
           stage_shape_ok(3 * hd, d, CL_PLAIN);
}

extern "C" int asr_decoder_chain_prepare(void) { return cluster_workspace("asr_decoder_chain_prepare"); }

// chain A of a layer: self-attention context -> out-projection, residual + LayerNorm -> Q projection of the encoder-decoder attention
int asr_dec_chain_a(const asr_dec_layer_plan* p, void* stream) {
    const int d = p->d, hd = p->H * p->dk;
    ClArgs a = {};
    a.A0 = (const bf16_t*)p->ctx_s; a.M = p->B * p->To; a.To = p->To; a.lens = p->dec_len; a.nstage = 2;
    a.st[0] = with_ln(p->w_fc_s, p->b_fc_s, d, hd, p->y_s, p->x_in, p->g_s, p->be_s, p->a_s, p->rstd_s, 1);
    a.st[1] = plain(p->w_q_c, p->b_q_c, hd, d, p->q_c, CL_PLAIN, 0);
    return launch_cluster(a, "asr_decoder_layer_fwd (chain A)", (hipStream_t)stream);
}

// chain B: cross-attention context -> out-projection, residual + LayerNorm -> w_1, ReLU -> w_2, residual + LayerNorm [-> Q|K|V of the next layer]
int asr_dec_chain_b(const asr_dec_layer_plan* p, void* stream) {
    const int d = p->d, hd = p->H * p->dk, ff = p->ff;
    const asr_dec_layer_plan* q = p->next;
    ClArgs a = {};
    a.A0 = (const bf16_t*)p->ctx_c; a.M = p->B * p->To; a.To = p->To; a.lens = p->dec_len; a.nstage = q ? 4 : 3;
    a.st[0] = with_ln(p->w_fc_c, p->b_fc_c, d, hd, p->y_c, p->y_s, p->g_c, p->be_c, p->a_c, p->rstd_c, 1);
    a.st[1] = plain(p->w_1, p->b_1, ff, d, p->h, CL_RELU, 1);
    a.st[2] = with_ln(p->w_2, p->b_2, d, ff, p->y_f, nullptr, p->g_f, p->be_f, p->o, p->rstd_f, q ? 1 : 0);      // residual y_c: still in this lane's registers
    if (q) a.st[3] = plain(q->w_qkv_s, q->b_qkv_s, 3 * hd, d, q->qkv_s, CL_PLAIN, 0);
    return launch_cluster(a, "asr_decoder_layer_fwd (chain B)", (hipStream_t)stream);
}
