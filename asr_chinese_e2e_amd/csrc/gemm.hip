// bf16 MFMA GEMMs for the dense projections (fp32 accumulate).
//
//   asr_gemm_nt_bf16 : C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ res[m][n])
//       forward of nn.Linear / Conv1d(k=1) with W stored (out, in), and - with W^T - their dgrad.
//       128 x 128 x 64 tiles, 4 waves (2 x 2), each wave 64 x 64 as 2 x 2 MFMA 32x32x16 tiles.
//       The product is issued as D = W_frag x A_frag, i.e. the accumulator holds C^T (n in
//       registers, m on the lane): bias is then a per-register constant and every lane stores
//       4 consecutive n as one 8-byte piece.  Both operands are K-contiguous, so fragments are
//       plain ds_read_b128 rows of LDS tiles with a 144-byte row stride (conflict-free).
//       Global->LDS staging goes through registers, issued one k-tile ahead (guide T14).
//   asr_gemm_tn_bf16 : dW[n][k] (+)= sum_m dY[m][n] X[m][k]     (weight gradient)
//       the reduction index m is the ROW index of both operands, so both fragments come from
//       ds_read_b64_tr_b16 (hardware transpose) on 64-row LDS tiles with a 320-byte row stride
//       (4 consecutive rows land on disjoint bank quarters).  The M range is split across
//       workgroups (grid.y); partial tiles are added with fp32 atomics (full-rate shape: every
//       wave-instruction adds two 128-byte row segments).
#include <stdlib.h>

#include <type_traits>

#include "asr_common.h"

namespace {

// Switches that produce WRONG results (timing experiments: kernels with their atomics, DMA or MFMAs removed) exist only in
// builds made with `make DEBUG_SWITCHES=1` (-DASR_DEBUG_SWITCHES); the shipped library ignores the variables, so a stray
// environment variable cannot silently corrupt gradients.
static int unsafe_debug_env(const char* name) {
#ifdef ASR_DEBUG_SWITCHES
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int AS = 72;  // LDS row stride (elements) of the NT tiles

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

struct Stage4 { u32x4 v[4]; };

// rows [row0, row0+128) x k [k0, k0+64) of a (rows, ld) K-contiguous matrix; OOB -> 0
__device__ __forceinline__ void nt_load(Stage4& st, const bf16_t* __restrict__ base, size_t ld, int row0, int rows, int k0, int K, int tid) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        const int gr = row0 + row, gk = k0 + ch * 8;
        u32x4 z = {0u, 0u, 0u, 0u};
        st.v[c] = (gr < rows && gk < K) ? *(const u32x4*)(base + (size_t)gr * ld + gk) : z;
    }
}
__device__ __forceinline__ void nt_store(const Stage4& st, bf16_t* tile, int tid) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        *(u32x4*)(tile + row * AS + ch * 8) = st.v[c];
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                      const bf16_t* __restrict__ res, bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                      int ldc, int tiles_n) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BM * AS];
    bf16_t* As = smem;
    bf16_t* Ws = smem + BM * AS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = w >> 1, wn = w & 1;
    f32x16 acc[2][2];  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    Stage4 sa, sw;
    nt_load(sa, A, lda, m0, M, 0, K, tid);
    nt_load(sw, W, ldb, n0, N, 0, K, tid);
    const int r = lane & 31, hh = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        nt_store(sa, As, tid);
        nt_store(sw, Ws, tid);
        __syncthreads();
        if (k0 + BK < K) {
            nt_load(sa, A, lda, m0, M, k0 + BK, K, tid);
            nt_load(sw, W, ldb, n0, N, k0 + BK, K, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8*)(As + (wm * 64 + i * 32 + r) * AS + 16 * ks + 8 * hh);
                wf[i] = *(const bf16x8*)(Ws + (wn * 64 + i * 32 + r) * AS + 16 * ks + 8 * hh);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    // epilogue: accumulator = C^T tile (n in registers, m on the lane)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + r;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn * 64 + ni * 32 + 8 * g4 + 4 * hh;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g4 + e];
                if (n + 3 < N) {
                    if (bias) {
                        const f32x4 b4 = *(const f32x4*)(bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b4[e];
                    }
                    if (ACT == ASR_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (res) {
                        const f32x4 r4 = load4<bf16_t>(res + (size_t)m * ldc + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r4[e];
                    }
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    store4<bf16_t>(C + (size_t)m * ldc + n, o);
                } else {
                    for (int e = 0; e < 4 && n + e < N; ++e) {
                        float x = v[e] + (bias ? bias[n + e] : 0.f);
                        if (ACT == ASR_ACT_RELU) x = fmaxf(x, 0.f);
                        if (res) x += (float)res[(size_t)m * ldc + n + e];
                        C[(size_t)m * ldc + n + e] = (bf16_t)x;
                    }
                }
            }
    }
}


// ------------------------------------------------------------------------- NT: LDS-DMA building blocks
// Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging and no ds_write) in whole 128-byte lines; an LDS-DMA
// wave-instruction writes 1 KiB lane-linearly (8 rows of 128 B), so the bank swizzle goes on the per-lane SOURCE address (16-byte chunk c of
// row R lands in slot c ^ f(R)) and is undone on the fragment read.  Rows past M / N are clamped on load (their results are never stored).
// (The non-persistent 128/256 x 128 kernels of rounds 1 - 2 that introduced this - two workgroups per CU, ring of 2 or 3 - are in the git
// history: gemm_nt_dma_kernel; every shape they served runs on the persistent loader / consumer kernel below.)
constexpr int DBK = 64;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// CUs the planners of the one-workgroup-per-CU kernels (persistent NT grid, weight-gradient M-splits) size their launches for: the
// device's, or fewer under the tuning option "cu_limit" (> 0) - the engine sets it around the large launches it puts on the auxiliary /
// weight-gradient streams BESIDE the decoder's chain of small kernels, whose workgroups need whole CUs too (132 - 148 KiB of LDS) and
// otherwise wait for one of these launches to finish.
static int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    const int lim = asr_option(ASR_OPT_CU_LIMIT);
    return lim > 0 && lim < n ? (lim < 8 ? 8 : lim) : n;
}

// ------------------------------------------------------------------------- NT, persistent
// Attribution of the non-persistent LDS-DMA kernel on the config-2 shapes (rounds 1 - 2): one k-step plus
// the fixed per-tile costs (dispatch, first-tile latency, store tail) was HALF its time, and the
// other half DMA latency + MFMA in series (ring 2: one stage in flight per workgroup); LDS
// fragment reads and bank conflicts were not on the critical path.  So the persistent form
//   * is PERSISTENT: one workgroup per CU walks a list of tiles, and the LDS-DMA ring runs across
//     tile boundaries - the first stages of the next tile are in flight during the store tail;
//   * uses 256 x 128 x 64 tiles (4 waves, wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 tiles): 0.75 x
//     the staged bytes per flop of 128 x 128, and a 3-stage ring (144 KiB) keeps TWO 48-KiB
//     stages in flight under every MFMA block (counted vmcnt, one raw s_barrier per k-step);
//   * stores through wave-private 8-KiB LDS transpose buffers (64 rows x 128 B, XOR-swizzled) placed
//     in the ring slot that was consumed last, while the other two slots receive the next tile;
//   * orders each XCD's tiles so that the 32 CUs of an XCD work on consecutive tiles (shared A
//     row-panels and W served by that XCD's L2).
constexpr int PBM = 256, PBN = 128, PBK = 64, PRING = 3;
constexpr int NT_ACT_ADD_RES = 3;      // kernel-internal: ASR_ACT_NONE with a residual operand (C = A W^T + bias + res; res may be C itself)
constexpr int PSTAGE = (PBM + PBN) * 128;          // 49152 B
constexpr int PLDS = PRING * PSTAGE;               // 147456 B
// ------------------------------------------------------------------------- NT, persistent, loader / consumer waves
// In the round-2 form of this kernel (gemm_nt_persist_kernel, git history) every wave issued its share of the LDS-DMA between its own MFMAs,
// and a 1-KiB DMA instruction holds the issuing wave for ~165 cycles (in-kernel stamps: 0.52 us of DMA issue + 0.82 us of MFMA per k-step
// overlap to 0.93 us, not to 0.82).  Here the two
// jobs belong to different waves of the same workgroup: waves 4..7 only LOAD (12 DMA instructions per k-step each, two k-steps ahead,
// counted vmcnt), waves 0..3 only COMPUTE (wave tile 128 x 64 = 4 x 2 MFMA blocks, 32 back-to-back MFMAs per k-step, fragments read
// one k-sub-step ahead) and store.  Same tile (256 x 128 x 64), same ring (3 x 48 KiB), same one s_barrier per k-step, same tile
// order; a loader and a consumer share each SIMD, so DMA issue and matrix issue come from different instruction streams.
typedef float f32x8u __attribute__((ext_vector_type(8), aligned(4)));      // eight floats at any 4-byte aligned address (scalar loads)

template <int ACT, bool KRAG = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_spec_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int tiles_n, int ntiles,
                                                              const bf16_t* __restrict__ mask, const void* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nb_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int tq = ntiles >> 3, trem = ntiles & 7;
    const int lo = xcd * tq + min(xcd, trem), hi = lo + tq + (xcd < trem ? 1 : 0);
    const int first = lo + idx;
    if (first >= hi) return;
    const int my_tiles = (hi - first + nb_x - 1) / nb_x;
    const int nk = KRAG ? (K + PBK - 1) / PBK : K / PBK;
    const int total = my_tiles * nk;
    const int srow = lane >> 3;
    if (w >= 4) {
        // ================================================================ loader waves
        constexpr int LP = (PBM + PBN) / 8 / 4;      // 12 one-KiB pieces per loader wave and stage
        const int lw = w - 4;
        const bf16_t* src[LP];
        auto set_tile = [&](int tile) {
            const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
            const int m0 = tm * PBM, n0 = tn * PBN;
#pragma unroll
            for (int j = 0; j < LP; ++j) {
                const int g = lw * LP + j;                       // 8-row group: 0..31 = A rows, 32..47 = W rows
                const int schunk = (lane & 7) ^ (((g & 1) << 2) | (srow >> 1));
                src[j] = g < PBM / 8 ? A + (size_t)min(m0 + 8 * g + srow, M - 1) * lda + schunk * 8
                                     : W + (size_t)min(n0 + 8 * (g - PBM / 8) + srow, N - 1) * ldb + schunk * 8;
            }
        };
        int it_tile = first, it_k = 0, it_slot = 0, issued = 0;
        auto issue_stage = [&]() {
#pragma unroll
            for (int j = 0; j < LP; ++j) {
                const bf16_t* p = src[j] + it_k * PBK;
                if (KRAG && it_k == nk - 1) {
                    const int g = lw * LP + j;
                    const int schunk = (lane & 7) ^ (((g & 1) << 2) | (srow >> 1));
                    p = (it_k * PBK + schunk * 8 < K) ? p : (const bf16_t*)zero_page;
                }
                __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(smem_s + it_slot * PSTAGE + (lw * LP + j) * 1024), 16, 0, 0);
            }
            ++issued;
            it_slot = it_slot == PRING - 1 ? 0 : it_slot + 1;
            if (++it_k == nk) {
                it_k = 0;
                it_tile += nb_x;
                if (issued < total) set_tile(it_tile);
            }
        };
        set_tile(first);
        issue_stage();
        if (total > 1) issue_stage();
        int i = 0;
        for (int c_tile = first; c_tile < hi; c_tile += nb_x) {
            for (int c_k = 0; c_k < nk; ++c_k, ++i) {
                // item i has landed when only item i + 1 (LP instructions) is outstanding
                if (i + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // item i is complete for the consumers; they have finished item i - 1, whose slot is refilled now
                if (issued < total) issue_stage();
            }
            __builtin_amdgcn_s_barrier();          // the consumers' barrier in front of their store tail
        }
        return;
    }
    // ==================================================================== consumer waves
    constexpr int MI = 4;
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int a_row = (wm * 128 + r) * 128;                  // + mi * 32 * 128
    const int w_row = PBM * 128 + (wn * 64 + r) * 128;       // + ni * 32 * 128
    int c_slot = 0;
    const uint32_t lane_off = ((uint32_t)srow * (uint32_t)ldc + (uint32_t)(lane & 7) * 8u) * 2u;      // this lane's piece inside an 8-row group of C (BYTES: zext(offset) + uniform base is the scalar-base address form)
    for (int c_tile = first; c_tile < hi; c_tile += nb_x) {
        const int tm = c_tile / tiles_n, tn = c_tile - tm * tiles_n;
        const int m0 = tm * PBM + wm * 128, n0 = tn * PBN + wn * 64;
        // [ni][mi]: C^T blocks (n in registers, m on the lane), starting at the bias: eight scalar loads of eight floats (one uniform test
        // of the pointer; as 64 single loads behind 64 tests of it this prologue kept the tile's first MFMA waiting for ~1 us)
        f32x16 acc[2][MI];
        if (!(ACT == ASR_ACT_RELU_MASK || ACT == NT_ACT_ADD_RES) && bias) {      // the masked / residual forms are launched without a bias (the host sends bias + mask to the all-in-one kernel)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int nb = __builtin_amdgcn_readfirstlane(min(n0 + ni * 32 + 8 * g4, N - 8));
                    const f32x8u b8 = *(const f32x8u*)(bias + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = hh ? b8[4 + e] : b8[e];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[ni][mi][4 * g4 + e] = bv;
                    }
                }
        } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ni][mi][i] = 0.f;
        }
        // ReLU mask / residual tile (64 KiB per workgroup): the pieces of store rounds 0 and 1 are fetched under the k-loop (from k-step
        // 2 on: the ring is primed by then), rounds 2 and 3 two store rounds ahead of their use.  Fetched where the tail starts (round 0 at the last
        // k-step, round mi + 1 under round mi) every workgroup asked for its 64 KiB at the same moment - 16 MB at once, 11 us per launch
        // on top of the plain product's 22 (tools/gemm_bench.py mask).
        constexpr bool MASKED = ACT == ASR_ACT_RELU_MASK || ACT == NT_ACT_ADD_RES;
        constexpr int NPF = 2;      // rounds fetched under the k-loop (three: the kernel spills)
        u32x4 hmp[NPF][4];
        const int pf_k = nk > 2 ? 2 : nk - 1;
        // A tile that lies wholly inside the matrix (all but the last row / column of tiles) loads and stores without predicates:
        // the compiler then counts the outstanding memory operations (vmcnt(N), N > 0) and a round never waits for the stores of
        // the round before it; behind per-lane predicates it falls back to vmcnt(0) in front of every use of a fetched piece.
        const bool full = tm * PBM + PBM <= M && tn * PBN + PBN <= N;      // wave-uniform
        auto load_masks = [&](u32x4 (&dst)[4], int mi, auto full_c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = q * 8 + srow, ch = lane & 7;
                const int m = m0 + mi * 32 + row, n = n0 + ch * 8;
                const u32x4 z = {0u, 0u, 0u, 0u};
                // uniform row pointer + one per-lane 32-bit offset (scalar base addressing: no 64-bit address pair per piece)
                const bf16_t* rowp = mask + (size_t)(m0 + mi * 32 + q * 8) * ldc + n0;
                if constexpr (decltype(full_c)::value) dst[q] = *(const u32x4*)((const char*)rowp + lane_off);
                else dst[q] = (m < M && n + 8 <= N) ? *(const u32x4*)((const char*)rowp + lane_off) : z;
            }
        };
        for (int c_k = 0; c_k < nk; ++c_k) {
            __builtin_amdgcn_s_barrier();
            if (MASKED && c_k == pf_k) {
                if (full) {
#pragma unroll
                    for (int mi = 0; mi < NPF; ++mi) load_masks(hmp[mi], mi, std::true_type{});
                } else {
#pragma unroll
                    for (int mi = 0; mi < NPF; ++mi) load_masks(hmp[mi], mi, std::false_type{});
                }
            }
            const char* sb = smem_s + c_slot * PSTAGE;
            c_slot = c_slot == PRING - 1 ? 0 : c_slot + 1;
            bf16x8 af[2][MI], wf[2][2];
            auto load_frags = [&](int buf, int ks) {
                const int coff = ((2 * ks + hh) ^ sw) << 4;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[buf][mi] = *(const bf16x8*)(sb + a_row + mi * 32 * 128 + coff);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) wf[buf][ni] = *(const bf16x8*)(sb + w_row + ni * 32 * 128 + coff);
            };
            load_frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) load_frags((ks + 1) & 1, ks + 1);
#pragma unroll
                for (int q = 0; q < 2 * MI; ++q) {
                    const int ni = q / MI, mi = q % MI;
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][ni], af[ks & 1][mi], acc[ni][mi], 0, 0, 0);
                }
            }
        }
        // ---- store tail: four rounds of 32 rows x 64 columns through a wave-private 4-KiB buffer in the slot consumed last
        __builtin_amdgcn_s_barrier();
        char* epi = smem_s + (c_slot == 0 ? PRING - 1 : c_slot - 1) * PSTAGE + w * 4096;
        auto store_tail = [&](auto full_c) {
            u32x4 hm[4], hmt[MI - NPF][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (MASKED) {
                    if (mi + NPF < MI) load_masks(hmt[mi], mi + NPF, full_c);      // two rounds ahead; round mi - 1's registers are free by now
#pragma unroll
                    for (int q = 0; q < 4; ++q) hm[q] = mi < NPF ? hmp[mi < NPF ? mi : 0][q] : hmt[mi < NPF ? 0 : mi - NPF][q];
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[ni][mi][4 * g4 + e];
                            if (ACT == ASR_ACT_RELU) x = fmaxf(x, 0.f);
                            o[e] = x;
                        }
                        store4<bf16_t>((bf16_t*)(epi + r * 128 + (((ni * 4 + g4) ^ sw) << 4) + hh * 8), o);
                    }
                __builtin_amdgcn_wave_barrier();
                u32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 8 + srow, ch = lane & 7;
                    v[q] = *(const u32x4*)(epi + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
                }
                if (ACT == NT_ACT_ADD_RES) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t a = v[q][e], r2 = hm[q][e];
                            const bf16_t lo_ = (bf16_t)(__uint_as_float(a << 16) + __uint_as_float(r2 << 16));
                            const bf16_t hi_ = (bf16_t)(__uint_as_float(a & 0xffff0000u) + __uint_as_float(r2 & 0xffff0000u));
                            v[q][e] = (uint32_t)__builtin_bit_cast(unsigned short, lo_) | ((uint32_t)__builtin_bit_cast(unsigned short, hi_) << 16);
                        }
                }
                if (ACT == ASR_ACT_RELU_MASK) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t h = hm[q][e];
                            const uint32_t lo_ = ((h & 0x7fffu) != 0u && !(h & 0x8000u)) ? 0x0000ffffu : 0u;
                            const uint32_t hi_ = ((h & 0x7fff0000u) != 0u && !(h & 0x80000000u)) ? 0xffff0000u : 0u;
                            v[q][e] &= lo_ | hi_;
                        }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 8 + srow, ch = lane & 7;
                    const int m = m0 + mi * 32 + row, n = n0 + ch * 8;
                    bf16_t* rowp = C + (size_t)(m0 + mi * 32 + q * 8) * ldc + n0;
                    if constexpr (decltype(full_c)::value) stream_store(v[q], (u32x4*)((char*)rowp + lane_off));
                    else if (m < M && n + 8 <= N) stream_store(v[q], (u32x4*)((char*)rowp + lane_off));
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (full) store_tail(std::true_type{});
        else store_tail(std::false_type{});
    }
}

template <int ACT, bool KRAG = false>
static void launch_nt_spec(const bf16_t* a, const bf16_t* w, const float* bias, bf16_t* c, int M, int N, int K, int lda, int ldb, int ldc, hipStream_t st,
                           const bf16_t* mask = nullptr, const void* zero_page = nullptr) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_spec_kernel<ACT, KRAG>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        attr = true;
    }
    const int t_n = ceil_div(N, PBN), t_m = ceil_div(M, PBM), ntiles = t_n * t_m;
    const int grid = ntiles < cu_count() ? ntiles : cu_count();
    asr_launch_armed(gemm_nt_spec_kernel<ACT, KRAG>, dim3(grid), dim3(512), PLDS, st, a, w, bias, c, M, N, K, lda, ldb, ldc, t_n, ntiles, mask, zero_page);
}

// ------------------------------------------------------------------------- small-M projections (decoder)
// The decoder works on B * To ~ 550 rows: a 256 x 128 persistent tile gives such a GEMM 12 workgroups and ~11 us of
// mostly fixed cost (144 KiB ring prologue, store tail), and the library GEMM its input gradients go to costs the host
// ~28 us per call (the joint step is host-bound as much as GPU-bound).  This kernel: 64 x 64 tiles (72 .. 216 workgroups),
// 4 waves (2 x 2, one MFMA 32x32x16 block each), register-staged double-buffered LDS tiles, one barrier per 64-deep k-step.
//   TB = false: B is (N, K), K-contiguous: C = A B^T            (forward: y = x W^T + bias, optional ReLU)
//   TB = true : B is (K, N), N-contiguous: C = A B              (input gradient: dx = dy W; W as stored, no transposed copy:
//               the operand fragments come from transposed LDS reads in natural k order)
// mask (TB only): C = 0 where mask <= 0 (the ReLU backward of module.py:70-71 in the store tail).
constexpr int SM = 64, SS = 72;      // tile edge, LDS row stride (elements) of a 64-column tile
// k-step: 256.  With 64-deep steps the kernel ran one global-load round trip per step (the next step's loads are only
// covered by 4 MFMAs): 8 .. 24 round trips, 13 us on average.  A 256-deep step is 2 .. 6 round trips; the step's operands
// (A 64 x 256, B 64 x 256 or 256 x 64) are 16 16-byte loads per thread, in flight together.
constexpr int SK = 256;
constexpr int SAS = SK + 8;          // row stride (elements) of a K-contiguous 64 x 256 tile: 528 B, rows 4 banks apart
constexpr int S_A = SM * SAS;        // elements of the A tile (and of a K-contiguous B tile)
constexpr int S_BT = SK * SS;        // elements of a [k][n] B tile
struct StageK { u32x4 v[8]; };
// rows [row0, row0+64) x columns [c0, c0+256) of a (rows, ld) matrix; OOB -> 0
__device__ __forceinline__ void sk_load_rows(StageK& st, const bf16_t* __restrict__ base, size_t ld, int row0, int rows, int c0, int cols, int tid) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int id = tid + 256 * c, row = id >> 5, ch = id & 31;
        const int gr = row0 + row, gc = c0 + ch * 8;
        const u32x4 z = {0u, 0u, 0u, 0u};
        st.v[c] = (gr < rows && gc < cols) ? *(const u32x4*)(base + (size_t)gr * ld + gc) : z;
    }
}
__device__ __forceinline__ void sk_store_rows(const StageK& st, bf16_t* tile, int tid) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int id = tid + 256 * c, row = id >> 5, ch = id & 31;
        *(u32x4*)(tile + row * SAS + ch * 8) = st.v[c];
    }
}
// rows (= k) [row0, row0+256) x columns (= n) [c0, c0+64)
__device__ __forceinline__ void sk_load_cols(StageK& st, const bf16_t* __restrict__ base, size_t ld, int row0, int rows, int c0, int cols, int tid) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        const int gr = row0 + row, gc = c0 + ch * 8;
        const u32x4 z = {0u, 0u, 0u, 0u};
        st.v[c] = (gr < rows && gc < cols) ? *(const u32x4*)(base + (size_t)gr * ld + gc) : z;
    }
}
__device__ __forceinline__ void sk_store_cols(const StageK& st, bf16_t* tile, int tid) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int id = tid + 256 * c, row = id >> 3, ch = id & 7;
        *(u32x4*)(tile + row * SS + ch * 8) = st.v[c];
    }
}
// operand fragment from a [k][n] tile by transposed reads, natural k order: lane (r, hh), j -> tile[16 s + 8 hh + j][col0 + r]
__device__ __forceinline__ bf16x8 sm_frag_tr(const bf16_t* tile, int col0, int s, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const bf16_t* p = tile + (16 * s + 8 * (G >> 1) + (i >> 2)) * SS + col0 + 16 * (G & 1) + 4 * (i & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * SS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool TB>
constexpr int small_lds_bytes() { return 2 * (S_A + (TB ? S_BT : S_A)) * 2; }

template <bool TB, int ACT>
__global__ __launch_bounds__(256) void gemm_small_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, const float* __restrict__ bias,
                                                         const bf16_t* __restrict__ mask, bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                         int ldc, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem_s[];      // [2 buffers][A | B]
    constexpr int BUF = S_A + (TB ? S_BT : S_A);
    bf16_t* smem = (bf16_t*)smem_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * SM, n0 = tn * SM;
    const int wm = w >> 1, wn = w & 1, r = lane & 31, hh = lane >> 5;
    f32x16 acc[2];   // C^T block: n in registers, m on the lane; two accumulators (even / odd k-steps) halve the dependent MFMA chain
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    // two register stages: the loads of steps kt + 1 AND kt + 2 are in flight under the MFMAs of step kt (a workgroup is one
    // wave per SIMD and nothing else hides the global latency)
    StageK sa0, sb0, sa1, sb1;
#define SK_LOAD(SA_, SB_, K0_)                                                                 \
    {                                                                                          \
        sk_load_rows(SA_, A, lda, m0, M, (K0_), K, tid);                                       \
        if (TB) sk_load_cols(SB_, Bm, ldb, (K0_), K, n0, N, tid); /* rows = k, columns = n */   \
        else sk_load_rows(SB_, Bm, ldb, n0, N, (K0_), K, tid);    /* rows = n, columns = k */   \
    }
#define SK_STORE(SA_, SB_, BUF_)                                                               \
    {                                                                                          \
        sk_store_rows(SA_, (BUF_), tid);                                                       \
        if (TB) sk_store_cols(SB_, (BUF_) + S_A, tid);                                         \
        else sk_store_rows(SB_, (BUF_) + S_A, tid);                                            \
    }
#define SK_MMA(KT_)                                                                                                                   \
    {                                                                                                                                 \
        const bf16_t* As = smem + ((KT_) & 1) * BUF;                                                                                  \
        const bf16_t* Bs = As + S_A;                                                                                                  \
        /* all 16 MFMA steps, unrolled: the loads zero-pad a partial last step, and a run-time trip count made acc[ks & 1] a   */     \
        /* run-time register index (s_set_gpr_idx + 32 register copies per MFMA: 4.3 us per step)                            */     \
        _Pragma("unroll") for (int ks = 0; ks < SK / 16; ++ks) {                                                                      \
            const bf16x8 af = *(const bf16x8*)(As + (wm * 32 + r) * SAS + 16 * ks + 8 * hh);                                          \
            const bf16x8 bf = TB ? sm_frag_tr(Bs, wn * 32, ks, lane) : *(const bf16x8*)(Bs + (wn * 32 + r) * SAS + 16 * ks + 8 * hh); \
            acc[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[ks & 1], 0, 0, 0);                                      \
        }                                                                                                                             \
    }
    const int nk = (K + SK - 1) / SK;
    SK_LOAD(sa0, sb0, 0);
    if (nk > 1) SK_LOAD(sa1, sb1, SK);
    SK_STORE(sa0, sb0, smem);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: buffer 0 holds step kt, stage 1 holds step kt + 1
        if (kt + 2 < nk) SK_LOAD(sa0, sb0, (kt + 2) * SK);
        SK_MMA(kt);
        if (kt + 1 < nk) SK_STORE(sa1, sb1, smem + BUF);      // buffer 1 was last read in step kt - 1, before that step's barrier
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: buffer 1 holds step kt + 1, stage 0 holds step kt + 2
        if (kt + 3 < nk) SK_LOAD(sa1, sb1, (kt + 3) * SK);
        SK_MMA(kt + 1);
        if (kt + 2 < nk) SK_STORE(sa0, sb0, smem);
        __syncthreads();
    }
#undef SK_LOAD
#undef SK_STORE
#undef SK_MMA
    const int m = m0 + wm * 32 + r;
    if (m >= M) return;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 32 + 8 * g4 + 4 * hh;
        if (n >= N) continue;      // N % 8 == 0: a 4-element piece is inside or outside as a whole
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[0][4 * g4 + e] + acc[1][4 * g4 + e] + (bias ? bias[n + e] : 0.f);
            if (ACT == ASR_ACT_RELU) x = fmaxf(x, 0.f);
            o[e] = x;
        }
        if (ACT == ASR_ACT_RELU_MASK) {
            const f32x4 h4 = load4<bf16_t>(mask + (size_t)m * ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = h4[e] > 0.f ? o[e] : 0.f;
        }
        store4<bf16_t>(C + (size_t)m * ldc + n, o);
    }
}

// ---------------------------------------------------------------------------------------- TN
constexpr int TM = 64;     // reduction rows per LDS stage

// ------------------------------------------------------------------------- TN, LDS-DMA version
// Same pipeline as the persistent NT kernel, applied to the weight gradient: a 4-stage ring of
// 64-row stages (dY 64 x 128 | X 64 x 128, 32 KiB) filled by LDS-DMA, one raw s_barrier per
// stage, counted vmcnt with two to three stages in flight, the eight DMA instructions of a wave
// issued BETWEEN its 16 MFMAs.  The register-staged kernel above spends ~1.3 us per 64-row stage
// where the MFMAs need 0.33 us; the limit of this form is the LDS array (transposed fragment reads
// of both operands + the DMA writes: ~96 KiB per stage).
//   * a DMA instruction writes 1 KiB = 4 rows of 256 B; 16-byte chunk c of row R is stored in
//     slot c ^ (4 * (R & 3)) (swizzle on the source address), so the four rows one half-wave
//     reads with ds_read_b64_tr_b16 sit on disjoint 64-byte bank groups;
//   * rows past the end of the split's range and columns past N / K come from a zero page.
constexpr int TSTAGE = 2 * TM * 256;   // 32768 B
__device__ __attribute__((aligned(256))) unsigned tn_zero_page[64];

// ds_read_b64_tr_b16 through inline asm: after an LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of
// the builtin form of this read (it cannot tell the read from the DMA's LDS target), which drains
// the whole prefetch ring every k-step.  The asm form is invisible to that pass, so the reads are
// ordered by hand: counted lgkmcnt waits + sched_barrier in the main loop (guide 5.4 rule 18).
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr16(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// byte offset inside a stage half (64 rows x 256 B, swizzled) of this lane's piece of the fragment
// lane (r, hh), j = 0..7  ->  tile[16*s + 8*(j>>2) + 4*hh + (j&3)][col0 + r];  + s*4096 (+2048 for j >= 4)
__device__ __forceinline__ unsigned tn_frag_off(int col0, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const int row = 4 * (G >> 1) + (i >> 2);                    // row & 3 == (i >> 2) & 3 for every s
    const int chunk = ((col0 >> 3) + 2 * (G & 1) + ((i & 3) >> 1)) ^ (4 * ((i >> 2) & 3));
    return (unsigned)(row * 256 + (chunk << 4) + 8 * (i & 1));
}
template <int S>
__device__ __forceinline__ bf16x8 tn_frag_swz(unsigned addr) {
    const bf16x4 lo = lds_tr16<S * 4096>(addr), hi = lds_tr16<S * 4096 + 2048>(addr);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The body of one workgroup: tile (n0, k0) of dW over reduction rows [mbeg, mend) - shared by the single-problem kernel and the
// multi-problem kernel below (same code, same bits).
template <int TRING>
__device__ __forceinline__ void tn_dma_tile(char* smem_t, const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, float* __restrict__ dW, int N, int K, int ldy,
                                            int ldx, int ldw, int n0, int k0, int tk, int mbeg, int mend, int use_atomic, const void* __restrict__ zero_page,
                                            float* __restrict__ dbias, int split, int bias_split_stride) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int SR = TM;                     // rows per ring slot
    constexpr int SBYTES = TSTAGE;             // bytes per ring slot: [dY 64 x 256 B | X 64 x 256 B]
    const int nsteps = (mend - mbeg + SR - 1) / SR;
    if (nsteps <= 0) return;
    const int wn = w >> 1, wk = w & 1;

    // this lane's part of each of the wave's 8 one-KiB pieces: the first half of the waves stages dY
    // (rows 4g.. of piece g), the second half X.  Running pointers, advanced by one uniform add per stage: the
    // address work per DMA must stay within a few instructions, or it - not the MFMAs - paces the
    // loop (an MFMA hides about five other instructions of its wave).
    // Columns past N / K are CLAMPED to the last valid 16-byte chunk: they only feed accumulator
    // columns that are never stored.  Rows past the end of the range must read zeros: only the last
    // stage of a range can be partial, and only there the per-lane select runs.
    const int lrow = lane >> 4, slot = lane & 15;
    const bool is_y = w < 2;
    const bf16_t* cur[8];
    int prow[8];   // row inside the stage
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gg = (w & 1) * 8 + j;
        const int row = 4 * gg + lrow;
        const int col = (slot ^ (4 * (row & 3))) * 8;
        prow[j] = row;
        cur[j] = is_y ? dY + (size_t)(mbeg + row) * ldy + min(n0 + col, N - 8) : X + (size_t)(mbeg + row) * ldx + min(k0 + col, K - 8);
    }
    const size_t stage_step = (size_t)SR * (is_y ? ldy : ldx);   // elements per stage (wave-uniform)
    const bool ragged = ((mend - mbeg) % SR) != 0;
    int it = 0, it_slot = 0;   // next stage to issue
    // rows past the end of the range read the zero page: only the last stage can be partial, and the pointers
    // are redirected once, when the cursor reaches it (a uniform branch per stage instead of selects per DMA)
    auto fix_last = [&]() {
        if (ragged && it == nsteps - 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = (mbeg + it * SR + prow[j] < mend) ? cur[j] : (const bf16_t*)zero_page;
        }
    };
    fix_last();
    auto dma = [&](int j) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)cur[j], (lds_void_t*)(smem_t + it_slot * SBYTES + (w * 8 + j) * 1024), 16, 0, 0);
    };
    auto advance = [&]() {
        ++it;
        it_slot = it_slot == TRING - 1 ? 0 : it_slot + 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] += stage_step;
        fix_last();
    };

    f32x16 acc[2][2];  // [ni][ki]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_t;
    const unsigned y_off[2] = {tn_frag_off(wn * 64, lane), tn_frag_off(wn * 64 + 32, lane)};
    const unsigned x_off[2] = {SR * 256 + tn_frag_off(wk * 64, lane), SR * 256 + tn_frag_off(wk * 64 + 32, lane)};
    // Bias gradient (column sums of dY) from the dY stages already in LDS, by the workgroups of the
    // first k-tile: thread (chunk = tid & 15, phase = tid >> 4) adds rows phase, phase + 16, ... of
    // its 8 columns; rows with equal (r & 3) keep a chunk in the same swizzled slot.  Replaces a
    // separate pass over dY (colsum + finalize launches) per projection.
    const bool do_bias = dbias != nullptr && tk == 0;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    const unsigned b_off = (unsigned)((tid >> 4) * 256 + (((tid & 15) ^ (4 * ((tid >> 4) & 3))) << 4));

    for (int p = 0; p < TRING - 1 && p < nsteps; ++p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dma(j);
        advance();
    }
    int c_slot = 0;
    for (int i = 0; i < nsteps; ++i) {
        // stage i has landed when only the younger stages (8 DMA instructions each) are outstanding
        const int younger = min(TRING - 2, nsteps - 1 - i);
        if (TRING >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (TRING >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave's part of stage i landed; nobody still reads the slot refilled during this step
        const bool more = it < nsteps;
        const unsigned sbase = lds_base + c_slot * SBYTES;
        c_slot = c_slot == TRING - 1 ? 0 : c_slot + 1;
        bf16x8 yf[2][2], xf[2][2];
#define TN_LOAD(BUF, S)                                              \
    do {                                                             \
        yf[BUF][0] = tn_frag_swz<S>(sbase + y_off[0]);               \
        yf[BUF][1] = tn_frag_swz<S>(sbase + y_off[1]);               \
        xf[BUF][0] = tn_frag_swz<S>(sbase + x_off[0]);               \
        xf[BUF][1] = tn_frag_swz<S>(sbase + x_off[1]);               \
    } while (0)
#define TN_MMA(BUF, KS)                                                                                                       \
    do {                                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                       \
            const int ni = q >> 1, ki = q & 1;                                                                                \
            acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[BUF][ni], xf[BUF][ki], acc[ni][ki], 0, 0, 0);            \
            if (q & 1) { /* one DMA instruction after every second MFMA */                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
                if (more) dma(KS * 2 + (q >> 1));                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
        if (do_bias) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const u32x4 c8 = *(const u32x4*)(smem_t + (sbase - lds_base) + b_off + qq * 16 * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(c8[e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(c8[e] & 0xffff0000u);
                }
            }
        }
        // 8 reads per k-sub-step; LDS reads retire in order, so lgkmcnt(8) = "all but the 8 just issued"
        TN_LOAD(0, 0);
        TN_LOAD(1, 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TN_MMA(0, 0);
        TN_LOAD(0, 2);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TN_MMA(1, 1);
        TN_LOAD(1, 3);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TN_MMA(0, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TN_MMA(1, 3);
#undef TN_LOAD
#undef TN_MMA
        if (more) advance();
    }
    // accumulator: row = n (registers), col = k (lane): 32 consecutive k per register -> 128-B segments
    {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
            const int kc = k0 + wk * 64 + ki * 32 + (lane & 31);
            if (kc >= K) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + ni * 32 + acc_row(e, lane);
                if (n >= N) continue;
                float* dst = dW + (size_t)n * ldw + kc;
                if (use_atomic) atomicAdd(dst, acc[ni][ki][e]);
                else *dst = acc[ni][ki][e];
            }
        }
    }
    if (do_bias) {   // 16 row phases -> one sum per column, through the (now idle) ring memory
        __syncthreads();
        float* red = (float*)smem_t;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int ph = 0; ph < 16; ++ph) t += red[ph * 128 + tid];
            if (bias_split_stride) dbias[(size_t)split * bias_split_stride + n0 + tid] = t;   // deterministic mode: slab per split
            else atomicAdd(dbias + n0 + tid, t);
        }
    }
}

// TRING ring slots of 64 rows; four waves (2 x 2 over the 128 x 128 tile).  (An 8-wave form with an intra-workgroup split of the
// reduction - two wave groups per ring slot - is in the git history: faster alone, slower beside the main stream.)
template <int TRING>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, float* __restrict__ dW, int M, int N,
                                                             int K, int ldy, int ldx, int ldw, int tiles_k, int tiles_n, int rows_per_split,
                                                             int use_atomic, const void* __restrict__ zero_page, float* __restrict__ dbias, size_t split_stride,
                                                             int bias_split_stride, int skew = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int ntiles = tiles_k * tiles_n;
    const int vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    const int split = vid / ntiles, tile = vid - split * ntiles;
    const int tk = tile % tiles_k, tn = tile / tiles_k;
    // skew > 0 (the default): split s reduces rows_per_split + skew * s rows, so the splits - which all add their tile
    // to memory with fp32 atomics when they finish - finish one after the other instead of together
    const int mbeg = split * rows_per_split + skew * (split * (split - 1) / 2), mend = min(M, mbeg + rows_per_split + skew * split);
    // deterministic mode: every split owns a slab (summed by tn_slab_reduce_kernel)
    tn_dma_tile<TRING>(smem_t, dY, X, dW + (size_t)split * split_stride, N, K, ldy, ldx, ldw, tn * 128, tk * 128, tk, mbeg, mend, use_atomic, zero_page, dbias, split,
                       bias_split_stride);
}

// Several weight gradients over the SAME M rows in ONE launch of the kernel above (round 5): the two projections of a feed-forward block
// (w_2, w_1) or of an attention block (out-projection, Q|K|V) have 64 tiles of 128 x 128 between them, so one round of one workgroup per
// CU needs 4 M-splits instead of 8 / 8 or 16 / 5 - and every workgroup ends by adding its 64-KiB fp32 tile to memory with atomics, which
// costs a launch ~12 us whatever the projection (256 workgroups x 64 KiB at the chip's ~1.3 TB/s atomic rate).  Two launches of 16 MB of
// atomics, two kernel boundaries and two ring prologues become one.  Same tile code, staggered splits as above.
constexpr int TNM_MAX = 4;
struct TnMultiProb {
    const bf16_t* dY;
    const bf16_t* X;
    float* dW;
    float* dbias;
    int N, K, ldy, ldx, ldw;
    int tiles_k, tiles, block_begin;      // 128 x 128 tiles; first workgroup (virtual id) of the problem
};
struct TnMulti {
    TnMultiProb p[TNM_MAX];
    int nprob, M, nsplit, r0, skew;
};
template <int TRING>
__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(const TnMulti grp, const void* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    int pi = 0;
    for (int i = 1; i < grp.nprob; ++i) pi = vid >= grp.p[i].block_begin ? i : pi;
    const TnMultiProb& pr = grp.p[pi];
    const int local = vid - pr.block_begin;
    const int split = local / pr.tiles, tile = local - split * pr.tiles;
    const int tk = tile % pr.tiles_k, tn = tile / pr.tiles_k;
    const int mbeg = split * grp.r0 + grp.skew * (split * (split - 1) / 2), mend = min(grp.M, mbeg + grp.r0 + grp.skew * split);
    tn_dma_tile<TRING>(smem_t, pr.dY, pr.X, pr.dW, pr.N, pr.K, pr.ldy, pr.ldx, pr.ldw, tn * 128, tk * 128, tk, mbeg, mend, 1, zero_page, pr.dbias, split, 0);
}

template <int S>
__device__ __forceinline__ bf16x8 tg_frag(unsigned addr) {   // tn_frag_swz through the builtin (compiler-counted lgkmcnt)
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(size_t)(addr + S * 4096));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(size_t)(addr + S * 4096 + 2048));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight-gradient GEMM: up to TG_MAX independent problems dW_p (+)= dY_p^T X_p in ONE
// launch.  A single projection's gradient has only N*K/128^2 = 16..48 output tiles, so filling 256
// CUs means 5..16 splits of the M = B*T reduction: every split pays N*K*4 bytes of atomics, a ring
// prologue and an epilogue.  All projections of a layer together have enough tiles for 256 x 128
// tiles (a third less LDS-DMA and LDS-read traffic per FLOP than 128 x 128: the texture path takes
// 64 B/clk/CU, exactly what a 128 x 128 tile needs at the MFMA rate) with only ~4 splits.
// Workgroup = 4 waves, tile 256 (n) x 128 (k), wave tile 128 x 64 (4 x 2 MFMA blocks, 128
// accumulator registers); stage = 64 reduction rows = [dY cols 0..127 | dY cols 128..255 | X],
// three 16-KiB panels in the layout of gemm_tn_dma_kernel; ring of 3 stages (144 KiB).
constexpr int TG_MAX = 8;
struct TnProb {
    const bf16_t* dY;
    const bf16_t* X;
    float* dW;
    float* dbias;
    int M, N, K, ldy, ldx, ldw;
    int tiles_k, tiles;            // 128-wide k-tiles; tiles = n-tiles(256) * tiles_k
    int rows_per_split, block_begin;
};
struct TnGroup {
    TnProb p[TG_MAX];
    int nprob;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ROWS reduction rows per stage (64 or 32), RING stages of 3 * ROWS * 256 bytes.  The LDS footprint decides
// what the main stream can run on the same CU beside this kernel (attention needs 18 KiB per workgroup):
// a 16-KiB pad on the single-problem kernel cost the training step 6 %.
// DBG: timing experiments (wrong results): 1 = no DMA in the main loop, 2 = no MFMA, 3 = no LDS reads
template <int ROWS, int RING, int DBG>
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const TnGroup grp, int accumulate, int no_atomic, const void* __restrict__ zero_page) {
    constexpr int KS = ROWS / 16;          // k-sub-steps (one MFMA depth) per stage
    constexpr int PPW = 3 * ROWS / 16;     // one-KiB DMA pieces per wave per stage
    constexpr int PPP = ROWS / 4;          // pieces per panel
    constexpr int PANEL = ROWS * 256;      // bytes per panel
    constexpr int STAGE = 3 * PANEL;
    constexpr int PART = 3 * (KS - 1);     // DMA instructions of the newest stage issued before the barrier of a stage
    static_assert(KS == 2 || KS == 4, "stage = 32 or 64 rows");
    static_assert(RING >= 3 && RING <= 5, "ring of 3..5 stages");
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vid = xcd_virtual_id(blockIdx.x, gridDim.x);
    int pi = 0;
    for (int i = 1; i < grp.nprob; ++i) pi = vid >= grp.p[i].block_begin ? i : pi;
    const TnProb& pr = grp.p[pi];
    const bf16_t* __restrict__ dY = pr.dY;
    const bf16_t* __restrict__ X = pr.X;
    const int M = pr.M, N = pr.N, K = pr.K, ldy = pr.ldy, ldx = pr.ldx;
    const int local = vid - pr.block_begin;
    const int split = local / pr.tiles, tile = local - split * pr.tiles;
    const int tk = tile % pr.tiles_k, tn = tile / pr.tiles_k;
    const int n0 = tn * 256, k0 = tk * 128;
    const int mbeg = split * pr.rows_per_split, mend = min(M, mbeg + pr.rows_per_split);
    const int nsteps = (mend - mbeg + ROWS - 1) / ROWS;
    if (nsteps <= 0) return;
    const int use_atomic = no_atomic ? 0 : (accumulate || pr.rows_per_split < M);
    const int wn = w >> 1, wk = w & 1;

    // 3 PPP one-KiB pieces per stage, PPW per wave: piece P = PPW w + j -> panel P / PPP, rows 4 (P % PPP)..+3
    const int lrow = lane >> 4, slot = lane & 15;
    const bf16_t* cur[PPW];
    int prow[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int P = w * PPW + j, panel = P / PPP, g = P % PPP;
        const int row = 4 * g + lrow;
        const int col = (slot ^ (4 * (row & 3))) * 8;
        prow[j] = row;
        cur[j] = panel < 2 ? dY + (size_t)(mbeg + row) * ldy + min(n0 + panel * 128 + col, N - 8) : X + (size_t)(mbeg + row) * ldx + min(k0 + col, K - 8);
    }
    const size_t ystep = (size_t)ROWS * ldy, xstep = (size_t)ROWS * ldx;
    const bool ragged = ((mend - mbeg) % ROWS) != 0;
    int it = 0, it_slot = 0;
    // rows past the end of the range read the zero page: only the last stage can be partial, and the
    // pointers are redirected once, when the cursor reaches it (a uniform branch, not per DMA)
    auto fix_last = [&]() {
        if (ragged && it == nsteps - 1) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) cur[j] = (mbeg + it * ROWS + prow[j] < mend) ? cur[j] : (const bf16_t*)zero_page;
        }
    };
    fix_last();
    // The LDS-DMA goes through inline asm and the transposed reads through the builtin - the other way
    // round than in gemm_tn_dma_kernel: the compiler then counts lgkmcnt for the reads itself (also
    // in front of any register copy it makes of a fragment - with asm reads it copied a loop-carried
    // fragment before the data had arrived), while the DMA it would order against every LDS read
    // with vmcnt(0) stays invisible to it and is waited for by hand.
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_g;
    const unsigned dma_base = __builtin_amdgcn_readfirstlane(lds_base + w * PPW * 1024);
    auto dma = [&](int j) {
        const unsigned dst = dma_base + it_slot * STAGE + j * 1024;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(cur[j]), "s"(dst) : "memory");   // m0: no compiler-generated user in this kernel (checked in the ISA)
    };
    auto advance = [&]() {
        ++it;
        it_slot = it_slot == RING - 1 ? 0 : it_slot + 1;
#pragma unroll
        for (int j = 0; j < PPW; ++j) cur[j] += (w * PPW + j < 2 * PPP) ? ystep : xstep;   // pieces below 2 PPP are dY
        fix_last();
    };

    f32x16 acc[4][2];  // [ni][ki]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    unsigned y_off[4], x_off[2];
#pragma unroll
    for (int a = 0; a < 4; ++a) y_off[a] = wn * PANEL + tn_frag_off(a * 32, lane);
#pragma unroll
    for (int b = 0; b < 2; ++b) x_off[b] = 2 * PANEL + tn_frag_off(wk * 64 + b * 32, lane);
    // bias gradient = column sums of the dY panels, by the workgroups of the first k-tile (see gemm_tn_dma_kernel)
    const bool do_bias = pr.dbias != nullptr && tk == 0;
    float bsum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bsum[e] = 0.f;
    const unsigned b_off = (unsigned)((tid >> 4) * 256 + (((tid & 15) ^ (4 * ((tid >> 4) & 3))) << 4));

    for (int p = 0; p < RING - 1 && p < nsteps; ++p) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) dma(j);
        advance();
    }
    // Software pipeline of ONE wave per SIMD: a wave issues in order, so a burst of transposed reads or a
    // DMA instruction the texture unit is not ready for stalls the MFMAs behind it.  Hence (1) the
    // reads of the next k-sub-step and the DMA instructions are spread between the 8 MFMAs of the
    // current one, and (2) the barrier for stage i + 1 sits BEFORE the last k-sub-step of stage i,
    // whose MFMAs cover the first fragment reads of stage i + 1.
    bf16x8 yf[2][4], xf[2][2];
    if (DBG == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int a = 0; a < 4; ++a) yf[u][a] = bf16x8{};
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[u][b] = bf16x8{};
        }
    }
#define TG_RDY(BUF, S, BASE, A) do { if (DBG != 3) yf[BUF][A] = tg_frag<S>((BASE) + y_off[A]); } while (0)
#define TG_RDX(BUF, S, BASE, B) do { if (DBG != 3) xf[BUF][B] = tg_frag<S>((BASE) + x_off[B]); } while (0)
#define TG_MFMA(BUF, Q) do { if (DBG != 2) acc[(Q) >> 1][(Q) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[BUF][(Q) >> 1], xf[BUF][(Q) & 1], acc[(Q) >> 1][(Q) & 1], 0, 0, 0); } while (0)
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
#define TG_DMA(J) do { if (more && DBG != 1) dma(J); } while (0)
// one k-sub-step: 8 MFMAs on buffer BUF; in their shadow the 12 reads of (slot NBASE, sub-step NS) into
// buffer 1 - BUF (if RD) and DMA instructions 3 KSI .. 3 KSI + 2 of the stage RING - 1 ahead
#define TG_KSTEP(BUF, KSI, RD, NS, NBASE)                                                  \
    do {                                                                                   \
        TG_MFMA(BUF, 0); TG_SB();                                                          \
        if (RD) { TG_RDY(1 - BUF, NS, NBASE, 0); TG_RDX(1 - BUF, NS, NBASE, 0); } TG_SB(); \
        TG_MFMA(BUF, 1); TG_SB();                                                          \
        if (RD) { TG_RDX(1 - BUF, NS, NBASE, 1); } TG_DMA(3 * (KSI)); TG_SB();             \
        TG_MFMA(BUF, 2); TG_SB();                                                          \
        if (RD) { TG_RDY(1 - BUF, NS, NBASE, 1); } TG_SB();                                \
        TG_MFMA(BUF, 3); TG_SB();                                                          \
        if (RD) { TG_RDY(1 - BUF, NS, NBASE, 2); } TG_SB();                                \
        TG_MFMA(BUF, 4); TG_SB();                                                          \
        if (RD) { TG_RDY(1 - BUF, NS, NBASE, 3); } TG_DMA(3 * (KSI) + 1); TG_SB();         \
        TG_MFMA(BUF, 5); TG_SB();                                                          \
        TG_MFMA(BUF, 6); TG_SB();                                                          \
        TG_DMA(3 * (KSI) + 2); TG_SB();                                                    \
        TG_MFMA(BUF, 7); TG_SB();                                                          \
    } while (0)
    {   // stage 0 landed: the other prologue stages (PPW instructions each) may still be in flight
        const int younger = (nsteps < RING - 1 ? nsteps : RING - 1) - 1;
        if (younger >= 3) wait_vmcnt<3 * PPW>();
        else if (younger == 2) wait_vmcnt<2 * PPW>();
        else if (younger == 1) wait_vmcnt<PPW>();
        else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    TG_SB();
#pragma unroll
    for (int a = 0; a < 4; ++a) TG_RDY(0, 0, lds_base, a);
#pragma unroll
    for (int b = 0; b < 2; ++b) TG_RDX(0, 0, lds_base, b);
    int c_slot = 0;
    for (int i = 0; i < nsteps; ++i) {
        const bool more = it < nsteps;          // stage i + RING - 1 exists: issue it during this stage
        const bool next = i + 1 < nsteps;
        const unsigned sbase = lds_base + c_slot * STAGE;
        c_slot = c_slot == RING - 1 ? 0 : c_slot + 1;
        const unsigned nbase = lds_base + c_slot * STAGE;
        if (do_bias) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int qq = 0; qq < ROWS / 16; ++qq) {
                    const u32x4 c8 = *(const u32x4*)(smem_g + (sbase - lds_base) + h * PANEL + b_off + qq * 16 * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum[h * 8 + 2 * e] += __uint_as_float(c8[e] << 16);
                        bsum[h * 8 + 2 * e + 1] += __uint_as_float(c8[e] & 0xffff0000u);
                    }
                }
        }
        TG_SB();
        TG_KSTEP(0, 0, true, 1, sbase);
        if constexpr (KS == 4) {
            TG_KSTEP(1, 1, true, 2, sbase);
            TG_KSTEP(0, 2, true, 3, sbase);
        }
        // every read of this stage's slot has completed (the slot is refilled during the next stage); stage
        // i + 1 must have landed in all waves before its fragments are read.  Younger than its DMA
        // instructions: the whole stages i + 2 .. i + RING - 2 and PART instructions of stage i + RING - 1.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (next) {
            if (more) wait_vmcnt<(RING - 3) * PPW + PART>();
            else {
                const int full = nsteps - 2 - i;    // stages i + 2 .. nsteps - 1, all of them issued completely
                if (RING >= 5 && full >= 2) wait_vmcnt<2 * PPW>();
                else if (RING >= 4 && full >= 1) wait_vmcnt<PPW>();
                else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        TG_SB();
        TG_KSTEP(1, KS - 1, next, 0, nbase);
        if (more) advance();
    }
#undef TG_KSTEP
#undef TG_DMA
#undef TG_SB
#undef TG_MFMA
#undef TG_RDX
#undef TG_RDY
    // accumulator: row = n (registers), col = k (lane): 32 consecutive k per register -> 128-B segments
    float* __restrict__ dW = pr.dW;
    const int ldw = pr.ldw;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
            const int kc = k0 + wk * 64 + ki * 32 + (lane & 31);
            if (kc >= K) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 128 + ni * 32 + acc_row(e, lane);
                if (n >= N) continue;
                float* dst = dW + (size_t)n * ldw + kc;
                if (use_atomic) atomicAdd(dst, acc[ni][ki][e]);
                else *dst = acc[ni][ki][e];
            }
        }
    if (do_bias) {   // 16 row phases -> one sum per column, through the (now idle) ring memory
        __syncthreads();
        float* red = (float*)smem_g;
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(tid >> 4) * 256 + (e >> 3) * 128 + (tid & 15) * 8 + (e & 7)] = bsum[e];
        __syncthreads();
        if (n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int ph = 0; ph < 16; ++ph) t += red[ph * 256 + tid];
            atomicAdd(pr.dbias + n0 + tid, t);
        }
    }
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* p, int rows, int cols, int ld) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cols, c = i - r * cols;
        p[r * ld + c] = 0.f;
    }
}

// Deterministic mode, second pass: dW[n][k] (+)= slab[0][n][k] + slab[1][n][k] + ... in split order; the same for the
// bias slabs (dbias[n] += ...).  blockIdx.y == 1 handles the bias.
__global__ __launch_bounds__(256) void tn_slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int N, int K, float* __restrict__ dW, int ldw,
                                                             int accumulate, const float* __restrict__ bslab, float* __restrict__ dbias) {
    if (blockIdx.y == 1) {
        if (!dbias) return;
        for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
            float t = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) t += bslab[(size_t)sp * N + n];
            dbias[n] += t;
        }
        return;
    }
    const size_t total4 = (size_t)N * K / 4, plane = (size_t)N * K;   // K % 8 == 0
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 t = *(const f32x4*)(slab + i * 4);
        for (int sp = 1; sp < nsplit; ++sp) {
            const f32x4 v = *(const f32x4*)(slab + sp * plane + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += v[e];
        }
        const size_t n = (i * 4) / K, k = i * 4 - n * K;
        float* dst = dW + n * ldw + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = accumulate ? dst[e] + t[e] : t[e];
    }
}

static int tn_splits(int M, int N, int K) {
    const int tiles = ceil_div(N, 128) * ceil_div(K, 128);
    int s = ceil_div(256, tiles);            // ~one workgroup per CU: every extra split adds N*K*4 B of atomics
    const int max_s = ceil_div(M, 4 * TM);   // at least 4 reduction tiles per workgroup
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

extern "C" int asr_gemm_nt_bf16(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                int act, void* stream) {
    if (!A || !W || !C) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    if (K % 8 || lda % 8 || ldb % 8 || ldc % 4 || lda < K || ldb < K || ldc < N) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: K, lda, ldb must be multiples of 8 and ldc of 4 (K=%d lda=%d ldb=%d ldc=%d)", K, lda, ldb, ldc);
    if ((((uintptr_t)A | (uintptr_t)W) % 16) || ((uintptr_t)C % 8) || (res && (uintptr_t)res % 8) || (bias && (uintptr_t)bias % 16)) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    if (act != ASR_ACT_RELU && act != ASR_ACT_NONE && act != ASR_ACT_RELU_MASK) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: unknown activation %d", act);
    // every projection / input gradient of the training step takes the persistent loader / consumer kernel (gemm_nt_spec_kernel: >= 2 k-steps,
    // N % 8 == 0, 16-byte aligned rows); what is left (K < 128, odd leading dimensions) goes to the register-staged 128 x 128 kernel below
    const bool rows16 = N % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)C % 16) == 0;
    if (act == ASR_ACT_RELU_MASK) {   // C = (A W^T) where res > 0, else 0 (the input gradient through a ReLU; no bias: gradients have none)
        if (!res || (uintptr_t)res % 16 || K % DBK || K < 2 * PBK || !rows16)
            ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: ASR_ACT_RELU_MASK needs the mask in `res`, K %% 64 == 0, K >= 128, N, ldc %% 8 == 0 and 16-byte aligned pointers");
        if (bias) ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: ASR_ACT_RELU_MASK takes no bias");
        launch_nt_spec<ASR_ACT_RELU_MASK>((const bf16_t*)A, (const bf16_t*)W, nullptr, (bf16_t*)C, M, N, K, lda, ldb, ldc, st, (const bf16_t*)res);
        ASR_CHECK_LAUNCH("asr_gemm_nt_bf16");
        return ASR_OK;
    }
    if (res && act == ASR_ACT_NONE && !bias && K % DBK == 0 && K >= 2 * PBK && rows16 && ((uintptr_t)res % 16) == 0) {
        // residual add in the store tail (res = C accumulates in place: every element is read and written by one lane)
        launch_nt_spec<NT_ACT_ADD_RES>((const bf16_t*)A, (const bf16_t*)W, nullptr, (bf16_t*)C, M, N, K, lda, ldb, ldc, st, (const bf16_t*)res);
        ASR_CHECK_LAUNCH("asr_gemm_nt_bf16");
        return ASR_OK;
    }
    if (K % DBK != 0 && K > PBK && !res && act == ASR_ACT_NONE && rows16) {      // >= 2 k-steps, the last one ragged (linear_in: K = 80; the CTC head's input gradient: K = 4232)
        static void* zero_page = nullptr;
        if (!zero_page && (hipGetSymbolAddress(&zero_page, HIP_SYMBOL(tn_zero_page)) != hipSuccess || !zero_page))
            ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: zero page symbol not found");
        launch_nt_spec<ASR_ACT_NONE, true>((const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldb, ldc, st, nullptr, zero_page);
        ASR_CHECK_LAUNCH("asr_gemm_nt_bf16");
        return ASR_OK;
    }
    if (K % DBK == 0 && K >= 2 * PBK && !res && rows16) {
        if (act == ASR_ACT_RELU) launch_nt_spec<ASR_ACT_RELU>((const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldb, ldc, st);
        else launch_nt_spec<ASR_ACT_NONE>((const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldb, ldc, st);
        ASR_CHECK_LAUNCH("asr_gemm_nt_bf16");
        return ASR_OK;
    }
    const int tiles_n = ceil_div(N, BN), tiles_m = ceil_div(M, BM);
    if (act == ASR_ACT_RELU)
        gemm_nt_kernel<ASR_ACT_RELU><<<tiles_n * tiles_m, 256, 0, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldb, ldc, tiles_n);
    else if (act == ASR_ACT_NONE)
        gemm_nt_kernel<ASR_ACT_NONE><<<tiles_n * tiles_m, 256, 0, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldb, ldc, tiles_n);
    else
        ASR_FAIL(ASR_EINVAL, "asr_gemm_nt_bf16: unknown activation %d", act);
    ASR_CHECK_LAUNCH("asr_gemm_nt_bf16");
    return ASR_OK;
}

extern "C" int asr_gemm_small_bf16(const void* A, const void* Bm, const float* bias, const void* mask, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                   int trans_b, int act, void* stream) {
    if (!A || !Bm || !C) ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    if (N % 8 || K % 8 || lda % 8 || ldb % 8 || ldc % 4 || lda < K || ldc < N || ldb < (trans_b ? N : K))
        ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: N, K, lda, ldb must be multiples of 8, ldc of 4 (N=%d K=%d lda=%d ldb=%d ldc=%d)", N, K, lda, ldb, ldc);
    if ((((uintptr_t)A | (uintptr_t)Bm) % 16) || ((uintptr_t)C % 8) || (mask && (uintptr_t)mask % 8) || (bias && (uintptr_t)bias % 4)) ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: misaligned pointer");
    if (act != ASR_ACT_NONE && act != ASR_ACT_RELU && act != ASR_ACT_RELU_MASK) ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: unknown activation %d", act);
    if ((act == ASR_ACT_RELU_MASK) != (mask != nullptr)) ASR_FAIL(ASR_EINVAL, "asr_gemm_small_bf16: ASR_ACT_RELU_MASK needs the activations in `mask` (and only it)");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = ceil_div(N, SM), grid = tiles_n * ceil_div(M, SM);
    const bf16_t *a = (const bf16_t*)A, *b = (const bf16_t*)Bm, *mk = (const bf16_t*)mask;
    bf16_t* c = (bf16_t*)C;
    static bool attr = false;
    if (!attr) {      // 135 / 141 KiB of dynamic LDS
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<true, ASR_ACT_RELU_MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<true>());
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<true, ASR_ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<true>());
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<true, ASR_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<true>());
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<false, ASR_ACT_RELU_MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<false>());
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<false, ASR_ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<false>());
        (void)hipFuncSetAttribute((const void*)gemm_small_kernel<false, ASR_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<false>());
        attr = true;
    }
#define SMALL(TB_, ACT_) asr_launch_armed(gemm_small_kernel<TB_, ACT_>, dim3(grid), dim3(256), small_lds_bytes<TB_>(), st, a, b, bias, mk, c, M, N, K, lda, ldb, ldc, tiles_n)      /* may carry an armed completion event */
    if (trans_b) {
        if (act == ASR_ACT_RELU_MASK) SMALL(true, ASR_ACT_RELU_MASK);
        else if (act == ASR_ACT_RELU) SMALL(true, ASR_ACT_RELU);
        else SMALL(true, ASR_ACT_NONE);
    } else {
        if (act == ASR_ACT_RELU_MASK) SMALL(false, ASR_ACT_RELU_MASK);
        else if (act == ASR_ACT_RELU) SMALL(false, ASR_ACT_RELU);
        else SMALL(false, ASR_ACT_NONE);
    }
#undef SMALL
    ASR_CHECK_LAUNCH("asr_gemm_small_bf16");
    return ASR_OK;
}

namespace {
// Staggered M-splits: lengths r0 + skew * s, s = 0 .. nsplit - 1, from ~0.75 to ~1.25 of the mean, all multiples of the 64-row stage,
// covering M.  Every split ends by adding its 128 x 128 tile to memory with fp32 atomics (256 workgroups x 64 KiB = 16 MB per launch at
// the chip's 1.3 TB/s atomic rate): splits that finish one after the other put that traffic under the others' compute (step 3.326 ->
// 3.314 ms; a spread of 30 / 80 / 120 % of the mean: 3.328 / 3.376 / 3.459 against 3.305 at 50 %).
static void tn_stagger(int M, int rows_per_split, int nsplit, int* r0, int* skew) {
    *r0 = rows_per_split;
    *skew = 0;
    if (nsplit >= 3) {
        int sk = (int)((long long)rows_per_split * 50 / 100 / (nsplit - 1)) / TM * TM;
        const int tri = nsplit * (nsplit - 1) / 2;
        int r = ceil_div(ceil_div(M - sk * tri > 0 ? M - sk * tri : M, nsplit), TM) * TM;
        if (sk <= 0 || r < 4 * TM || r * (nsplit - 1) + sk * ((nsplit - 1) * (nsplit - 2) / 2) >= M) return;      // the last split must start inside the range
        *r0 = r;
        *skew = sk;
    }
}

// Problems over the same >= 4096 rows whose 128 x 128 tiles fill the chip with a few M-splits: ONE launch of the single-problem kernel's tile
// code (gemm_tn_multi_kernel).  Returns false when the group does not qualify (the caller then takes the 256 x 128-tile grouped kernel).
static bool tn_multi_launch(const asr_tn_problem* probs, int nprob, int accumulate, hipStream_t st, void* zero_page) {
    if (!asr_option(ASR_OPT_TN_MULTI) || nprob > TNM_MAX || probs[0].M < 4096) return false;
    TnMulti g;
    memset(&g, 0, sizeof(g));
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        const asr_tn_problem& q = probs[i];
        if (q.M != probs[0].M) return false;
        TnMultiProb& t = g.p[i];
        t.dY = (const bf16_t*)q.dY; t.X = (const bf16_t*)q.X; t.dW = q.dW; t.dbias = q.dbias;
        t.N = q.N; t.K = q.K; t.ldy = q.ldy; t.ldx = q.ldx; t.ldw = q.ldw;
        t.tiles_k = ceil_div(q.K, 128);
        t.tiles = ceil_div(q.N, 128) * t.tiles_k;
        tiles += t.tiles;
    }
    const int cus = cu_count(), M = probs[0].M;
    int splits = cus / tiles;
    const int max_s = ceil_div(M, 4 * TM);   // at least 4 reduction stages per workgroup
    if (splits > max_s) splits = max_s;
    if (splits < 1) splits = 1;
    if (tiles * splits * 5 < cus * 4 && splits < max_s) return false;      // would leave > 20 % of the CUs without a workgroup
    const int rows_per_split = ceil_div(ceil_div(M, splits), TM) * TM;
    g.nsplit = ceil_div(M, rows_per_split);
    g.nprob = nprob;
    g.M = M;
    tn_stagger(M, rows_per_split, g.nsplit, &g.r0, &g.skew);
    int begin = 0;
    for (int i = 0; i < nprob; ++i) {
        g.p[i].block_begin = begin;
        begin += g.p[i].tiles * g.nsplit;
        if (!accumulate) {      // the tile code adds with atomics
            const size_t total = (size_t)g.p[i].N * g.p[i].K;
            const int zg = (int)((total + 255) / 256);
            zero_f32_kernel<<<zg < 1024 ? zg : 1024, 256, 0, st>>>(g.p[i].dW, g.p[i].N, g.p[i].K, g.p[i].ldw);
        }
    }
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TSTAGE); attr = true; }
    gemm_tn_multi_kernel<3><<<begin, 256, 3 * TSTAGE, st>>>(g, zero_page);
    return true;
}
}  // namespace

extern "C" int asr_gemm_tn_grouped_bf16(const asr_tn_problem* probs, int nprob, int accumulate, void* stream) {
    if (!probs || nprob <= 0 || nprob > TG_MAX) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: 1..%d problems per call (got %d)", TG_MAX, nprob);
    if (asr_deterministic()) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: the grouped kernel combines splits with fp32 atomics - in deterministic mode call asr_gemm_tn_bias_bf16 per problem");
    hipStream_t st = (hipStream_t)stream;
    TnGroup g;
    memset(&g, 0, sizeof(g));
    g.nprob = nprob;
    long long work = 0;
    int max_m = 0;
    for (int i = 0; i < nprob; ++i) {
        const asr_tn_problem& q = probs[i];
        if (!q.dY || !q.X || !q.dW) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: null pointer in problem %d", i);
        if (q.M <= 0 || q.N <= 0 || q.K <= 0) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: bad shape M=%d N=%d K=%d in problem %d", q.M, q.N, q.K, i);
        if (q.N % 8 || q.K % 8 || q.ldy % 8 || q.ldx % 8 || q.ldy < q.N || q.ldx < q.K || q.ldw < q.K)
            ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: N, K, ldy, ldx must be multiples of 8 (problem %d: N=%d K=%d ldy=%d ldx=%d)", i, q.N, q.K, q.ldy, q.ldx);
        if (((uintptr_t)q.dY | (uintptr_t)q.X) % 16) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: misaligned pointer in problem %d", i);
        TnProb& t = g.p[i];
        t.dY = (const bf16_t*)q.dY; t.X = (const bf16_t*)q.X; t.dW = q.dW; t.dbias = q.dbias;
        t.M = q.M; t.N = q.N; t.K = q.K; t.ldy = q.ldy; t.ldx = q.ldx; t.ldw = q.ldw;
        t.tiles_k = ceil_div(q.K, 128);
        t.tiles = ceil_div(q.N, 256) * t.tiles_k;
        work += (long long)t.tiles * q.M;
        if (q.M > max_m) max_m = q.M;
    }
    static void* zero_page = nullptr;
    if (!zero_page && (hipGetSymbolAddress(&zero_page, HIP_SYMBOL(tn_zero_page)) != hipSuccess || !zero_page))
        ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_grouped_bf16: zero page symbol not found");
    if (tn_multi_launch(probs, nprob, accumulate, st, zero_page)) {
        ASR_CHECK_LAUNCH("asr_gemm_tn_grouped_bf16");
        return ASR_OK;
    }
    // rows per workgroup R (one value for the whole group = balanced work): the smallest multiple of 64
    // with sum_p tiles_p * ceil(M_p / R) <= #CUs, i.e. one round of one workgroup per CU; >= 4 stages each
    const int cus = cu_count();
    int R = (int)((work + cus - 1) / cus);
    R = (R + TM - 1) / TM * TM;
    if (R < 4 * TM) R = 4 * TM;
    int blocks = 0;
    for (;; R += TM) {
        blocks = 0;
        for (int i = 0; i < nprob; ++i) blocks += g.p[i].tiles * ceil_div(g.p[i].M, R);
        if (blocks <= cus || R >= max_m) break;
    }
    static const int tn_noatomic = unsafe_debug_env("ASR_GEMM_TN_NOATOMIC");   // timing experiments only (wrong results): debug builds only
    int begin = 0;
    for (int i = 0; i < nprob; ++i) {
        TnProb& t = g.p[i];
        t.rows_per_split = R < t.M ? R : t.M;
        t.block_begin = begin;
        const int nsplit = ceil_div(t.M, R);
        begin += t.tiles * nsplit;
        if (nsplit > 1 && !accumulate) {
            size_t total = (size_t)t.N * t.K;
            int zg = (int)((total + 255) / 256);
            zero_f32_kernel<<<zg < 1024 ? zg : 1024, 256, 0, st>>>(t.dW, t.N, t.K, t.ldw);
        }
    }
    // stage rows x ring: 64 x 3 = 144 KiB is the fastest alone (0.67 PFLOP/s on a config-2 layer), but this GEMM runs
    // beside the main stream, whose attention / LayerNorm workgroups need the rest of the CU's LDS
    static const int dbg = unsafe_debug_env("ASR_GEMM_TNG_DBG");
#define TNG_LAUNCH(ROWS_, RING_, DBG_)                                                                                                   \
    do {                                                                                                                                   \
        static bool attr = false;                                                                                                          \
        constexpr int lds = RING_ * 3 * ROWS_ * 256;                                                                                       \
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel<ROWS_, RING_, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
        gemm_tn_grouped_kernel<ROWS_, RING_, DBG_><<<begin, 256, lds, st>>>(g, accumulate, tn_noatomic, zero_page);                        \
    } while (0)
    (void)dbg;
    TNG_LAUNCH(32, 4, 0);      // 32-row stages x 4 = 96 KiB (64 x 3 = 144 KiB is the fastest alone, 0.67 PFLOP/s on a config-2 layer, but excludes the main stream from the CU)
#undef TNG_LAUNCH
    ASR_CHECK_LAUNCH("asr_gemm_tn_grouped_bf16");
    return ASR_OK;
}

namespace {
struct TnPlan { int ring, rows_per_split, nsplit; };
// M-splits: every split adds N*K*4 bytes of atomics, every workgroup beyond what is resident at once adds a whole second round.  One 4-wave
// workgroup per CU with a 3-stage ring (96 KiB) when the tiles x splits fill >= 80 % of the CUs that way, else two per CU with 2-stage rings.
// (Tried and in the git history: an 8-wave form - 12 % faster alone, step 2 % slower: its waves take issue slots from the main stream's kernels
// beside it; loader / consumer waves - same result; a 4-stage ring of 128 KiB - 1-5 % faster alone, step 1.4 % slower; fewer / more splits.)
static TnPlan tn_plan(int M, int N, int K) {
    const int tiles = ceil_div(N, 128) * ceil_div(K, 128);
    const int cus = cu_count();
    const int max_s = ceil_div(M, 4 * TM);   // at least 4 reduction stages per workgroup
    TnPlan pl;
    int splits;
    const int s_floor = cus / tiles > 0 ? cus / tiles : 1;
    if (tiles * s_floor * 5 >= cus * 4) { splits = s_floor; pl.ring = 3; }
    else { splits = ceil_div(2 * cus, tiles) > 1 ? (2 * cus) / tiles : 1; pl.ring = 2; }
    if (splits > max_s) splits = max_s;
    if (splits < 1) splits = 1;
    pl.rows_per_split = ceil_div(ceil_div(M, splits), TM) * TM;
    pl.nsplit = ceil_div(M, pl.rows_per_split);
    return pl;
}
}  // namespace

extern "C" size_t asr_gemm_tn_workspace_bytes(int M, int N, int K) {
    // default mode: partial tiles are combined with fp32 atomics, no scratch.  Deterministic mode: one (N, K) fp32
    // slab and one N-vector (bias gradient) per M-split.
    if (!asr_deterministic() || M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan pl = tn_plan(M, N, K);
    return ((size_t)pl.nsplit * N * K + (size_t)pl.nsplit * N) * sizeof(float);
}

extern "C" int asr_gemm_tn_bias_bf16(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, int ldy, int ldx, int ldw,
                                     int accumulate, void* ws, size_t ws_bytes, void* stream);

extern "C" int asr_gemm_tn_bf16(const void* dY, const void* X, float* dW, int M, int N, int K, int ldy, int ldx, int ldw, int accumulate, void* ws,
                                size_t ws_bytes, void* stream) {
    return asr_gemm_tn_bias_bf16(dY, X, dW, nullptr, M, N, K, ldy, ldx, ldw, accumulate, ws, ws_bytes, stream);
}

extern "C" int asr_gemm_tn_bias_bf16(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, int ldy, int ldx, int ldw,
                                     int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!dY || !X || !dW) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_bf16: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    if (N % 8 || K % 8 || ldy % 8 || ldx % 8 || ldy < N || ldx < K || ldw < K) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_bf16: N, K, ldy, ldx must be multiples of 8 (N=%d K=%d ldy=%d ldx=%d)", N, K, ldy, ldx);
    if (((uintptr_t)dY | (uintptr_t)X) % 16) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_bf16: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = ceil_div(N, 128), tiles_k = ceil_div(K, 128), tiles = tiles_n * tiles_k;
    static const int tn_noatomic = unsafe_debug_env("ASR_GEMM_TN_NOATOMIC");   // timing experiments only (wrong results): debug builds only
    const TnPlan pl = tn_plan(M, N, K);
    const int ring = pl.ring, rows_per_split = pl.rows_per_split, nsplit = pl.nsplit;
    // Deterministic mode: no atomics - each split stores its partial tile into its own slab of the workspace and
    // tn_slab_reduce_kernel adds the slabs in split order (also for a single split when accumulating into dW).
    const bool det = asr_deterministic() != 0;
    float* dW_k = dW;
    float* dbias_k = dbias;
    int ldw_k = ldw, bias_split_stride = 0;
    size_t split_stride = 0;
    if (det) {
        const size_t need = asr_gemm_tn_workspace_bytes(M, N, K);
        if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) ASR_FAIL(ASR_EWORKSPACE, "asr_gemm_tn_bf16: deterministic mode needs a 16-byte aligned workspace of %zu bytes (got %zu)", need, ws_bytes);
        dW_k = (float*)ws;
        ldw_k = K;
        split_stride = (size_t)N * K;
        if (dbias) { dbias_k = (float*)ws + (size_t)nsplit * N * K; bias_split_stride = N; }
    }
    const int use_atomic = (tn_noatomic || det) ? 0 : ((nsplit > 1) || accumulate);
    if (nsplit > 1 && !accumulate && !det) {
        size_t total = (size_t)N * K;
        int g = (int)((total + 255) / 256);
        zero_f32_kernel<<<g < 1024 ? g : 1024, 256, 0, st>>>(dW, N, K, ldw);
    }
    const int grid = tiles * nsplit;
    int sk_r0, sk_skew;      // staggered M-splits (tn_stagger)
    tn_stagger(M, rows_per_split, nsplit, &sk_r0, &sk_skew);
    static void* zero_page = nullptr;
    if (!zero_page) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TSTAGE);
        (void)hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TSTAGE);
        if (hipGetSymbolAddress(&zero_page, HIP_SYMBOL(tn_zero_page)) != hipSuccess || !zero_page) ASR_FAIL(ASR_EINVAL, "asr_gemm_tn_bf16: zero page symbol not found");
    }
    // the ring is 3 stages = 96 KiB (one workgroup per CU) or 2 stages (two per CU): this kernel runs BESIDE the main stream, whose attention /
    // LayerNorm workgroups need the rest of the CU's LDS (a 16-KiB pad on top of 128 KiB cost the training step 6 %, 30 KiB 12 %)
    if (ring == 3)
        gemm_tn_dma_kernel<3><<<grid, 256, 3 * TSTAGE, st>>>((const bf16_t*)dY, (const bf16_t*)X, dW_k, M, N, K, ldy, ldx, ldw_k, tiles_k, tiles_n, sk_r0, use_atomic, zero_page, dbias_k, split_stride, bias_split_stride, sk_skew);
    else
        gemm_tn_dma_kernel<2><<<grid, 256, 2 * TSTAGE, st>>>((const bf16_t*)dY, (const bf16_t*)X, dW_k, M, N, K, ldy, ldx, ldw_k, tiles_k, tiles_n, sk_r0, use_atomic, zero_page, dbias_k, split_stride, bias_split_stride, sk_skew);
    if (det) {
        const size_t total4 = (size_t)N * K / 4;
        int g = (int)((total4 + 255) / 256);
        tn_slab_reduce_kernel<<<dim3(g < 1024 ? g : 1024, dbias ? 2 : 1), 256, 0, st>>>((const float*)ws, nsplit, N, K, dW, ldw, accumulate, dbias_k, dbias);
    }
    ASR_CHECK_LAUNCH("asr_gemm_tn_bf16");
    return ASR_OK;
}
