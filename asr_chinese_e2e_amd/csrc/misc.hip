// Error reporting, element-wise glue, column sums, decoder target preparation, embedding,
// and the fused clip + Noam + Adam optimizer over the flat parameter buffer.
#include <stdarg.h>
#include <stdlib.h>

#include "asr_common.h"

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int asr_abi_version(void) { return ASR_ABI_VERSION; }

// Deterministic mode: every cross-workgroup floating-point reduction runs in a fixed order (weight-gradient split
// sums through partial slabs + an ordered second pass instead of fp32 atomics, single-workgroup column-sum
// finalisation, serial embedding scatter).  Slower; meant for reproducibility runs and equivalence tests.
static int g_deterministic = -1;
int asr_deterministic(void) {
    if (g_deterministic < 0) {
        const char* e = getenv("ASR_DETERMINISTIC");
        g_deterministic = (e && atoi(e) != 0) ? 1 : 0;
    }
    return g_deterministic;
}
extern "C" int asr_get_deterministic(void) { return asr_deterministic(); }
extern "C" int asr_set_deterministic(int on) {
    const int old = asr_deterministic();
    g_deterministic = on ? 1 : 0;
    return old;
}

// ---- tuning options: process-wide integer switches under which every value gives correct results (asr_set_option).  ABI 8 keeps one,
// "cu_limit" (gemm.hip: cu_count); the kernel-variant switches of rounds 2 - 3 left with the variants.
// "tn_multi" (round 5; default 1): asr_gemm_tn_grouped_bf16 runs problems over the same >= 4096 rows on the 128 x 128-tile kernel (one launch,
// fewer M-splits); 0 = the 256 x 128-tile grouped kernel for every group (A/B timing inside one process).
// "sdpa_pair" (round 5; default 0): 1 = the attention forward without causal / band mask and without dropout on the kernel that takes both 32-query
// blocks of a wave through ONE pass over the key tiles (sdpa_fwd_pair_bf16_kernel: bit-identical results, measured 33.5 vs 31.6 us - opt-in).
static const char* const g_opt_names[ASR_OPT_COUNT] = {"cu_limit", "tn_multi", "sdpa_pair"};
static int g_opt_val[ASR_OPT_COUNT] = {0, 1, 0};
static void opt_init() {}
int asr_option(int key) {
    opt_init();
    return (key >= 0 && key < ASR_OPT_COUNT) ? g_opt_val[key] : 0;
}
int asr_option_set(int key, int value) {
    opt_init();
    if (key < 0 || key >= ASR_OPT_COUNT) return 0;
    const int old = g_opt_val[key];
    g_opt_val[key] = value;
    return old;
}
extern "C" int asr_set_option(const char* name, int value, int* previous) {
    opt_init();
    if (!name) ASR_FAIL(ASR_EINVAL, "asr_set_option: null name");
    for (int i = 0; i < ASR_OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) {
            if (previous) *previous = g_opt_val[i];
            g_opt_val[i] = value;
            return ASR_OK;
        }
    ASR_FAIL(ASR_EINVAL, "asr_set_option: unknown option '%s'", name);
}

extern "C" int asr_last_error(char* buf, size_t n) {
    if (buf && n) {
        strncpy(buf, g_err, n - 1);
        buf[n - 1] = 0;
    }
    return (int)strlen(g_err);
}

// ---- stream fork: "everything queued on `to` from now on runs after what is queued on `from` so far" ----------------------
// One call = hipEventRecord + hipStreamWaitEvent on an event from a process-wide pool (timing disabled).  The engine forks
// to its side streams ~50 times per step; through torch.cuda.Event objects that costs the host ~12 us each (measured),
// through this entry point ~2 us.  Re-using an event is safe: a wait refers to the record that preceded it at call time.
// Not for use during stream capture (captured events belong to their graph): the caller keeps torch events there.
// Events that only order streams of THIS device: a device-scope release when the event is recorded / when the kernel it is bound to
// completes (hipEventReleaseToDevice).  The default is a system-scope release - the L2 written back so that the host or another device
// could look at the data - and the main queue started its next kernel ~5 us late after every kernel that carried such an event
// (step timeline, round 3).
static unsigned fork_event_flags() { return hipEventDisableTiming | hipEventReleaseToDevice; }
static hipEvent_t g_fork_events[256];
static unsigned g_fork_next = 0;
static bool g_fork_init = false;
extern "C" int asr_stream_fork(void* from_stream, void* to_stream) {
    if (!g_fork_init) {
        for (int i = 0; i < 256; ++i)
            if (hipEventCreateWithFlags(&g_fork_events[i], fork_event_flags()) != hipSuccess) ASR_FAIL(ASR_EHIP, "asr_stream_fork: hipEventCreateWithFlags failed");
        g_fork_init = true;
    }
    hipEvent_t ev = g_fork_events[g_fork_next++ & 255u];
    hipError_t e = hipEventRecord(ev, (hipStream_t)from_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to_stream, ev, 0);
    if (e != hipSuccess) ASR_FAIL(ASR_EHIP, "asr_stream_fork: %s", hipGetErrorString(e));
    return ASR_OK;
}

// Armed hand-over (see asr_launch_armed in asr_common.h): one pending (from, to) pair per process - the engine arms, calls ONE entry
// point, and checks that the arm was taken.
static hipStream_t g_arm_from = nullptr, g_arm_to = nullptr;
static bool g_arm_set = false;
extern "C" int asr_stream_arm(void* from_stream, void* to_stream) {
    if (!g_fork_init) {
        for (int i = 0; i < 256; ++i)
            if (hipEventCreateWithFlags(&g_fork_events[i], fork_event_flags()) != hipSuccess) ASR_FAIL(ASR_EHIP, "asr_stream_arm: hipEventCreateWithFlags failed");
        g_fork_init = true;
    }
    g_arm_from = (hipStream_t)from_stream;
    g_arm_to = (hipStream_t)to_stream;
    g_arm_set = true;
    return ASR_OK;
}
// 1 while an arm is still waiting for its kernel (the entry point called since had no armed launch, or launched on another stream):
// the caller then falls back to asr_stream_fork.  Clears the arm.
extern "C" int asr_stream_arm_pending(void) {
    const int p = g_arm_set ? 1 : 0;
    g_arm_set = false;
    return p;
}
bool asr_arm_take(hipStream_t st, hipStream_t* to, hipEvent_t* ev) {
    if (!g_arm_set || st != g_arm_from) return false;
    g_arm_set = false;
    *to = g_arm_to;
    *ev = g_fork_events[g_fork_next++ & 255u];
    return true;
}

// A stream of the HIP runtime this library is linked against (the one torch loaded first), at the lowest (priority < 0),
// default (0) or highest (> 0) priority the device offers.
extern "C" int asr_stream_create(int priority, void** out_stream) {
    if (!out_stream) ASR_FAIL(ASR_EINVAL, "asr_stream_create: null output");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);      // numerically: least >= greatest
    if (e != hipSuccess) ASR_FAIL(ASR_EHIP, "asr_stream_create: %s", hipGetErrorString(e));
    const int prio = priority < 0 ? least : priority > 0 ? greatest : 0;
    hipStream_t st = nullptr;
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio);
    if (e != hipSuccess || !st) ASR_FAIL(ASR_EHIP, "asr_stream_create: %s", hipGetErrorString(e));
    *out_stream = (void*)st;
    return ASR_OK;
}

namespace {

constexpr int EW_BLOCK = 256;
ASR_FULL_WAVES(EW_BLOCK);
static int ew_grid(size_t nvec) {
    size_t g = (nvec + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(g < 2048 ? (g ? g : 1) : 2048);  // cap + grid-stride (guide G11)
}

template <typename T> __global__ __launch_bounds__(EW_BLOCK) void relu_fwd_kernel(T* x, size_t n8, size_t n) {
    for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < n8; i += (size_t)gridDim.x * EW_BLOCK) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        store8<T>(x + i * 8, v);
    }
    if (blockIdx.x == 0) {  // tail
        size_t i = n8 * 8 + threadIdx.x;
        if (i < n) x[i] = from_f32<T>(fmaxf(to_f32<T>(x[i]), 0.f));
    }
}

template <typename S, typename D> __global__ __launch_bounds__(EW_BLOCK) void cast_kernel(const S* s, D* d, size_t n8, size_t n) {
    for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < n8; i += (size_t)gridDim.x * EW_BLOCK) {
        float v[8];
        load8<S>(s + i * 8, v);
        store8<D>(d + i * 8, v);
    }
    if (blockIdx.x == 0) {
        size_t i = n8 * 8 + threadIdx.x;
        if (i < n) d[i] = from_f32<D>(to_f32<S>(s[i]));
    }
}

// Column sums: a block covers 64 columns-groups of CW columns and strides over rows; each wave
// keeps CW per-lane accumulators, partials go to ws[slot][cols].
constexpr int CS_WAVES = 4;
template <typename T, bool RELU_BWD>
__global__ __launch_bounds__(CS_WAVES* WAVE) void colsum_partial_kernel(T* __restrict__ x, const T* __restrict__ a,
                                                                         float* __restrict__ ws, int rows, int cols, int ld,
                                                                         int row_slots) {
    // grid.x = column tiles of 256 (4 per lane), grid.y = row_slots; slot = blockIdx.y*CS_WAVES + wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + lane * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool full = (c0 + 3 < cols) && (ld % 4 == 0);
    const int rstride = row_slots * CS_WAVES;
    int r = blockIdx.y * CS_WAVES + w;
    if (full) {   // 4 rows in flight per lane: the pass is latency-bound with one load per iteration
        for (; r + 3 * rstride < rows; r += 4 * rstride) {
            f32x4 v[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = load4<T>(x + (size_t)(r + u * rstride) * ld + c0);
                if constexpr (RELU_BWD) av[u] = load4<T>(a + (size_t)(r + u * rstride) * ld + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (RELU_BWD) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = av[u][j] > 0.f ? v[u][j] : 0.f;
                    store4<T>(x + (size_t)(r + u * rstride) * ld + c0, v[u]);
                }
                acc += v[u];
            }
        }
    }
    for (; r < rows; r += rstride) {
        T* p = x + (size_t)r * ld + c0;
        if (full) {
            f32x4 v = load4<T>(p);
            if constexpr (RELU_BWD) {
                f32x4 av = load4<T>(a + (size_t)r * ld + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = av[j] > 0.f ? v[j] : 0.f;
                store4<T>(p, v);
            }
            acc += v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < cols) {
                    float v = to_f32<T>(p[j]);
                    if constexpr (RELU_BWD) {
                        v = to_f32<T>(a[(size_t)r * ld + c0 + j]) > 0.f ? v : 0.f;
                        p[j] = from_f32<T>(v);
                    }
                    acc[j] += v;
                }
        }
    }
    __shared__ __attribute__((aligned(16))) float sred[CS_WAVES][256];
    *(f32x4*)&sred[w][lane * 4] = acc;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) ws[(size_t)blockIdx.y * cols + c] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
}

static int cs_row_slots(int rows, int cols) {
    int ctiles = ceil_div(cols, 256);
    int want = 2048 / ctiles;  // ~2048 workgroups in total
    int maxs = ceil_div(rows, CS_WAVES);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    return want;
}

// ---- decoder target preparation --------------------------------------------------------------
__global__ void dec_preprocess_kernel(const int64_t* __restrict__ tgt, int32_t* ys_in, int32_t* ys_out,
                                      int32_t* labels32, int32_t* dec_len, int32_t* lab_len, float* n_valid,
                                      int B, int Lmax, int sos, int eos, const int64_t* __restrict__ len_a, int32_t* len32_a,
                                      const int64_t* __restrict__ len_b, int32_t* len32_b) {
    // one thread per utterance (B is small, rows are short); thread 0 also sums n_valid
    __shared__ int cnt[1024];
    const int b = threadIdx.x;
    int n = 0;
    if (b < B) {
        // the batch's int64 length vectors (wave_len, tgt_len of the reference's batch contract) as the int32 the kernels take: two
        // elementwise launches of ~6 us each in front of every step otherwise
        if (len_a) len32_a[b] = (int32_t)len_a[b];
        if (len_b) len32_b[b] = (int32_t)len_b[b];
        const int To = Lmax + 1;
        int32_t* yi = ys_in + (size_t)b * To;
        int32_t* yo = ys_out + (size_t)b * To;
        int32_t* lb = labels32 + (size_t)b * Lmax;
        yi[0] = sos;
        for (int j = 0; j < Lmax; ++j) {
            int v = (int)tgt[(size_t)b * Lmax + j];
            if (v != 0) {  // y[y != IGNORE_ID]
                lb[n] = v;
                yi[1 + n] = v;
                yo[n] = v;
                ++n;
            }
        }
        yo[n] = eos;
        for (int j = n + 1; j < To; ++j) { yi[j] = eos; yo[j] = 0; }
        for (int j = n; j < Lmax; ++j) lb[j] = 0;
        dec_len[b] = n + 1;
        lab_len[b] = n;
    }
    cnt[threadIdx.x] = (b < B) ? n + 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < B; ++i) s += cnt[i];
        *n_valid = (float)s;
    }
}

// ---- embedding -------------------------------------------------------------------------------
template <typename T, typename W>
__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(const int32_t* __restrict__ ids, const W* __restrict__ emb,
                                                           const float* __restrict__ pe, T* __restrict__ y, float scale,
                                                           int rows, int To, int d, int V, uint32_t seed, uint32_t thr, float dscale) {
    const int row = blockIdx.x;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const int t = row % To;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float v = to_f32<W>(emb[(size_t)id * d + c]) * scale + pe[(size_t)t * d + c];
        if (thr) v = drop_keep_at((uint32_t)row * (uint32_t)d + c, seed, thr) ? v * dscale : 0.f;
        y[(size_t)row * d + c] = from_f32<T>(v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dy, const T* __restrict__ dy2,
                                                        float* __restrict__ demb, float scale, int d, int V, uint32_t seed, uint32_t thr,
                                                        float dscale) {
    const int row = blockIdx.x;
    const int id = ids[row];
    if (id < 0 || id >= V) return;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float g = (to_f32<T>(dy[(size_t)row * d + c]) + (dy2 ? to_f32<T>(dy2[(size_t)row * d + c]) : 0.f)) * scale;
        if (thr) g = drop_keep_at((uint32_t)row * (uint32_t)d + c, seed, thr) ? g * dscale : 0.f;
        if (g != 0.f) atomicAdd(&demb[(size_t)id * d + c], g);
    }
}

// Deterministic form: thread = one column, tokens in order (rows of the same id are added in a fixed order).
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_serial_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dy, const T* __restrict__ dy2,
                                                               float* demb, float scale, int rows, int d, int V, uint32_t seed, uint32_t thr,
                                                               float dscale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    for (int row = 0; row < rows; ++row) {
        const int id = ids[row];
        if (id < 0 || id >= V) continue;
        float g = (to_f32<T>(dy[(size_t)row * d + c]) + (dy2 ? to_f32<T>(dy2[(size_t)row * d + c]) : 0.f)) * scale;
        if (thr) g = drop_keep_at((uint32_t)row * (uint32_t)d + c, seed, thr) ? g * dscale : 0.f;
        if (g != 0.f) demb[(size_t)id * d + c] += g;
    }
}

// mask[r][c] for c < cols, element counter r * cols_pad + c (cols_pad = cols rounded up to even)
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* mask, size_t rows, int cols, uint32_t seed, uint32_t thr) {
    const size_t n = rows * (size_t)cols;
    const uint32_t cp = (uint32_t)((cols + 1) & ~1);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cols;
        const uint32_t c = (uint32_t)(i - r * cols);
        mask[i] = drop_keep_at((uint32_t)r * cp + c, seed, thr) ? 1 : 0;
    }
}

// ---- optimizer -------------------------------------------------------------------------------
constexpr int SS_BLOCK = 256;
ASR_FULL_WAVES(SS_BLOCK);
__global__ __launch_bounds__(SS_BLOCK) void sumsq_partial_kernel(const float* __restrict__ g, size_t n4, size_t n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)SS_BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * SS_BLOCK) {
        f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) {
        size_t i = n4 * 4 + threadIdx.x;
        if (i < n) s += g[i] * g[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ part, int P, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}
__device__ __forceinline__ void noam_update(int32_t* step, float* hyper, float model_size, float warmup, float factor, float lr_const, float b1, float b2) {
    const int s = *step + 1;
    *step = s;
    const double sd = (double)s;
    double lr = lr_const;
    if (warmup > 0.f) {
        double a = pow(sd, -0.5), b = sd * pow((double)warmup, -1.5);
        lr = (double)factor * pow((double)model_size, -0.5) * (a < b ? a : b);
    }
    hyper[0] = (float)lr;
    hyper[1] = (float)(1.0 - pow((double)b1, sd));
    hyper[2] = (float)sqrt(1.0 - pow((double)b2, sd));
    hyper[3] = (float)s;
}
// the finalizer of the squared gradient norm also advances the Noam schedule (asr_grad_sumsq_noam): one launch less between backward and Adam
__global__ __launch_bounds__(256) void sum_finalize_noam_kernel(const float* __restrict__ part, int P, float* out, int32_t* step, float* hyper, float model_size,
                                                                float warmup, float factor, float lr_const, float b1, float b2) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        *out = s;
        noam_update(step, hyper, model_size, warmup, factor, lr_const, b1, b2);
    }
}
static int ss_grid(size_t n) {
    size_t g = (n / 4 + SS_BLOCK - 1) / SS_BLOCK;
    return (int)(g < 1024 ? (g ? g : 1) : 1024);
}

__global__ void noam_hyper_kernel(int32_t* step, float* hyper, float model_size, float warmup, float factor,
                                  float lr_const, float b1, float b2) {
    noam_update(step, hyper, model_size, warmup, factor, lr_const, b1, b2);
}

__global__ __launch_bounds__(EW_BLOCK) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ p_lp, size_t n4, size_t n,
                                                        const float* __restrict__ hyper, const float* __restrict__ sumsq,
                                                        float max_norm, float b1, float b2, float eps, int write_clipped) {
    const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2];
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float norm = sqrtf(*sumsq);
        coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    }
    const float step_size = lr / bc1;
    write_clipped = write_clipped && coef != 1.f;      // nothing clipped: g already holds what would be written (4 of 34 B per parameter less)
    for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * EW_BLOCK) {
        f32x4 gv = *(const f32x4*)(g + i * 4), mv = *(const f32x4*)(m + i * 4), vv = *(const f32x4*)(v + i * 4), pv = *(const f32x4*)(p + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = gv[j] * coef;
            gv[j] = gg;
            mv[j] = b1 * mv[j] + (1.f - b1) * gg;
            vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
            pv[j] -= step_size * mv[j] / (sqrtf(vv[j]) / bc2s + eps);
        }
        *(f32x4*)(m + i * 4) = mv;
        *(f32x4*)(v + i * 4) = vv;
        *(f32x4*)(p + i * 4) = pv;
        if (write_clipped) *(f32x4*)(g + i * 4) = gv;
        if (p_lp) store4<bf16_t>(p_lp + i * 4, pv);
    }
    if (blockIdx.x == 0) {
        size_t i = n4 * 4 + threadIdx.x;
        if (i < n) {
            const float gg = g[i] * coef;
            const float mm = b1 * m[i] + (1.f - b1) * gg;
            const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
            const float pp = p[i] - step_size * mm / (sqrtf(vv) / bc2s + eps);
            m[i] = mm; v[i] = vv; p[i] = pp;
            if (write_clipped) g[i] = gg;
            if (p_lp) p_lp[i] = (bf16_t)pp;
        }
    }
}

__global__ __launch_bounds__(256) void loss_combine_kernel(const float* row_nll, int M, const float* n_valid, const float* nll, int B,
                                                           float w_ce, float w_ctc, float* loss) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f;
    if (row_nll) for (int i = threadIdx.x; i < M; i += 256) s += row_nll[i];
    if (nll) for (int i = threadIdx.x; i < B; i += 256) c += nll[i];
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        const float ce = row_nll ? s / *n_valid : 0.f;
        const float ctc = nll ? c / (float)B : 0.f;
        loss[0] = (row_nll ? w_ce * ce : 0.f) + (nll ? w_ctc * ctc : 0.f);
        loss[1] = ce;
        loss[2] = ctc;
    }
}

}  // namespace

extern "C" int asr_relu_fwd(void* x, size_t n, int dtype, void* stream) {
    if (!x) ASR_FAIL(ASR_EINVAL, "asr_relu_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_F32) relu_fwd_kernel<float><<<ew_grid(n / 8), EW_BLOCK, 0, st>>>((float*)x, n / 8, n);
    else if (dtype == ASR_BF16) relu_fwd_kernel<bf16_t><<<ew_grid(n / 8), EW_BLOCK, 0, st>>>((bf16_t*)x, n / 8, n);
    else ASR_FAIL(ASR_EDTYPE, "asr_relu_fwd: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_relu_fwd");
    return ASR_OK;
}

// Batched 2-D transposes out of one flat bf16 buffer: tile t of `tiles` = {element offset of the matrix in src, rows N, cols K,
// (tile row << 16) | tile col, element offset of the copy in dst, row stride of the copy};
// dst[dst_off + k * ldd + n] = src[src_off + n * K + k].
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int32_t* __restrict__ tiles) {
    __shared__ bf16_t tile[64][66];
    const int32_t* e = tiles + 6 * (size_t)blockIdx.x;
    const size_t off = (size_t)(uint32_t)e[0], doff = (size_t)(uint32_t)e[4];
    const int N = e[1], K = e[2], r0 = (e[3] >> 16) * 64, c0 = (e[3] & 0xffff) * 64, ldd = e[5];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < N && c < K) tile[ty + 4 * i][tx] = src[off + (size_t)r * K + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < N && c < K) dst[doff + (size_t)c * ldd + r] = tile[tx][ty + 4 * i];
    }
}

// The same with 16-byte global accesses (round 5: the element-wise form above moves the step's 100 MB of weight copies at 1.9 TB/s, beside the
// first kernels of the forward pass): rows of the source tile arrive as 16-byte pieces, the tile turns in LDS, rows of the copy leave as
// 16-byte pieces.  Per tile, falls back to element accesses when the matrix, its offsets or the copy's row stride are not multiples of 8
// elements (tile-uniform branch).
__global__ __launch_bounds__(256) void transpose_batched_vec_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int32_t* __restrict__ tiles) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];      // 144-byte rows: 16-byte pieces stay aligned
    const int32_t* e = tiles + 6 * (size_t)blockIdx.x;
    const size_t off = (size_t)(uint32_t)e[0], doff = (size_t)(uint32_t)e[4];
    const int N = e[1], K = e[2], r0 = (e[3] >> 16) * 64, c0 = (e[3] & 0xffff) * 64, ldd = e[5];
    const bool vec = ((N | K | ldd) % 8 == 0) && (off % 8 == 0) && (doff % 8 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (vec) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = threadIdx.x + 256 * c, r = id >> 3, ch = id & 7;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + r < N && c0 + ch * 8 < K) v = *(const u32x4*)(src + off + (size_t)(r0 + r) * K + c0 + ch * 8);
            *(u32x4*)(&tile[r][ch * 8]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = threadIdx.x + 256 * c, kk = id >> 3, ch = id & 7;      // row kk of the copy = column c0 + kk of the source
            if (c0 + kk < K && r0 + ch * 8 < N) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tile[ch * 8 + j][kk];
                *(bf16x8*)(dst + doff + (size_t)(c0 + kk) * ldd + r0 + ch * 8) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < N && c < K) tile[ty + 4 * i][tx] = src[off + (size_t)r * K + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < N && c < K) dst[doff + (size_t)c * ldd + r] = tile[tx][ty + 4 * i];
    }
}

extern "C" int asr_transpose_batched_bf16(const void* src, void* dst, const int32_t* tiles, int ntiles, void* stream) {
    if (!src || !dst || !tiles) ASR_FAIL(ASR_EINVAL, "asr_transpose_batched_bf16: null pointer");
    if (ntiles <= 0) ASR_FAIL(ASR_EINVAL, "asr_transpose_batched_bf16: ntiles=%d", ntiles);
    transpose_batched_vec_kernel<<<ntiles, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, (bf16_t*)dst, tiles);
    ASR_CHECK_LAUNCH("asr_transpose_batched_bf16");
    return ASR_OK;
}

extern "C" int asr_cast(const void* src, void* dst, size_t n, int sd, int dd, void* stream) {
    if (!src || !dst) ASR_FAIL(ASR_EINVAL, "asr_cast: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int g = ew_grid(n / 8);
    if (sd == ASR_F32 && dd == ASR_BF16) cast_kernel<float, bf16_t><<<g, EW_BLOCK, 0, st>>>((const float*)src, (bf16_t*)dst, n / 8, n);
    else if (sd == ASR_BF16 && dd == ASR_F32) cast_kernel<bf16_t, float><<<g, EW_BLOCK, 0, st>>>((const bf16_t*)src, (float*)dst, n / 8, n);
    else if (sd == ASR_F32 && dd == ASR_F32) cast_kernel<float, float><<<g, EW_BLOCK, 0, st>>>((const float*)src, (float*)dst, n / 8, n);
    else if (sd == ASR_BF16 && dd == ASR_BF16) cast_kernel<bf16_t, bf16_t><<<g, EW_BLOCK, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, n / 8, n);
    else ASR_FAIL(ASR_EDTYPE, "asr_cast: dtypes %d -> %d", sd, dd);
    ASR_CHECK_LAUNCH("asr_cast");
    return ASR_OK;
}

extern "C" size_t asr_colsum_workspace_bytes(int rows, int cols) {
    return (size_t)cs_row_slots(rows, cols) * cols * sizeof(float);
}

template <bool RELU_BWD>
static int colsum_impl(void* x, const void* a, float* out, void* ws, size_t ws_bytes, int rows, int cols, int ld,
                       int accumulate, int dtype, hipStream_t st, const char* name) {
    if (!x || (RELU_BWD && !a)) ASR_FAIL(ASR_EINVAL, "%s: null pointer", name);
    if (rows <= 0 || cols <= 0 || ld < cols) ASR_FAIL(ASR_EINVAL, "%s: bad shape rows=%d cols=%d ld=%d", name, rows, cols, ld);
    const bool want_sum = out != nullptr;
    if (want_sum && (!ws || ws_bytes < asr_colsum_workspace_bytes(rows, cols))) ASR_FAIL(ASR_EWORKSPACE, "%s: workspace too small", name);
    const int slots = cs_row_slots(rows, cols);
    dim3 grid(ceil_div(cols, 256), slots);
    // without `out` (relu_bwd only) the partials still need somewhere to go: require ws then too
    if (!ws || ws_bytes < asr_colsum_workspace_bytes(rows, cols)) ASR_FAIL(ASR_EWORKSPACE, "%s: workspace too small", name);
    if (dtype == ASR_F32) colsum_partial_kernel<float, RELU_BWD><<<grid, CS_WAVES * WAVE, 0, st>>>((float*)x, (const float*)a, (float*)ws, rows, cols, ld, slots);
    else if (dtype == ASR_BF16) colsum_partial_kernel<bf16_t, RELU_BWD><<<grid, CS_WAVES * WAVE, 0, st>>>((bf16_t*)x, (const bf16_t*)a, (float*)ws, rows, cols, ld, slots);
    else ASR_FAIL(ASR_EDTYPE, "%s: dtype %d", name, dtype);
    if (want_sum) colsum_finalize_kernel<<<dim3(ceil_div(cols, 32), accumulate && slots >= 1024 && !asr_deterministic() ? 4 : 1), 1024, 0, st>>>((const float*)ws, slots, (size_t)cols, cols, cols, out, nullptr, nullptr, accumulate);
    ASR_CHECK_LAUNCH(name);
    return ASR_OK;
}

extern "C" int asr_colsum(const void* x, float* out, void* ws, size_t ws_bytes, int rows, int cols, int ld, int accumulate,
                          int dtype, void* stream) {
    if (!out) ASR_FAIL(ASR_EINVAL, "asr_colsum: null out");
    return colsum_impl<false>((void*)x, nullptr, out, ws, ws_bytes, rows, cols, ld, accumulate, dtype, (hipStream_t)stream, "asr_colsum");
}

extern "C" int asr_relu_bwd(void* da, const void* a, float* dbias, void* ws, size_t ws_bytes, int rows, int cols, int dtype,
                            void* stream) {
    return colsum_impl<true>(da, a, dbias, ws, ws_bytes, rows, cols, cols, 1, dtype, (hipStream_t)stream, "asr_relu_bwd");
}

extern "C" int asr_dec_preprocess(const int64_t* tgt, int32_t* ys_in, int32_t* ys_out, int32_t* labels32, int32_t* dec_len,
                                  int32_t* lab_len, float* n_valid, int B, int Lmax, int sos, int eos, const int64_t* len_a, int32_t* len32_a,
                                  const int64_t* len_b, int32_t* len32_b, void* stream) {
    if (!tgt || !ys_in || !ys_out || !labels32 || !dec_len || !lab_len || !n_valid) ASR_FAIL(ASR_EINVAL, "asr_dec_preprocess: null pointer");
    if ((len_a && !len32_a) || (len_b && !len32_b)) ASR_FAIL(ASR_EINVAL, "asr_dec_preprocess: a length vector without its int32 output");
    if (B <= 0 || B > 1024 || Lmax <= 0) ASR_FAIL(ASR_EINVAL, "asr_dec_preprocess: bad shape B=%d Lmax=%d (B <= 1024)", B, Lmax);
    const int threads = ceil_div(B, 64) * 64;
    dec_preprocess_kernel<<<1, threads, 0, (hipStream_t)stream>>>(tgt, ys_in, ys_out, labels32, dec_len, lab_len, n_valid, B, Lmax, sos, eos, len_a, len32_a, len_b, len32_b);
    ASR_CHECK_LAUNCH("asr_dec_preprocess");
    return ASR_OK;
}

extern "C" int asr_embed_pe_fwd(const int32_t* ids, const void* emb, const float* pe, void* y, float scale, int B, int To, int d,
                                int V, float drop_p, uint32_t drop_seed, int dtype, void* stream) {
    if (!ids || !emb || !pe || !y) ASR_FAIL(ASR_EINVAL, "asr_embed_pe_fwd: null pointer");
    if (B <= 0 || To <= 0 || d <= 0 || V <= 0) ASR_FAIL(ASR_EINVAL, "asr_embed_pe_fwd: bad shape");
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (d & 1))) ASR_FAIL(ASR_EINVAL, "asr_embed_pe_fwd: bad dropout p=%f", drop_p);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_thr16(drop_p);
    const float ds = 1.f / (1.f - drop_p);
    // the gather always reads the fp32 master embedding (exact), whatever the activation dtype
    if (dtype == ASR_F32) embed_pe_fwd_kernel<float, float><<<B * To, 256, 0, st>>>(ids, (const float*)emb, pe, (float*)y, scale, B * To, To, d, V, drop_seed, thr, ds);
    else if (dtype == ASR_BF16) embed_pe_fwd_kernel<bf16_t, float><<<B * To, 256, 0, st>>>(ids, (const float*)emb, pe, (bf16_t*)y, scale, B * To, To, d, V, drop_seed, thr, ds);
    else ASR_FAIL(ASR_EDTYPE, "asr_embed_pe_fwd: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_embed_pe_fwd");
    return ASR_OK;
}

extern "C" int asr_embed_bwd(const int32_t* ids, const void* dy, const void* dy2, float* demb, float scale, int rows, int d, int V, float drop_p,
                             uint32_t drop_seed, int dtype, void* stream) {
    if (!ids || !dy || !demb) ASR_FAIL(ASR_EINVAL, "asr_embed_bwd: null pointer");
    if (rows <= 0 || d <= 0) ASR_FAIL(ASR_EINVAL, "asr_embed_bwd: bad shape");
    if (drop_p < 0.f || drop_p >= 1.f) ASR_FAIL(ASR_EINVAL, "asr_embed_bwd: bad dropout p=%f", drop_p);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_thr16(drop_p);
    const float ds = 1.f / (1.f - drop_p);
    if (dtype != ASR_F32 && dtype != ASR_BF16) ASR_FAIL(ASR_EDTYPE, "asr_embed_bwd: dtype %d", dtype);
    if (asr_deterministic()) {
        if (dtype == ASR_F32) embed_bwd_serial_kernel<float><<<ceil_div(d, 256), 256, 0, st>>>(ids, (const float*)dy, (const float*)dy2, demb, scale, rows, d, V, drop_seed, thr, ds);
        else embed_bwd_serial_kernel<bf16_t><<<ceil_div(d, 256), 256, 0, st>>>(ids, (const bf16_t*)dy, (const bf16_t*)dy2, demb, scale, rows, d, V, drop_seed, thr, ds);
    } else if (dtype == ASR_F32) embed_bwd_kernel<float><<<rows, 256, 0, st>>>(ids, (const float*)dy, (const float*)dy2, demb, scale, d, V, drop_seed, thr, ds);
    else embed_bwd_kernel<bf16_t><<<rows, 256, 0, st>>>(ids, (const bf16_t*)dy, (const bf16_t*)dy2, demb, scale, d, V, drop_seed, thr, ds);
    ASR_CHECK_LAUNCH("asr_embed_bwd");
    return ASR_OK;
}

extern "C" int asr_dropout_mask(uint8_t* mask, int rows, int cols, float drop_p, uint32_t drop_seed, void* stream) {
    if (!mask || rows <= 0 || cols <= 0 || (cols & 1)) ASR_FAIL(ASR_EINVAL, "asr_dropout_mask: bad arguments (cols must be even)");
    const size_t n = (size_t)rows * cols;
    dropout_mask_kernel<<<(int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), 256, 0, (hipStream_t)stream>>>(mask, (size_t)rows, cols, drop_seed, drop_thr16(drop_p));
    ASR_CHECK_LAUNCH("asr_dropout_mask");
    return ASR_OK;
}

extern "C" int asr_sdpa_dropout_mask(uint8_t* mask, int B, int H, int Tq, int Tk, float drop_p, uint32_t drop_seed, void* stream) {
    if (!mask || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) ASR_FAIL(ASR_EINVAL, "asr_sdpa_dropout_mask: bad arguments");
    const size_t n = (size_t)B * H * Tq * Tk;   // element index ((b*H+h)*Tq+q)*Tk+k is the linear index
    dropout_mask_kernel<<<(int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), 256, 0, (hipStream_t)stream>>>(mask, (size_t)B * H * Tq, Tk, drop_seed, drop_thr16(drop_p));
    ASR_CHECK_LAUNCH("asr_sdpa_dropout_mask");
    return ASR_OK;
}

extern "C" size_t asr_sumsq_workspace_bytes(size_t n) { return (size_t)ss_grid(n) * sizeof(float); }

extern "C" int asr_grad_sumsq(const float* g, size_t n, float* sumsq, void* ws, size_t ws_bytes, void* stream) {
    if (!g || !sumsq || !ws) ASR_FAIL(ASR_EINVAL, "asr_grad_sumsq: null pointer");
    if (ws_bytes < asr_sumsq_workspace_bytes(n)) ASR_FAIL(ASR_EWORKSPACE, "asr_grad_sumsq: workspace too small");
    if (((uintptr_t)g) % 16) ASR_FAIL(ASR_EINVAL, "asr_grad_sumsq: g must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = ss_grid(n);
    sumsq_partial_kernel<<<grid, SS_BLOCK, 0, st>>>(g, n / 4, n, (float*)ws);
    sum_finalize_kernel<<<1, 256, 0, st>>>((const float*)ws, grid, sumsq);
    ASR_CHECK_LAUNCH("asr_grad_sumsq");
    return ASR_OK;
}

extern "C" int asr_grad_sumsq_noam(const float* g, size_t n, float* sumsq, void* ws, size_t ws_bytes, int32_t* step, float* hyper, float model_size,
                                   float warmup, float factor, float lr_const, float b1, float b2, void* stream) {
    if (!g || !sumsq || !ws || !step || !hyper) ASR_FAIL(ASR_EINVAL, "asr_grad_sumsq_noam: null pointer");
    if (ws_bytes < asr_sumsq_workspace_bytes(n)) ASR_FAIL(ASR_EWORKSPACE, "asr_grad_sumsq_noam: workspace too small");
    if (((uintptr_t)g) % 16) ASR_FAIL(ASR_EINVAL, "asr_grad_sumsq_noam: g must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = ss_grid(n);
    sumsq_partial_kernel<<<grid, SS_BLOCK, 0, st>>>(g, n / 4, n, (float*)ws);
    sum_finalize_noam_kernel<<<1, 256, 0, st>>>((const float*)ws, grid, sumsq, step, hyper, model_size, warmup, factor, lr_const, b1, b2);
    ASR_CHECK_LAUNCH("asr_grad_sumsq_noam");
    return ASR_OK;
}

extern "C" int asr_noam_hyper(int32_t* step, float* hyper, float model_size, float warmup, float factor, float lr_const, float b1,
                              float b2, void* stream) {
    if (!step || !hyper) ASR_FAIL(ASR_EINVAL, "asr_noam_hyper: null pointer");
    noam_hyper_kernel<<<1, 1, 0, (hipStream_t)stream>>>(step, hyper, model_size, warmup, factor, lr_const, b1, b2);
    ASR_CHECK_LAUNCH("asr_noam_hyper");
    return ASR_OK;
}

extern "C" int asr_adam_step(float* p, float* g, float* m, float* v, void* p_lp, size_t n, const float* hyper, const float* sumsq,
                             float max_norm, float b1, float b2, float eps, int write_clipped, void* stream) {
    if (!p || !g || !m || !v || !hyper) ASR_FAIL(ASR_EINVAL, "asr_adam_step: null pointer");
    if (max_norm > 0.f && !sumsq) ASR_FAIL(ASR_EINVAL, "asr_adam_step: clipping needs sumsq");
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) % 16) ASR_FAIL(ASR_EINVAL, "asr_adam_step: buffers must be 16-byte aligned");
    adam_kernel<<<ew_grid(n / 4), EW_BLOCK, 0, (hipStream_t)stream>>>(p, g, m, v, (bf16_t*)p_lp, n / 4, n, hyper, sumsq, max_norm, b1, b2, eps, write_clipped);
    ASR_CHECK_LAUNCH("asr_adam_step");
    return ASR_OK;
}

extern "C" int asr_loss_combine(const float* row_nll, int M, const float* n_valid, const float* nll, int B, float w_ce, float w_ctc,
                                float* loss, void* stream) {
    if (!loss || (row_nll && !n_valid)) ASR_FAIL(ASR_EINVAL, "asr_loss_combine: null pointer");
    loss_combine_kernel<<<1, 256, 0, (hipStream_t)stream>>>(row_nll, M, n_valid, nll, B, w_ce, w_ctc, loss);
    ASR_CHECK_LAUNCH("asr_loss_combine");
    return ASR_OK;
}
