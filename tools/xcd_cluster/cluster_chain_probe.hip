// Probe 2: a row-local run of a decoder layer (out-projection + residual + LayerNorm + the next projection) as ONE launch in which every
// projection's N is split across the 32 workgroups of a cluster and the rows are exchanged between them inside the kernel.
//
// xcd_barrier_probe.hip measured the exchange primitives: a barrier between 32 workgroups built from RELAXED agent-scope atomics costs 1.3 us
// per round (5.8 us with release / acquire - the L2 write-back and invalidate of the fences are what a kernel boundary pays too), data passed
// with sc1 stores / loads arrives intact.  tools/row_chain showed why N must be split: a CU pulls ~40 GB/s out of L2, so a workgroup that
// streams every weight of its chain is slower than the separate kernels.  Here cluster c = blockIdx % 8 (in practice the workgroups of one
// XCD, but nothing depends on that: every exchange is agent scope) owns rows [c * rpc, (c + 1) * rpc) and its workgroup j = blockIdx / 8 owns
// 16 output columns of every projection:
//   op 1  z = ctx W_fc^T + b + x_in          (16 columns per workgroup, all rows of the cluster; A by plain loads: written by an earlier kernel)
//         per-row (sum, M2) of the 16 columns -> part[c][row][j]                                   barrier 1
//         LayerNorm statistics by merging the 32 partials (Chan's parallel variance), y = LN(z) gamma + beta, rows t >= len zeroed;
//         y slice by sc1 stores, xhat / rstd by plain stores                                        barrier 2
//   op 2  q = y W_q^T + b                     (A = y by sc1 loads: written by the other workgroups of the cluster in THIS kernel)
// One wave per 16-row block (MFMA 16x16x32, C^T form: n in registers, m on the lane).  Checked against a CPU fp32 evaluation; timed over
// 200 launches.  Every spin is bounded (abort flag).
// Build: hipcc --offload-arch=gfx950 -O3 -o cluster_chain_probe cluster_chain_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int NCL = 8, CL = 32, D = 512, KS = D / 32, SPIN_LIMIT = 1 << 18, MAX_RPC = 128;

struct Args {
    int M, rpc, To;
    const int* lens;
    const bf16_t *ctx, *x_in, *w_fc, *w_q;
    const float *b_fc, *b_q, *gamma, *beta;
    bf16_t *y, *xhat, *q;
    float* rstd;
    f32x2* part;          // [NCL][MAX_RPC][CL] (sum, M2) of a workgroup's 16 columns
    unsigned* ctr;        // [NCL][32]: one counter per cluster on its own line
    unsigned* abort_flag;
    unsigned base;        // value of every counter before this launch
};

// sc1 loads through inline asm: the WAIT is part of the same statement.  Outside it the compiler does not know the results are still in
// flight - it re-used a destination register as the address of a later instruction, the late-landing load overwrote it, and the kernel faulted
// on address 0 (first version of this probe).
__device__ __forceinline__ void load16x16_sc1(bf16x8 (&v)[16], const bf16_t* p) {      // 16 B at p + 64 B * i, i < 16
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
__device__ __forceinline__ void store8_sc1(bf16_t* p, bf16x4 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store8f_sc1(f32x2* p, f32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void load16fx4_sc1(f32x4 (&v)[4], const f32x2* p) {      // 16 B at p + 16 B * i, i < 4
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(p)
                 : "memory");
}

// all threads: own stores done; thread 0: arrive and wait for `target` arrivals.  false = gave up (abort flag set)
__device__ __forceinline__ bool cluster_barrier(unsigned* ctr, unsigned target, unsigned* abort_flag, unsigned* s_dead) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *s_dead == 0;
}

__global__ __launch_bounds__(512) void cluster_chain_kernel(const Args a) {
    __shared__ unsigned s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    const int c = blockIdx.x & (NCL - 1), j = blockIdx.x >> 3, n0 = 16 * j;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, mi = lane & 15, kq = lane >> 4;
    const int row_end = min(a.M, (c + 1) * a.rpc), lrow = 16 * w + mi, row = c * a.rpc + lrow;
    const bool valid = lrow < a.rpc && row < row_end;
    const size_t rowc = valid ? (size_t)row : 0;      // invalid lanes read row 0, results unused
    unsigned* ctr = a.ctr + 32 * c;
    __syncthreads();
    // ---- op 1: z = ctx W_fc^T  (D[n][m]: this lane holds m = mi, n = n0 + 4 kq + reg)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        bf16x8 af[KS], wf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            af[ks] = *(const bf16x8*)(a.ctx + rowc * D + 32 * ks + 8 * kq);
            wf[ks] = *(const bf16x8*)(a.w_fc + (size_t)(n0 + mi) * D + 32 * ks + 8 * kq);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
    }
    const int n = n0 + 4 * kq;
    float z[4];
    {
        const f32x4 b = *(const f32x4*)(a.b_fc + n);
        const bf16x4 r = *(const bf16x4*)(a.x_in + rowc * D + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = acc[i] + b[i] + (float)r[i];
    }
    float s = z[0] + z[1] + z[2] + z[3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float lmean = s * (1.f / 16.f);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) m2 += (z[i] - lmean) * (z[i] - lmean);
    m2 += __shfl_xor(m2, 16);
    m2 += __shfl_xor(m2, 32);
    f32x2* prow = a.part + ((size_t)c * MAX_RPC + lrow) * CL;
    if (kq == 0 && lrow < a.rpc) {
        f32x2 v = {s, m2};
        store8f_sc1(prow + j, v);
    }
    if (!cluster_barrier(ctr, a.base + CL, a.abort_flag, &s_dead)) return;
    // ---- LayerNorm: merge the 32 partials of this row (each over 16 values): mean = sum s / 512, M2 = sum M2_j + 16 sum (mean_j - mean)^2
    float mean, rstd;
    {
        const f32x2* pr = a.part + ((size_t)c * MAX_RPC + (lrow < a.rpc ? lrow : 0)) * CL + 8 * kq;      // this lane: partials 8 kq .. 8 kq + 7
        f32x4 p[4];
        load16fx4_sc1(p, pr);
        float ts = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ts += p[i][0] + p[i][2];
        ts += __shfl_xor(ts, 16);
        ts += __shfl_xor(ts, 32);
        mean = ts * (1.f / (float)D);
        float t2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d0 = p[i][0] * (1.f / 16.f) - mean, d1 = p[i][2] * (1.f / 16.f) - mean;
            t2 += p[i][1] + p[i][3] + 16.f * (d0 * d0 + d1 * d1);
        }
        t2 += __shfl_xor(t2, 16);
        t2 += __shfl_xor(t2, 32);
        rstd = rsqrtf(t2 * (1.f / (float)D) + 1e-5f);
    }
    {
        bool keep = true;
        if (a.lens && valid) {
            const int b = row / a.To, t = row - b * a.To;
            keep = t < a.lens[b];
        }
        const f32x4 g = *(const f32x4*)(a.gamma + n), be = *(const f32x4*)(a.beta + n);
        bf16x4 xh, yv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = (z[i] - mean) * rstd;
            xh[i] = (bf16_t)x;
            yv[i] = (bf16_t)(keep ? x * g[i] + be[i] : 0.f);
        }
        if (valid) {
            store8_sc1(a.y + (size_t)row * D + n, yv);
            *(bf16x4*)(a.xhat + (size_t)row * D + n) = xh;
            if (j == 0 && kq == 0) a.rstd[row] = rstd;
        }
    }
    if (!cluster_barrier(ctr, a.base + 2 * CL, a.abort_flag, &s_dead)) return;
    // ---- op 2: q = y W_q^T + b   (y rows: written by the other workgroups of the cluster during this kernel -> sc1 loads)
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        bf16x8 af[KS], wf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *(const bf16x8*)(a.w_q + (size_t)(n0 + mi) * D + 32 * ks + 8 * kq);
        load16x16_sc1(af, a.y + rowc * D + 8 * kq);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
    }
    if (valid) {
        const f32x4 b = *(const f32x4*)(a.b_q + n);
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(acc[i] + b[i]);
        *(bf16x4*)(a.q + (size_t)row * D + n) = o;
    }
}

static float bf(float x) { return (float)(bf16_t)x; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, To = argc > 2 ? atoi(argv[2]) : 17, reps = 200;
    const int M = B * To, rpc = (M + NCL - 1) / NCL, waves = (rpc + 15) / 16;
    if (rpc > MAX_RPC) { fprintf(stderr, "rows per cluster %d > %d\n", rpc, MAX_RPC); return 1; }
    printf("M = %d rows (%d x %d), %d rows per cluster, %d waves per workgroup, 256 workgroups\n", M, B, To, rpc, waves);
    srand(7);
    auto rnd = [](float s) { return s * ((rand() & 0xffff) / 32768.f - 1.f); };
    std::vector<float> ctx((size_t)M * D), xin((size_t)M * D), wfc((size_t)D * D), wq((size_t)D * D), bfc(D), bq(D), ga(D), be(D);
    for (auto& v : ctx) v = bf(rnd(1.f));
    for (auto& v : xin) v = bf(rnd(1.f));
    for (auto& v : wfc) v = bf(rnd(0.06f));
    for (auto& v : wq) v = bf(rnd(0.06f));
    for (int i = 0; i < D; ++i) { bfc[i] = rnd(0.1f); bq[i] = rnd(0.1f); ga[i] = 1.f + rnd(0.2f); be[i] = rnd(0.1f); }
    std::vector<int> lens(B);
    for (int b = 0; b < B; ++b) lens[b] = To - (b % 5);
    auto up_bf = [&](const std::vector<float>& h) {
        std::vector<bf16_t> t(h.size());
        for (size_t i = 0; i < h.size(); ++i) t[i] = (bf16_t)h[i];
        bf16_t* d;
        CK(hipMalloc(&d, t.size() * 2));
        CK(hipMemcpy(d, t.data(), t.size() * 2, hipMemcpyHostToDevice));
        return d;
    };
    auto up_f = [&](const std::vector<float>& h) {
        float* d;
        CK(hipMalloc(&d, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        return d;
    };
    Args a = {};
    a.M = M; a.rpc = rpc; a.To = To;
    int* dl;
    CK(hipMalloc(&dl, B * 4));
    CK(hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice));
    a.lens = dl;
    a.ctx = up_bf(ctx); a.x_in = up_bf(xin); a.w_fc = up_bf(wfc); a.w_q = up_bf(wq);
    a.b_fc = up_f(bfc); a.b_q = up_f(bq); a.gamma = up_f(ga); a.beta = up_f(be);
    CK(hipMalloc(&a.y, (size_t)M * D * 2)); CK(hipMalloc(&a.xhat, (size_t)M * D * 2)); CK(hipMalloc(&a.q, (size_t)M * D * 2));
    CK(hipMalloc(&a.rstd, M * 4));
    CK(hipMalloc(&a.part, sizeof(f32x2) * NCL * MAX_RPC * CL));
    CK(hipMalloc(&a.ctr, 4 * 32 * NCL)); CK(hipMemset(a.ctr, 0, 4 * 32 * NCL));
    CK(hipMalloc(&a.abort_flag, 4)); CK(hipMemset(a.abort_flag, 0, 4));
    unsigned launches = 0;
    auto launch = [&]() {
        a.base = launches * 2 * CL;
        cluster_chain_kernel<<<NCL * CL, 64 * waves>>>(a);
        ++launches;
    };
    launch();
    CK(hipDeviceSynchronize());
    // ---- CPU reference (fp32 accumulation, y rounded to bf16 before the second projection as on the device)
    std::vector<float> y((size_t)M * D), q((size_t)M * D), xh((size_t)M * D), rs(M);
    for (int m = 0; m < M; ++m) {
        float z[D];
        double s = 0;
        for (int n = 0; n < D; ++n) {
            float acc = 0.f;
            for (int k = 0; k < D; ++k) acc += ctx[(size_t)m * D + k] * wfc[(size_t)n * D + k];
            z[n] = acc + bfc[n] + xin[(size_t)m * D + n];
            s += z[n];
        }
        const float mean = (float)(s / D);
        double v = 0;
        for (int n = 0; n < D; ++n) v += (double)(z[n] - mean) * (z[n] - mean);
        const float rstd = 1.f / sqrtf((float)(v / D) + 1e-5f);
        rs[m] = rstd;
        const bool keep = (m % To) < lens[m / To];
        for (int n = 0; n < D; ++n) {
            const float x = (z[n] - mean) * rstd;
            xh[(size_t)m * D + n] = x;
            y[(size_t)m * D + n] = bf(keep ? x * ga[n] + be[n] : 0.f);
        }
        for (int n = 0; n < D; ++n) {
            float acc = 0.f;
            for (int k = 0; k < D; ++k) acc += y[(size_t)m * D + k] * wq[(size_t)n * D + k];
            q[(size_t)m * D + n] = acc + bq[n];
        }
    }
    auto check = [&](const char* name, const bf16_t* dev, const std::vector<float>& ref) {
        std::vector<bf16_t> h(ref.size());
        CK(hipMemcpy(h.data(), dev, h.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
            worst = fmax(worst, fabs((double)(float)h[i] - ref[i]));
            big = fmax(big, fabs(ref[i]));
        }
        printf("  %-5s max |device - cpu| = %.4f (largest |value| %.2f)\n", name, worst, big);
        return worst <= 0.02 * big + 0.01;
    };
    unsigned ab = 0;
    CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
    bool ok = !ab;
    ok &= check("y", a.y, y);
    ok &= check("xhat", a.xhat, xh);
    ok &= check("q", a.q, q);
    {
        std::vector<float> h(M);
        CK(hipMemcpy(h.data(), a.rstd, M * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int m = 0; m < M; ++m) worst = fmax(worst, fabs(h[m] - rs[m]) / rs[m]);
        printf("  rstd  max relative error %.2e\n", worst);
        ok &= worst < 1e-4;
    }
    printf("parity: %s (abort flag %u)\n", ok ? "OK" : "FAILED", ab);
    if (!ok) return 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
        printf("%d back-to-back launches: %.2f us per launch (out-projection + LayerNorm + Q projection; the three separate kernels: ~20 us alone, ~28 us in the step); abort %u\n",
               reps, ms * 1e3 / reps, ab);
    }
    return 0;
}
