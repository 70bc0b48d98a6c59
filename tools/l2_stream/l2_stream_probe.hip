// Probe: how many bytes per second does ONE CU pull through its vector memory path, and what does it depend on?
// The row-chain kernel (tools/row_chain) streamed weights at ~40 GB/s per CU with 256 KiB in flight; the GEMM kernels' tile streams come out at
// similar per-CU rates.  If that is a property of the CU (not of those kernels), every GEMM of the step is bound by FLOP per L2 byte.
// Each workgroup reads its region (private, or one region shared by all) `reps` times with 16-B loads, 1 KiB of consecutive bytes per wave and
// instruction, `U` instructions in flight per wave; sums what it read (so nothing is optimised away).
//   argv: region KiB per workgroup, shared (0/1), workgroups, waves per workgroup, reps
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream_probe l2_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (the LDS template switch is unused: plain stores into LDS are dead-store-eliminated together with their loads; the LDS-DMA path is what the
// GEMM kernels use and is measured there)
template <int U, bool LDS>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ buf, size_t region16, int shared, int reps, unsigned* out) {
    extern __shared__ u32x4 lds[];
    const u32x4* base = buf + (shared ? 0 : (size_t)blockIdx.x * region16);
    const int tid = threadIdx.x, nthr = blockDim.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = tid; i + (size_t)(U - 1) * nthr < region16; i += (size_t)U * nthr) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = base[i + (size_t)u * nthr];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (LDS) lds[tid + (u & 1) * nthr] = v[u];
                else acc += v[u];
            }
        }
    }
    if (LDS) acc = lds[tid];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) out[0] = 1;
}

template <int U, bool LDS>
static void run(const u32x4* buf, size_t region_bytes, int shared, int wgs, int waves, int reps, unsigned* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = LDS ? (size_t)2 * 64 * waves * 16 : 0;
    stream_kernel<U, LDS><<<wgs, 64 * waves, lds>>>(buf, region_bytes / 16, shared, 1, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    stream_kernel<U, LDS><<<wgs, 64 * waves, lds>>>(buf, region_bytes / 16, shared, reps, out);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)region_bytes * reps * wgs;
    printf("  U=%2d %s: %8.3f ms, %7.1f GB/s per workgroup, %6.2f TB/s in all\n", U, LDS ? "-> LDS " : "-> regs", ms, bytes / wgs / ms / 1e6, bytes / ms / 1e9);
}

int main(int argc, char** argv) {
    const size_t region_kib = argc > 1 ? atoi(argv[1]) : 64;
    const int shared = argc > 2 ? atoi(argv[2]) : 0, wgs = argc > 3 ? atoi(argv[3]) : 256, waves = argc > 4 ? atoi(argv[4]) : 8;
    int reps = argc > 5 ? atoi(argv[5]) : 0;
    const size_t region = region_kib * 1024;
    if (!reps) reps = (int)((size_t)64 * 1024 * 1024 / region) + 1;      // ~64 MiB per workgroup
    const size_t total = shared ? region : region * wgs;
    u32x4* buf;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    unsigned* out;
    CK(hipMalloc(&out, 4));
    printf("region %zu KiB per workgroup (%s, %.1f MiB in all), %d workgroups x %d waves, %d passes\n", region_kib, shared ? "one region shared by all" : "private", total / 1048576.0, wgs,
           waves, reps);
    // a pass must hold at least one trip of U instructions per thread (else the loop body never runs and the "rate" is the launch)
    const size_t per_thread = region / 16 / (64 * waves);
    if (per_thread >= 4) run<4, false>(buf, region, shared, wgs, waves, reps, out);
    if (per_thread >= 8) run<8, false>(buf, region, shared, wgs, waves, reps, out);
    if (per_thread >= 16) run<16, false>(buf, region, shared, wgs, waves, reps, out);
    return 0;
}
