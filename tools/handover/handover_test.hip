// Micro-benchmark of three ways to hand a kernel's output from one stream to another on MI355X (round 3):
//   0  hipEventRecord behind the producer + hipStreamWaitEvent            (a barrier packet in the producer's queue)
//   1  hipExtLaunchKernelGGL stop event on the producer + hipStreamWaitEvent (completion signal on the dispatch packet)
//   2  the producer's last workgroup writes a flag, the other stream polls it with hipStreamWaitValue32 (nothing in the producer's queue)
//   3  no hand-over at all (lower bound; the consumer may run early)
// Per iteration: P (main, ~20 us) -> [hand-over] -> W (side, ~20 us) ; D (main, ~20 us).  Prints us per iteration of the main chain.
//   hipcc --offload-arch=gfx950 -O2 tools/handover/handover_test.hip -o /tmp/handover_test && /tmp/handover_test
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy(float* out, int iters, unsigned* counter, unsigned* flag, unsigned seq) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[0] = v;
    if (counter) {      // the last workgroup to finish publishes `seq`
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(counter, 1u) == gridDim.x - 1) {
                *counter = 0u;
                __threadfence();
                atomicExch(flag, seq);
            }
        }
    }
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t mainS, sideS;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&sideS, hipStreamNonBlocking, least));
    float* out; CK(hipMalloc(&out, 1024));
    unsigned* counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    unsigned* flag = nullptr;
    if (can) { CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory)); CK(hipMemset(flag, 0, 8)); }
    hipEvent_t evs[64];
    for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming | hipEventReleaseToDevice));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int iters = 6000, N = 40;
    unsigned seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2 && !can) { printf("mode 2: no stream wait value support\n"); continue; }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, mainS));
            for (int i = 0; i < N; ++i) {
                hipEvent_t ev = evs[i & 63];
                ++seq;
                if (mode == 1) hipExtLaunchKernelGGL(busy, dim3(256), dim3(256), 0, mainS, nullptr, ev, 0, out, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
                else if (mode == 2) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, mainS, out, iters, counter, flag, seq);
                else hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, mainS, out, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
                if (mode == 0) { CK(hipEventRecord(ev, mainS)); CK(hipStreamWaitEvent(sideS, ev, 0)); }
                if (mode == 1) CK(hipStreamWaitEvent(sideS, ev, 0));
                if (mode == 2) CK(hipStreamWaitValue32(sideS, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                hipLaunchKernelGGL(busy, dim3(128), dim3(256), 0, sideS, out + 64, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
                hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, mainS, out + 128, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
            }
            CK(hipEventRecord(t1, mainS));
            CK(hipStreamSynchronize(mainS));
            CK(hipStreamSynchronize(sideS));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            printf("mode %d rep %d: %.2f us per iteration (main chain: two kernels)\n", mode, rep, ms * 1e3f / N);
        }
    }
    return 0;
}
