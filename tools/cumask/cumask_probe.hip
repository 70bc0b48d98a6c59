// Where do the workgroups of a CU-masked stream run?  hipExtStreamCreateWithCUMask takes one bit per CU; this prints, for a few masks, the set of
// (XCC, SE, SH, CU) a 1024-workgroup kernel of 64-thread workgroups landed on - i.e. the bit layout of the mask on an 8-XCD part - and whether a kernel on
// the NULL stream waits for a spin kernel on such a stream (hipExtStreamCreateWithCUMask streams are "blocking" streams).
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask/cumask_probe.hip -o tools/cumask/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>
#include <chrono>

__global__ void where_kernel(uint32_t* out) {
    // spin a little so that workgroups spread over every CU the stream may use
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID[3:0]
        out[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
    }
}
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
__global__ void tiny_kernel(uint32_t* p) { if (threadIdx.x == 0) p[0] += 1; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int run_mask(const char* name, const std::vector<uint32_t>& mask, uint32_t* dbuf, uint32_t* hbuf) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e)); return 0; }
    const int nwg = 4096;
    where_kernel<<<nwg, 64, 0, st>>>(dbuf);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hbuf, dbuf, nwg * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    std::set<uint32_t> xcc_cu[16];
    for (int i = 0; i < nwg; ++i) {
        const uint32_t v = hbuf[i], xcc = v >> 24, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
        xcc_cu[xcc & 15].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%-28s CUs used: %3zu   per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2zu", xcc_cu[x].size());
    printf("\n");
    CK(hipStreamDestroy(st));
    return 0;
}

int main() {
    uint32_t *dbuf, *hbuf = (uint32_t*)malloc(4096 * 4);
    CK(hipMalloc(&dbuf, 4096 * 4));
    auto bits = [](int lo, int hi) { std::vector<uint32_t> m(8, 0u); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); return m; };
    run_mask("all 256", bits(0, 256), dbuf, hbuf);
    run_mask("bits 0..191", bits(0, 192), dbuf, hbuf);
    run_mask("bits 0..127", bits(0, 128), dbuf, hbuf);
    run_mask("bits 0..31", bits(0, 32), dbuf, hbuf);
    run_mask("bits 0..7", bits(0, 8), dbuf, hbuf);
    run_mask("bits 192..255", bits(192, 256), dbuf, hbuf);
    { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; ++i) if ((i / 8) % 4 != 3) m[i / 32] |= 1u << (i % 32); run_mask("every 4th group of 8 off", m, dbuf, hbuf); }
    // does the NULL stream wait for a masked (blocking) stream?  and a non-blocking stream?
    hipStream_t masked, nb;
    auto m192 = bits(0, 192);
    CK(hipExtStreamCreateWithCUMask(&masked, 8, m192.data()));
    CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
    for (int which = 0; which < 2; ++which) {
        hipStream_t tiny_on = which == 0 ? (hipStream_t)0 : nb;
        CK(hipDeviceSynchronize());
        spin_kernel<<<1, 64, 0, masked>>>(200000000LL);      // ~2 s at 100 MHz wall clock
        auto t0 = std::chrono::steady_clock::now();
        tiny_kernel<<<1, 64, 0, tiny_on>>>(dbuf);
        CK(hipStreamSynchronize(tiny_on));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipDeviceSynchronize());
        printf("tiny kernel on the %s stream while a spin runs on the masked stream: done after %.2f ms\n", which == 0 ? "NULL" : "non-blocking", ms);
    }
    return 0;
}
