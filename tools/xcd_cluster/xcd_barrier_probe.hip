// Probe: what does a barrier between the workgroups of ONE XCD cost, against a device-wide one?
//
// Background: a fused decoder-layer kernel has to split every projection's N across CUs (a CU pulls ~40 GB/s out of L2)
// and exchange rows between CUs inside the kernel.  Device-wide that costs a kernel boundary; the 32 CUs of one XCD share an L2, so a cluster
// of workgroups on one XCD could synchronise and exchange through L2 alone.  This program measures it: 256 workgroups (one per CU) find
// their XCD by XCC_ID, take a rank in that XCD's cluster, and run R rounds of { write a word, barrier, read the neighbour's word and check it }:
//   mode 0  XCD-local: arrive with an atomic executed in L2 (no sc1), spin with L1-bypassing loads (sc0), data by plain store + s_waitcnt
//   mode 1  same clusters, agent-scope atomics and fences (what the memory model offers)
//   mode 2  one barrier across all 256 workgroups, agent scope
//   mode 3  XCD-local: arrive AND poll with atomics executed in L2 (an add of 0 is the poll), data by plain store + s_waitcnt, the reader
//           invalidates its L1 (buffer_inv sc1) before a plain load
//   mode 4  XCD clusters, agent-scope atomics all RELAXED (no release / acquire: no L2 write-back, no invalidate), data by agent-scope atomic store / load
// Every spin is bounded (an abort flag ends all waiting); a wrong value read after a barrier is counted (visibility check).
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_barrier_probe xcd_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int MAX_XCD = 16, MAX_RANK = 64, SPIN_LIMIT = 1 << 16;

struct Shared {
    unsigned ticket[MAX_XCD];                 // cluster ranks handed out per XCD
    unsigned registered;                      // workgroups that have taken a rank
    unsigned abort_flag;
    unsigned pad0[14];
    unsigned arrive[MAX_XCD][32];             // one barrier counter per XCD, on its own 128-B line
    unsigned grid_arrive[32];
    unsigned data[MAX_XCD][MAX_RANK][32];     // one line per workgroup
    unsigned long long t_ns[MAX_XCD][MAX_RANK];
    unsigned bad[MAX_XCD][MAX_RANK];
    unsigned xcc_of_block[1024];
};
// modes 5 - 7 (all relaxed agent-scope atomics, all 64 lanes take part):
//   mode 5  device-wide barrier, no payload
//   mode 6  XCD clusters, every workgroup stores `payload` bytes with sc1 stores before the barrier and reads its neighbour's with sc1 loads after it
//   mode 7  device-wide, the same payload exchange (the neighbour is the next block: another XCD)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 load16_sc1(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(64) void probe_payload(Shared* sh, u32x4* buf, int mode, int rounds, int payload) {
    __shared__ unsigned s_xcc, s_rank, s_size, s_dead;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        s_xcc = xcc;
        s_dead = 0;
        s_rank = __hip_atomic_fetch_add(&sh->ticket[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&sh->registered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&sh->registered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(&sh->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_size = __hip_atomic_load(&sh->ticket[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (__hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const bool wide = mode != 6;
    const unsigned xcc = s_xcc, rank = s_rank, size = wide ? gridDim.x : s_size;
    unsigned* ctr = wide ? sh->grid_arrive : sh->arrive[xcc];
    const int chunks = payload / 16;      // 16-B pieces per workgroup
    // slot of a workgroup: by block for the device-wide exchange, by (xcd, rank) inside a cluster
    const unsigned me = wide ? blockIdx.x : xcc * 32 + rank, nb = wide ? (blockIdx.x + 1) % gridDim.x : xcc * 32 + (rank + 1) % s_size;
    u32x4* mine = buf + (size_t)me * (chunks ? chunks : 1);
    const u32x4* next = buf + (size_t)nb * (chunks ? chunks : 1);
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        for (int c = threadIdx.x; c < chunks; c += 64) {
            u32x4 v = {(unsigned)r, me, (unsigned)c, 0u};
            store16_sc1(mine + c, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)r * size;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_LIMIT || __hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&sh->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_dead = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (s_dead) break;
        int c = threadIdx.x;
        for (; c + 7 * 64 < chunks; c += 8 * 64) {      // eight loads in flight per lane
            u32x4 v[8];
            const u32x4* q = next + c;
            asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
                         "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
                         "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(q), "v"(q + 256)
                         : "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i][0] != (unsigned)r || v[i][1] != nb || v[i][2] != (unsigned)(c + 64 * i)) ++bad;
        }
        for (; c < chunks; c += 64) {
            u32x4 v = load16_sc1(next + c);
            if (v[0] != (unsigned)r || v[1] != nb || v[2] != (unsigned)c) ++bad;
        }
    }
    for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o);
    if (threadIdx.x == 0) {
        sh->t_ns[xcc][rank] = (wall_clock64() - t0) * 10ull;
        sh->bad[xcc][rank] = bad;
        sh->xcc_of_block[blockIdx.x] = xcc;
    }
}

__device__ __forceinline__ unsigned load_l2(const unsigned* p) {      // bypasses the CU's L1, reads the XCD's L2
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned add_l2(unsigned* p, unsigned x) {      // atomic executed in the XCD's L2, returns the old value
    unsigned v;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(x) : "memory");
    return v;
}
__device__ __forceinline__ void store_l2(unsigned* p, unsigned x) {
    asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(x) : "memory");
}

__global__ __launch_bounds__(64) void probe(Shared* sh, int mode, int rounds) {
    __shared__ unsigned s_xcc, s_rank, s_size;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        s_xcc = xcc;
        sh->xcc_of_block[blockIdx.x] = xcc;
        s_rank = __hip_atomic_fetch_add(&sh->ticket[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&sh->registered, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&sh->registered, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {      // every workgroup is resident and has a rank
            if (++spins > SPIN_LIMIT || __hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&sh->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        s_size = __hip_atomic_load(&sh->ticket[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned xcc = s_xcc, rank = s_rank, size = mode == 2 ? gridDim.x : s_size;
    if (__hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    unsigned* ctr = mode == 2 ? sh->grid_arrive : sh->arrive[xcc];
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        for (int r = 1; r <= rounds; ++r) {
            unsigned* mine = &sh->data[xcc][rank][0];
            const unsigned* next = &sh->data[xcc][(rank + 1) % s_size][0];
            const unsigned target = (unsigned)r * size;
            bool dead = false;
            if (mode == 3) {
                store_l2(mine, (unsigned)r);
                add_l2(ctr, 1u);
                int spins = 0;
                while (add_l2(ctr, 0u) < target) {
                    if (++spins > SPIN_LIMIT) { dead = true; break; }
                }
                asm volatile("buffer_inv sc1" ::: "memory");
                unsigned v = *(volatile const unsigned*)next;
                if (!dead && v != (unsigned)r) ++bad;
            } else if (mode == 4) {
                __hip_atomic_store(mine, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > SPIN_LIMIT) { dead = true; break; }
                }
                if (!dead && __hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) ++bad;
            } else if (mode == 0) {
                store_l2(mine, (unsigned)r);
                add_l2(ctr, 1u);
                int spins = 0;
                while (load_l2(ctr) < target) {
                    if (++spins > SPIN_LIMIT) { dead = true; break; }
                }
                if (!dead && load_l2(next) != (unsigned)r) ++bad;
            } else {
                __hip_atomic_store(mine, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > SPIN_LIMIT) { dead = true; break; }
                }
                if (!dead && __hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) ++bad;
            }
            if (dead || __hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&sh->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        sh->t_ns[xcc][rank] = (wall_clock64() - t0) * 10ull;      // 100 MHz counter
        sh->bad[xcc][rank] = bad;
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, %d rounds per mode, one 64-thread workgroup per CU\n", prop.gcnArchName, cus, rounds);
    Shared* sh;
    CK(hipMalloc(&sh, sizeof(Shared)));
    Shared* host = (Shared*)malloc(sizeof(Shared));
    const int payload = argc > 2 ? atoi(argv[2]) : 65536;
    u32x4* buf;
    CK(hipMalloc(&buf, (size_t)512 * (payload > 16 ? payload : 16)));
    const char* names[8] = {"XCD-local (L2 atomics, sc0 loads)", "XCD clusters, agent-scope atomics", "device-wide, agent-scope atomics",
                            "XCD-local (L2 atomics, L2 polls, inv)", "XCD clusters, agent scope, relaxed", "device-wide, relaxed",
                            "XCD clusters, relaxed + sc1 payload", "device-wide, relaxed + sc1 payload"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(sh, 0, sizeof(Shared)));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (mode < 5) probe<<<cus, 64>>>(sh, mode, rounds);
            else probe_payload<<<cus, 64>>>(sh, buf, mode, rounds, mode == 5 ? 0 : payload);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(host, sh, sizeof(Shared), hipMemcpyDeviceToHost));
            unsigned long long tmax = 0, bad = 0;
            int nx = 0, smin = 1 << 30, smax = 0;
            for (int x = 0; x < MAX_XCD; ++x) {
                if (!host->ticket[x]) continue;
                ++nx;
                if ((int)host->ticket[x] < smin) smin = host->ticket[x];
                if ((int)host->ticket[x] > smax) smax = host->ticket[x];
                for (unsigned k = 0; k < host->ticket[x] && k < MAX_RANK; ++k) {
                    if (host->t_ns[x][k] > tmax) tmax = host->t_ns[x][k];
                    bad += host->bad[x][k];
                }
            }
            int rr = 0;      // does block i sit on XCD i % 8?
            for (int i = 0; i < cus; ++i) rr += host->xcc_of_block[i] == (unsigned)(i % nx);
            printf("mode %d %-36s rep %d: %d XCDs, cluster sizes %d..%d, block i on XCD i %% %d for %d of %d; kernel %.3f ms; %.0f ns per round (slowest workgroup); stale reads %llu; abort %u\n",
                   mode, names[mode], rep, nx, smin, smax, nx, rr, cus, ms, (double)tmax / rounds, bad, host->abort_flag);
        }
    }
    return 0;
}
