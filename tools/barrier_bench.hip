// Calibration micro-benchmark, round 3: what a grid barrier of the claim replay costs (cooperative launch, one arrival counter).
//   full     the barrier of replay_kernels.inc: workgroup-scope release, __syncthreads, ONE thread per workgroup does the
//            agent-scope release increment (L2 write-back), spins, agent-scope acquire (L2 invalidate), __syncthreads
//   nofence  the same with relaxed atomics only (no L2 write-back / invalidate): the floor of counter + spin + __syncthreads
//   dirty    full, with every thread storing to its own word of a large array before each barrier (dirty lines to write back)
//   xcd      two levels: arrival per XCD (counter per XCC_ID), the last workgroup of an XCD writes its L2 back and arrives at the
//            grid counter; after the grid phase every workgroup invalidates
// usage: barrier_bench [workgroups = 256] [threads = 1024] [barriers = 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool FENCE>
__device__ __forceinline__ void grid_barrier(unsigned long long *ctr, uint32_t &phase) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        phase++;
        const unsigned long long target = (unsigned long long)phase * gridDim.x;
        if (FENCE) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int MODE>
__global__ __launch_bounds__(1024) void bench(unsigned long long *ctl, unsigned long long *big, uint64_t n_big, int n_barriers, uint32_t *xcc_out) {
    uint32_t phase = 0;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 3) {
        uint32_t xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (threadIdx.x == 0) xcc_out[blockIdx.x] = xcc;
        // workgroups per XCD: counted once
        __shared__ unsigned long long s_per_xcd;
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctl[64 + xcc], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier<true>(ctl, phase);
        if (threadIdx.x == 0) s_per_xcd = __hip_atomic_load(&ctl[64 + xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned long long per_xcd = s_per_xcd;
        uint32_t ph2 = 0;
        for (int b = 0; b < n_barriers; b++) {
            big[(gid * 9973u + b) % n_big] = b;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (threadIdx.x == 0) {
                ph2++;
                // arrive at the XCD counter; the last one writes the XCD's L2 back and arrives at the grid counter for all of them
                const unsigned long long old = __hip_atomic_fetch_add(&ctl[16 + xcc], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned long long)ph2 * per_xcd)
                    __hip_atomic_fetch_add(&ctl[1], per_xcd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long target = (unsigned long long)ph2 * gridDim.x;
                while (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        return;
    }
    for (int b = 0; b < n_barriers; b++) {
        if (MODE == 2) big[(gid * 9973u + b) % n_big] = b;
        if (MODE == 1) grid_barrier<false>(ctl, phase);
        else grid_barrier<true>(ctl, phase);
    }
}

template <int MODE>
static void run(const char *name, int wgs, int threads, int n_barriers, unsigned long long *ctl, unsigned long long *big, uint64_t n_big, uint32_t *xcc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ctl, 0, 128 * 8));
        void *args[] = {&ctl, &big, &n_big, &n_barriers, &xcc};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(bench<MODE>), dim3(wgs), dim3(threads), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%-8s %4d workgroups x %4d threads: %.2f us per barrier\n", name, wgs, threads, ms * 1e3 / n_barriers);
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 1024, nb = argc > 3 ? atoi(argv[3]) : 200;
    unsigned long long *ctl, *big; uint32_t *xcc;
    const uint64_t n_big = 1ull << 28;  // 2 GB
    CK(hipMalloc(&ctl, 128 * 8)); CK(hipMalloc(&big, n_big * 8)); CK(hipMalloc(&xcc, 4096 * 4));
    CK(hipMemset(big, 0, n_big * 8));
    run<1>("nofence", wgs, threads, nb, ctl, big, n_big, xcc);
    run<0>("full", wgs, threads, nb, ctl, big, n_big, xcc);
    run<2>("dirty", wgs, threads, nb, ctl, big, n_big, xcc);
    run<3>("xcd", wgs, threads, nb, ctl, big, n_big, xcc);
    uint32_t h[16];
    CK(hipMemcpy(h, xcc, sizeof h, hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..15:");
    for (int i = 0; i < 16; i++) printf(" %u", h[i]);
    printf("\n");
    return 0;
}
