// Calibration micro-benchmark, round 3: what the claim replay's access mix can sustain. One thread per "source", as in a replay
// round: every thread issues K independent random 8-byte accesses to a table of n words and waits for them together.
//   load     8-byte loads
//   load16   16-byte loads (one request covers two neighbouring words)
//   store    8-byte stores
//   atomic   64-bit atomicMin without return value (the reservation)
//   atomicr  64-bit atomicMin whose old value is used
// usage: access_bench [n_words = 89523223] [threads = 1048576] [K = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}

template <int MODE, int K>
__global__ void touch(unsigned long long *tab, unsigned long long n, unsigned long long salt, unsigned long long *out) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long idx[K];
#pragma unroll
    for (int j = 0; j < K; j++) idx[j] = mix64(tid * K + j + salt) % n;
    unsigned long long acc = 0;
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < K; j++) acc += tab[idx[j]];
    } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < K; j++) { const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(tab)[idx[j] >> 1]; acc += v.x ^ v.y; }
    } else if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < K; j++) tab[idx[j]] = tid + j;
    } else if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < K; j++) atomicMin(&tab[idx[j]], (salt << 32) | tid);
    } else {
#pragma unroll
        for (int j = 0; j < K; j++) acc += atomicMin(&tab[idx[j]], (salt << 32) | tid);
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

template <int MODE, int K>
static void run(const char *name, unsigned long long *d, unsigned long long n, unsigned threads, unsigned long long *o) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int block : {256, 1024}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((touch<MODE, K>), dim3(threads / block), dim3(block), 0, 0, d, n, (unsigned long long)(0xFFFFFFF0u - rep), o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-8s table %.2f GB, %u threads x %d accesses, block %4d: %.3f ms, %.1f G accesses/s\n", name, n * 8.0 / 1e9, threads, K, block, best,
               (double)threads * K / best / 1e6);
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    const unsigned long long n = argc > 1 ? (unsigned long long)atoll(argv[1]) : 89523223ull;
    const unsigned threads = argc > 2 ? (unsigned)atoll(argv[2]) : 1048576u;
    unsigned long long *d, *o;
    CK(hipMalloc(&d, n * 8));
    CK(hipMemset(d, 0xFF, n * 8));
    CK(hipMalloc(&o, 64));
    CK(hipDeviceSynchronize());
    run<0, 8>("load", d, n, threads, o);
    run<1, 8>("load16", d, n, threads, o);
    run<2, 8>("store", d, n, threads, o);
    run<3, 8>("atomic", d, n, threads, o);
    run<4, 8>("atomicr", d, n, threads, o);
    run<0, 2>("load", d, n, threads, o);
    run<3, 2>("atomic", d, n, threads, o);
    return 0;
}
