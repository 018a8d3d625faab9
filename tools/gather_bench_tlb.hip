// Calibration micro-benchmark, round 3: what bounds dependent random 64-byte gathers once the table is larger than ~4 GB
// (the 2^27 graph's family blocks are 5.7 GB)? Variants, all with the same dependent chain per lane:
//   lane64   every lane loads its own 64-byte block with four 16-byte loads (what the enumeration level does)
//   lane16   every lane loads 16 bytes of its block
//   quad64   four lanes load one block together (lane q of a quad loads quarter q of block i of the quad in load i):
//            an instruction touches 16 lines / pages instead of 64
//   window   lane64, but the random indices fall into a window of the table (footprint of the access, not of the allocation)
// usage: gather_bench_tlb [n_blocks = 89523223] [contiguous = 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill(uint4 *tab, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        tab[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)x * 3u, (uint32_t)(x >> 32) * 5u);
    }
}

template <int MODE>  // 0 lane64, 1 lane16, 2 quad64
__global__ void chase(const uint4 *tab, uint32_t n_blk, uint32_t base, int steps, uint32_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = tid * 2654435761u % n_blk;
    uint32_t acc = 0;
    const int q = threadIdx.x & 3;
    for (int s = 0; s < steps; s++) {
        uint32_t mix = 0;
        if (MODE == 0) {
            const uint4 *p = tab + ((size_t)base + idx) * 4;
#pragma unroll
            for (int l = 0; l < 4; l++) { const uint4 v = p[l]; mix ^= v.x + v.w; acc += v.y; }
        } else if (MODE == 1) {
            const uint4 v = tab[((size_t)base + idx) * 4];
            mix ^= v.x + v.w; acc += v.y;
        } else {
            uint4 got[4];
#pragma unroll
            for (int l = 0; l < 4; l++) {  // load l: the quad reads the block of its lane l; this lane takes quarter q
                const uint32_t bi = __shfl(idx, (threadIdx.x & ~3) | l);
                got[l] = tab[((size_t)base + bi) * 4 + q];
            }
            // transpose inside the quad: this lane needs the four quarters of ITS block = got[q] of lanes 0..3 of the quad
#pragma unroll
            for (int l = 0; l < 4; l++) {
                uint4 v;
                const uint4 mine = q == 0 ? got[0] : q == 1 ? got[1] : q == 2 ? got[2] : got[3];  // (placeholder select keeps the data dependency)
                v.x = __shfl(mine.x, (threadIdx.x & ~3) | l); v.y = __shfl(mine.y, (threadIdx.x & ~3) | l);
                v.z = __shfl(mine.z, (threadIdx.x & ~3) | l); v.w = __shfl(mine.w, (threadIdx.x & ~3) | l);
                mix ^= v.x + v.w; acc += v.y;
            }
        }
        idx = (mix ^ (acc * 0x9E3779B1u)) % n_blk;
    }
    out[tid] = acc + idx;
}

template <int MODE>
static void run(const char *name, const uint4 *d, uint32_t n_blk, uint32_t base, uint32_t *o) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 64;
    for (int wpc : {8, 16, 32}) {
        for (int rep = 0; rep < 2; rep++) {
            const int blocks = 256 * wpc / 4;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(256), 0, 0, d, n_blk, base, steps, o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double g = (double)blocks * 256 * steps;
            if (rep) printf("%-8s footprint %.2f GB, waves/CU %2d: %.3f ms, %.2f G gathers/s, %.0f ns per dependent step\n", name, n_blk * 64.0 / 1e9, wpc, ms,
                            g / ms / 1e6, ms * 1e6 / steps);
        }
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    const uint32_t n_blk = argc > 1 ? (uint32_t)atoll(argv[1]) : 89523223;
    printf("table: %u blocks of 64 bytes = %.2f GB\n", n_blk, n_blk * 64.0 / 1e9);
    uint4 *d; uint32_t *o;
    const int contiguous = argc > 2 ? atoi(argv[2]) : 0;  // 1: hipDeviceMallocContiguous (physically contiguous: larger translation fragments?)
    if (contiguous) { CK(hipExtMallocWithFlags((void **)&d, (size_t)n_blk * 64, hipDeviceMallocContiguous)); printf("allocation: hipDeviceMallocContiguous\n"); }
    else CK(hipMalloc(&d, (size_t)n_blk * 64));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, (size_t)n_blk * 4);
    CK(hipDeviceSynchronize());
    CK(hipMalloc(&o, 256 * 32 * 64 * 4 * sizeof(uint32_t)));
    run<0>("lane64", d, n_blk, 0, o);
    run<1>("lane16", d, n_blk, 0, o);
    run<2>("quad64", d, n_blk, 0, o);
    if (contiguous) return 0;
    run<0>("window/2", d, n_blk / 2, n_blk / 4, o);
    run<0>("window/4", d, n_blk / 4, n_blk / 3, o);
    run<0>("window/8", d, n_blk / 8, n_blk / 2, o);
    return 0;
}
