// Calibration micro-benchmark, round 2: rate of DEPENDENT random gathers of 16, 32 or 64 bytes per lane from 64-byte aligned
// blocks of a table larger than the Infinity Cache (one, two or four 16-byte loads per lane and step). Answers what the family
// blocks of the SSSP enumeration level raised: is the random-access ceiling per 64-byte LINE, per 32-byte sector or per
// lane-request?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LOADS>
__global__ void chase(const uint4 *tab, uint32_t n_blk, int steps, uint32_t *out) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_blk;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4 *p = tab + (size_t)idx * 4;   // 64-byte block
        uint32_t mix = 0;
#pragma unroll
        for (int l = 0; l < LOADS; l++) { const uint4 v = p[l]; mix ^= v.x + v.w; acc += v.y; }
        idx = (mix ^ (acc * 0x9E3779B1u)) % n_blk;   // next index depends on the loaded data
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

template <int LOADS>
static void run(const uint4 *d, uint32_t n_blk, uint32_t *o) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 64;
    for (int wpc : {4, 8, 12, 16, 32}) {
        for (int rep = 0; rep < 2; rep++) {
            const int blocks = 256 * wpc / 4;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase<LOADS>, dim3(blocks), dim3(256), 0, 0, d, n_blk, steps, o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double g = (double)blocks * 256 * steps;
            if (rep) printf("%2d bytes per gather, waves/CU %2d: %.3f ms, %.2f G gathers/s, %.2f G lane-requests/s, %.0f ns per dependent step\n",
                            16 * LOADS, wpc, ms, g / ms / 1e6, g * LOADS / ms / 1e6, ms * 1e6 / steps);
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t n_blk = argc > 1 ? (uint32_t)atoll(argv[1]) : 11190402;  // default: 716 MB of 64-byte blocks (the 2^24 graph); 89523223 = the 2^27 graph (5.7 GB)
    printf("table: %u blocks of 64 bytes = %.2f GB\n", n_blk, n_blk * 64.0 / 1e9);
    std::vector<uint4> h((size_t)n_blk * 4);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32); v.z = v.x * 3; v.w = v.y * 5; }
    uint4 *d; uint32_t *o;
    CK(hipMalloc(&d, h.size() * sizeof(uint4)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CK(hipMalloc(&o, 256 * 32 * 64 * 4 * sizeof(uint32_t)));
    run<1>(d, n_blk, o);
    run<2>(d, n_blk, o);
    run<4>(d, n_blk, o);
    return 0;
}
