// Calibration micro-benchmark: rate of DEPENDENT random 32-byte gathers (the access pattern of the SSSP kernels)
// from a table larger than the Infinity Cache, all 64 lanes busy. Prints gathers/s for several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// `alu` = dependent integer operations per step between a gather's arrival and the next gather's issue (0 = pure chase):
// shows how much gather throughput a kernel with that much per-step work can reach at a given occupancy.
__global__ void chase(const uint4 *tab, uint32_t n_rec, int steps, uint32_t *out, int alu) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_rec;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4 lo = tab[(size_t)idx * 2];
        const uint4 hi = tab[(size_t)idx * 2 + 1];
        acc += lo.y + hi.w;
        for (int a = 0; a < alu; a++) acc = acc * 0x9E3779B1u + (acc >> 7);  // 3 dependent VALU operations
        idx = (lo.x ^ hi.x ^ (acc * 0x9E3779B1u)) % n_rec;   // next index depends on the loaded data
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main() {
    const uint32_t n_rec = 11190402;  // same record count as the bench graph (358 MB)
    std::vector<uint4> h((size_t)n_rec * 2);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32); v.z = v.x * 3; v.w = v.y * 5; }
    uint4 *d; uint32_t *o;
    CK(hipMalloc(&d, h.size() * sizeof(uint4)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CK(hipMalloc(&o, 256 * 32 * 64 * 4 * sizeof(uint32_t)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 64;
    for (int alu : {0, 40, 80, 120})
    for (int wpc : {4, 8, 12, 16, 24, 32}) {          // waves per CU
        for (int rep = 0; rep < 2; rep++) {
            const int blocks = 256 * wpc / 4;    // 256-thread blocks
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, o, alu);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double g = (double)blocks * 256 * steps;
            if (rep) printf("alu %3d x3, waves/CU %2d: %.3f ms, %.2f G gathers/s, %.2f TB/s of 64-B lines, %.0f ns per dependent step\n", alu, wpc, ms,
                            g / ms / 1e6, g * 64 / ms / 1e9, ms * 1e6 / steps);
        }
    }
    return 0;
}
