#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// one 16-byte record per gather
__global__ void chase16(const uint4 *tab, uint32_t n_rec, int steps, uint32_t *out) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_rec;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4 lo = tab[idx];
        acc += lo.y;
        idx = (lo.x ^ (acc * 0x9E3779B1u)) % n_rec;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}
int main() {
    for (uint32_t n_rec : {11190402u, 2 * 11190402u, 89523223u}) {
        std::vector<uint4> h((size_t)n_rec);
        uint64_t x = 88172645463325252ull;
        for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32); v.z = v.x * 3; v.w = v.y * 5; }
        uint4 *d; uint32_t *o;
        CK(hipMalloc(&d, h.size() * sizeof(uint4)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
        CK(hipMalloc(&o, 256 * 32 * 64 * 4 * sizeof(uint32_t)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int steps = 64;
        for (int wpc : {4, 8, 12, 16}) {
            for (int rep = 0; rep < 3; rep++) {
                const int blocks = 256 * wpc / 4;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(chase16, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, o);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double g = (double)blocks * 256 * steps;
                if (rep == 2) printf("16-B records, table %.0f MB, waves/CU %2d: %.3f ms, %.2f G gathers/s\n", n_rec * 16.0 / 1e6, wpc, ms, g / ms / 1e6);
            }
        }
        CK(hipFree(d)); CK(hipFree(o));
    }
    return 0;
}
