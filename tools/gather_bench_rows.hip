// Calibration micro-benchmark, round 6: does the gather of a 64-byte block by FOUR LANES stay at the quad form's rate when the four
// lanes are not neighbours? gfx950 has v_permlane16_swap / v_permlane32_swap: with the four lanes of a group 16 apart (lane l, l+16,
// l+32, l+48) the 4 x 4 transposition of the quarters is 16 instructions instead of the 128 of the DPP form (quad_perm moves + selects).
//   quad64   lanes 4i..4i+3 load one block together; the enumeration level's transposition (DPP moves + selects)
//   row64    lanes l, l+16, l+32, l+48 load one block together; transposition by the two swap instructions
//   verify   both forms against a plain load of the lane's own block
// usage: gather_bench_rows [n_blocks = 89523223]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill(uint4 *tab, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        tab[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)x * 3u, (uint32_t)(x >> 32) * 5u);
    }
}

__device__ __forceinline__ void swap32(uint32_t &a, uint32_t &b) {  // a[lanes 32..63] <-> b[lanes 0..31]
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap16(uint32_t &a, uint32_t &b) {  // a[odd rows of 16] <-> b[even rows]
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap32x4(uint4 &a, uint4 &b) { swap32(a.x, b.x); swap32(a.y, b.y); swap32(a.z, b.z); swap32(a.w, b.w); }
__device__ __forceinline__ void swap16x4(uint4 &a, uint4 &b) { swap16(a.x, b.x); swap16(a.y, b.y); swap16(a.z, b.z); swap16(a.w, b.w); }

// the block of `idx` of every lane, gathered by the lane's group. ROWS: group = lanes with the same (lane & 15), member = lane >> 4.
template <bool ROWS>
__device__ __forceinline__ void gather(const uint4 *tab, uint32_t idx, uint4 &b0, uint4 &b1, uint4 &b2, uint4 &b3) {
    const int lane = threadIdx.x & 63;
    uint4 g[4];
    if (ROWS) {
        const int r = lane >> 4;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t bi = __shfl(idx, (lane & 15) | (l << 4));
            g[l] = tab[(size_t)bi * 4 + r];
        }
        swap32x4(g[0], g[2]); swap32x4(g[1], g[3]);
        swap16x4(g[0], g[1]); swap16x4(g[2], g[3]);
        b0 = g[0]; b1 = g[1]; b2 = g[2]; b3 = g[3];
    } else {
        const int q = lane & 3;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t bi = __shfl(idx, (lane & ~3) | l);
            g[l] = tab[(size_t)bi * 4 + q];
        }
        // the enumeration level's transposition: two exchange stages (lane ^ 1, lane ^ 2) of DPP moves + selects
        b0 = g[0]; b1 = g[1]; b2 = g[2]; b3 = g[3];
        const bool odd1 = (lane & 1) != 0, odd2 = (lane & 2) != 0;
        auto xchg = [](uint32_t &lo, uint32_t &hi, bool odd, auto ctrl) {
            constexpr int C = decltype(ctrl)::value;
            const uint32_t from_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, C, 0xF, 0xF, false);
            const uint32_t from_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, C, 0xF, 0xF, false);
            lo = odd ? from_hi : lo;
            hi = odd ? hi : from_lo;
        };
        auto xchg4 = [&](uint4 &lo, uint4 &hi, bool odd, auto ctrl) {
            xchg(lo.x, hi.x, odd, ctrl); xchg(lo.y, hi.y, odd, ctrl); xchg(lo.z, hi.z, odd, ctrl); xchg(lo.w, hi.w, odd, ctrl);
        };
        xchg4(b0, b1, odd1, std::integral_constant<int, 0xB1>{});
        xchg4(b2, b3, odd1, std::integral_constant<int, 0xB1>{});
        xchg4(b0, b2, odd2, std::integral_constant<int, 0x4E>{});
        xchg4(b1, b3, odd2, std::integral_constant<int, 0x4E>{});
    }
}

template <bool ROWS>
__global__ void chase(const uint4 *tab, uint32_t n_blk, int steps, uint32_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = tid * 2654435761u % n_blk;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        uint4 b0, b1, b2, b3;
        gather<ROWS>(tab, idx, b0, b1, b2, b3);
        const uint32_t mix = (b0.x + b0.w) ^ (b1.x + b1.w) ^ (b2.x + b2.w) ^ (b3.x + b3.w);
        acc += b0.y + b1.y + b2.y + b3.y;
        idx = (mix ^ (acc * 0x9E3779B1u)) % n_blk;
    }
    out[tid] = acc + idx;
}

template <bool ROWS>
__global__ void verify(const uint4 *tab, uint32_t n_blk, uint32_t *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t idx = (tid * 2654435761u + 12345u) % n_blk;
    uint4 b0, b1, b2, b3;
    gather<ROWS>(tab, idx, b0, b1, b2, b3);
    const uint4 *p = tab + (size_t)idx * 4;
    auto eq = [](uint4 a, uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; };
    if (!(eq(b0, p[0]) && eq(b1, p[1]) && eq(b2, p[2]) && eq(b3, p[3]))) atomicAdd(bad, 1u);
}

template <bool ROWS>
static void run(const char *name, const uint4 *d, uint32_t n_blk, uint32_t *o) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 64;
    for (int wpc : {8, 16, 32}) {
        for (int rep = 0; rep < 2; rep++) {
            const int blocks = 256 * wpc / 4;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase<ROWS>, dim3(blocks), dim3(256), 0, 0, d, n_blk, steps, o);
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
    uint4 *d; uint32_t *o, *bad;
    CK(hipMalloc(&d, (size_t)n_blk * 64));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, (size_t)n_blk * 4);
    CK(hipDeviceSynchronize());
    CK(hipMalloc(&o, 256 * 32 * 64 * 4 * sizeof(uint32_t)));
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(verify<false>, dim3(1024), dim3(256), 0, 0, d, n_blk, bad);
    hipLaunchKernelGGL(verify<true>, dim3(1024), dim3(256), 0, 0, d, n_blk, bad + 1);
    uint32_t h[2]; CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("verify: quad64 %u, row64 %u mismatches (of %u lanes)\n", h[0], h[1], 1024 * 256);
    run<false>("quad64", d, n_blk, o);
    run<true>("row64", d, n_blk, o);
    run<false>("quad64", d, n_blk, o);
    run<true>("row64", d, n_blk, o);
    return 0;
}
