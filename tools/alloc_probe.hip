// alloc_probe.hip -- what device memory costs on THIS box: hipMalloc / first touch / hipFree by size, repeated, and the same through
// the virtual-memory API. The boxes of one pool differ by two orders of magnitude here (the device arena).
// build: hipcc -O2 --offload-arch=gfx950 tools/alloc_probe.hip -o tools/alloc_probe ; run: tools/alloc_probe [max_gb=16]
#include <hip/hip_runtime.h>

#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e = (x);                                                                     \
        if (e != hipSuccess) {                                                                  \
            std::fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorName(e), __LINE__); \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch_kernel(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}

static void touch(void *p, size_t bytes, const char *what) {
    const double t0 = now_ms();
    touch_kernel<<<4096, 256>>>((uint4 *)p, bytes / 16);
    CK(hipDeviceSynchronize());
    const double t1 = now_ms();
    touch_kernel<<<4096, 256>>>((uint4 *)p, bytes / 16);
    CK(hipDeviceSynchronize());
    const double t2 = now_ms();
    std::printf("    %s: first touch %.2f ms, second %.2f ms (%.0f GB/s)\n", what, t1 - t0, t2 - t1, bytes / 1e6 / (t2 - t1));
}

int main(int argc, char **argv) {
    const size_t max_gb = argc > 1 ? (size_t)std::atoi(argv[1]) : 16;
    double t0 = now_ms();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    std::printf("runtime init %.1f ms\n", now_ms() - t0);
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    std::printf("HBM: %.1f GB free of %.1f GB\n", fr / 1e9, tot / 1e9);
    for (int rep = 0; rep < 3; rep++) {
        std::printf("-- pass %d: hipMalloc / touch / hipFree by size\n", rep);
        for (size_t gb = 1; gb <= max_gb; gb *= 2) {
            const size_t bytes = gb << 30;
            void *p = nullptr;
            t0 = now_ms();
            CK(hipMalloc(&p, bytes));
            const double t1 = now_ms();
            std::printf("  %2zu GB: hipMalloc %8.2f ms (%.1f ms/GB)\n", gb, t1 - t0, (t1 - t0) / gb);
            if (rep == 0) touch(p, bytes, "hipMalloc'd");
            t0 = now_ms();
            CK(hipFree(p));
            std::printf("         hipFree   %8.2f ms\n", now_ms() - t0);
        }
    }
    {
        // does waiting help? free 16 GB (touched), sleep, then time allocations of 4 GB
        std::printf("-- free 16 GB, wait, allocate 4 GB three times\n");
        for (double wait_s : {0.0, 0.25, 1.0, 3.0}) {
            void *big = nullptr;
            CK(hipMalloc(&big, 16ull << 30));
            touch_kernel<<<4096, 256>>>((uint4 *)big, (16ull << 30) / 16);
            CK(hipDeviceSynchronize());
            CK(hipFree(big));
            const double t_free = now_ms();
            if (wait_s > 0) {
                struct timespec ts = {(time_t)wait_s, (long)((wait_s - (time_t)wait_s) * 1e9)};
                nanosleep(&ts, nullptr);
            }
            double worst = 0, total = 0;
            for (int i = 0; i < 3; i++) {
                void *p = nullptr;
                t0 = now_ms();
                CK(hipMalloc(&p, 4ull << 30));
                const double dt = now_ms() - t0;
                worst = dt > worst ? dt : worst;
                total += dt;
                touch_kernel<<<4096, 256>>>((uint4 *)p, (4ull << 30) / 16);
                CK(hipDeviceSynchronize());
                CK(hipFree(p));
            }
            std::printf("  waited %.2f s: three hipMalloc(4 GB) took %.1f ms in all (worst %.1f ms); %.2f s after the free\n", wait_s, total, worst, (now_ms() - t_free) / 1e3);
        }
    }
    {
        std::printf("-- many small: 64 x 256 MB\n");
        std::vector<void *> ps(64);
        t0 = now_ms();
        for (auto &p : ps) CK(hipMalloc(&p, 256u << 20));
        const double t1 = now_ms();
        for (auto &p : ps) CK(hipFree(p));
        std::printf("  64 x hipMalloc %.2f ms, 64 x hipFree %.2f ms\n", t1 - t0, now_ms() - t1);
    }
    {
        std::printf("-- virtual memory API: reserve 16 GB of addresses, then create + map + set access per 2 GB\n");
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) {
            std::printf("  not supported here\n");
            return 0;
        }
        std::printf("  granularity %zu\n", gran);
        const size_t total = 16ull << 30, piece = 2ull << 30;
        void *va = nullptr;
        t0 = now_ms();
        if (hipMemAddressReserve(&va, total, gran, nullptr, 0) != hipSuccess) { std::printf("  reserve failed\n"); return 0; }
        std::printf("  address reserve %.2f ms\n", now_ms() - t0);
        std::vector<hipMemGenericAllocationHandle_t> hs;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t off = 0; off < total; off += piece) {
            hipMemGenericAllocationHandle_t h;
            t0 = now_ms();
            if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { std::printf("  create failed\n"); break; }
            const double t1 = now_ms();
            CK(hipMemMap((char *)va + off, piece, 0, h, 0));
            const double t2 = now_ms();
            CK(hipMemSetAccess((char *)va + off, piece, &acc, 1));
            const double t3 = now_ms();
            std::printf("  2 GB at +%2zu GB: create %.2f ms, map %.2f ms, set access %.2f ms\n", off >> 30, t1 - t0, t2 - t1, t3 - t2);
            hs.push_back(h);
        }
        if (!hs.empty()) touch(va, hs.size() * piece, "mapped range");
        t0 = now_ms();
        for (size_t i = 0; i < hs.size(); i++) {
            CK(hipMemUnmap((char *)va + i * piece, piece));
            CK(hipMemRelease(hs[i]));
        }
        CK(hipMemAddressFree(va, total));
        std::printf("  unmap + release + free %.2f ms\n", now_ms() - t0);
    }
    return 0;
}
