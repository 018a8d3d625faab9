// arena_test.hip -- stress test of the library's device arena (matchtigs_amd/csrc/hip_util.hpp: DeviceArena), compiled by
// tests/test_gpu_arena.py with hipcc and run on the GPU box. Random allocations and frees of both kinds (plain = hipFree semantics,
// Buf = stream-ordered on the finish stream) against a shadow model: ranges never overlap, lie inside chunks, keep their content
// (a per-range pattern written and checked by kernels), everything coalesces back to whole chunks, release returns them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "hip_util.hpp"

using namespace mtg;
using namespace mtg::hu;

__global__ void fill_kernel(uint32_t *p, size_t n_words, uint32_t tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (uint32_t)i;
}
__global__ void check_kernel(const uint32_t *p, size_t n_words, uint32_t tag, unsigned long long *bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != (tag ^ (uint32_t)i)) atomicAdd(bad, 1ull);
}

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::fprintf(stderr, "arena_test: %s failed (line %d)\n", #c, __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

struct Live { char *p; size_t bytes; uint32_t tag; bool buf; };

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1;
    const int ops = argc > 2 ? std::atoi(argv[2]) : 4000;
    HIP_CHECK(hipSetDevice(0));
    DeviceArena &a = device_arena(0);
    hipStream_t fs = finish_stream(0);
    unsigned long long *d_bad = nullptr;
    HIP_CHECK(hipMalloc(&d_bad, 8));
    HIP_CHECK(hipMemset(d_bad, 0, 8));
    std::mt19937_64 rng(seed);
    std::vector<Live> live;
    size_t live_bytes = 0;
    a.reserve(256u << 20);  // one chunk up front, like a call's reservation; the rest grows on demand
    auto overlaps = [&](char *p, size_t n) {
        for (const Live &l : live)
            if (p < l.p + l.bytes && l.p < p + n) return true;
        return false;
    };
    for (int op = 0; op < ops; op++) {
        const bool do_alloc = live.empty() || (live_bytes < (1ull << 30) && (rng() % 100) < 55);
        if (do_alloc) {
            // sizes from 1 byte to 96 MB, skewed small; both allocation kinds
            const int cls = (int)(rng() % 10);
            size_t bytes = cls < 4 ? 1 + rng() % 4096 : cls < 8 ? 1 + rng() % (4u << 20) : 1 + rng() % (96u << 20);
            const bool buf = (rng() & 1) != 0;
            char *p = (char *)a.alloc(bytes, buf);
            REQUIRE(p != nullptr);
            REQUIRE(((uintptr_t)p & 255u) == 0);
            REQUIRE(a.chunk_of(p) != nullptr && a.chunk_of(p) == a.chunk_of(p + bytes - 1));
            REQUIRE(!overlaps(p, bytes));
            const uint32_t tag = (uint32_t)rng();
            fill_kernel<<<64, 256, 0, buf ? fs : nullptr>>>((uint32_t *)p, bytes / 4, tag);
            live.push_back(Live{p, bytes, tag, buf});
            live_bytes += DeviceArena::rounded(bytes);
        } else {
            const size_t i = rng() % live.size();
            const Live l = live[i];
            check_kernel<<<64, 256, 0, l.buf ? fs : nullptr>>>((const uint32_t *)l.p, l.bytes / 4, l.tag, d_bad);
            if (l.buf) a.free(l.p, true);  // (still in use by the check queued on the finish stream: "dirty")
            else device_free(l.p);
            live[i] = live.back();
            live.pop_back();
            live_bytes -= DeviceArena::rounded(l.bytes);
        }
        REQUIRE(a.live_bytes == live_bytes);
    }
    for (const Live &l : live) {
        check_kernel<<<64, 256, 0, l.buf ? fs : nullptr>>>((const uint32_t *)l.p, l.bytes / 4, l.tag, d_bad);
        if (l.buf) a.free(l.p, true);
        else device_free(l.p);
    }
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    HIP_CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    REQUIRE(bad == 0);  // no range was handed out twice while it was in use, no write landed in a neighbour
    REQUIRE(a.live_bytes == 0 && a.live.empty());
    // everything coalesced: exactly one free range per chunk, covering it
    REQUIRE(a.free_ranges.size() == a.chunks.size());
    for (auto &c : a.chunks) {
        auto it = a.free_ranges.find(c.base);
        REQUIRE(it != a.free_ranges.end() && it->second.bytes == c.bytes);
    }
    const size_t chunks_before = a.chunks.size(), reclaim = a.reclaimable_bytes();
    REQUIRE(reclaim == a.chunk_bytes);
    a.release_free_chunks(true);
    REQUIRE(a.chunks.empty() && a.chunk_bytes == 0 && a.reclaimable_bytes() == 0);
    std::printf("arena_test ok: seed %u, %d operations, %zu chunks at the end, peak %zu MB live\n", seed, ops, chunks_before, a.peak_bytes >> 20);
    return 0;
}
