// hip_util.hpp -- small device-side building blocks shared by the streaming HIP stages (finish_device.hip,
// euler_device.hip, synth_device.hip): error check, stream-ordered buffers, a three-phase exclusive scan.
// Every kernel here is `static`: each translation unit that includes the header gets its own copy.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "host_graph.hpp"

#ifndef HIP_CHECK
#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MTG_DIE("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
    } while (0)
#endif

namespace mtg {
namespace hu {

constexpr int EB = 256;  // threads per block of the element-wise kernels
inline unsigned grid_for(uint64_t n, int block = EB) { return (unsigned)((n + (uint64_t)block - 1) / (uint64_t)block); }
// global element index of a thread; 64-bit because the dart arrays pass 2^32 - 256 elements at the largest sizes
__device__ __forceinline__ uint64_t gid() { return (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; }

// ---- exclusive scan u32 -> T (u32 or u64), three phases, 2048 items per block ----
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = EB * SCAN_ITEMS;

template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *total) {
    __shared__ T wave_sum[EB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    T base = 0, all = 0;
    for (int w = 0; w < EB / 64; w++) {
        if (w < wave) base += wave_sum[w];
        all += wave_sum[w];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

template <typename T>
static __global__ __launch_bounds__(EB) void scan_reduce_kernel(const uint32_t *in, uint64_t n, T *block_sums) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    T total;
    block_exclusive_scan<T>(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
template <typename T>
static __global__ __launch_bounds__(EB) void scan_block_sums_kernel(T *block_sums, uint64_t n_blocks, T *total_out) {
    T carry = 0;
    for (uint64_t start = 0; start < n_blocks; start += EB) {
        const uint64_t i = start + threadIdx.x;
        const T v = i < n_blocks ? block_sums[i] : 0;
        T total;
        const T ex = block_exclusive_scan<T>(v, &total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}
template <typename T>
static __global__ __launch_bounds__(EB) void scan_apply_kernel(const uint32_t *in, uint64_t n, const T *block_offsets, T *out) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    T s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    T total;
    T run = block_offsets[blockIdx.x] + block_exclusive_scan<T>(s, &total);
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}
inline uint64_t scan_blocks(uint64_t n) { return (n + SCAN_CHUNK - 1) / SCAN_CHUNK; }
// out[i] = sum of in[0..i), *d_total = sum of all; `out` may alias `in` when T is uint32_t. block_sums: scan_blocks(n) + 1 elements of T.
template <typename T>
static void scan_u32(hipStream_t st, const uint32_t *in, uint64_t n, T *out, T *block_sums, T *d_total) {
    const uint64_t nb = scan_blocks(n);
    if (nb == 0) {
        HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(T), st));
        return;
    }
    scan_reduce_kernel<T><<<(unsigned)nb, EB, 0, st>>>(in, n, block_sums);
    scan_block_sums_kernel<T><<<1, EB, 0, st>>>(block_sums, nb, d_total);
    scan_apply_kernel<T><<<(unsigned)nb, EB, 0, st>>>(in, n, block_sums, out);
    HIP_CHECK(hipGetLastError());
}

// Stream-ordered allocation from the device's default memory pool (which the callers tell to keep freed memory: the work
// arrays of a call cost milliseconds to map afresh, and a driver that finishes graph after graph reuses them).
// Device work arrays of the finishing stages. Every call asks for the same sizes again, so released blocks are kept per device and
// handed back by size (the smallest that fits, at most twice the request); what is not there comes from hipMalloc. (The runtime's
// stream-ordered pool -- hipMallocAsync with the release threshold at its maximum -- was used until the end of round 3: on some
// boxes the twelve allocations at the head of the Euler decomposition then took 0.5 to 4.7 s on some steps, nothing on the others.)
// All users run on the one finish stream of their device, so a block released earlier on that stream is free when it is used again.
struct DeviceBlockCache {
    struct Kept { void *p; uint64_t gen; };
    std::mutex m;
    std::multimap<size_t, Kept> free_blocks;
    size_t held = 0;
    uint64_t gen = 1;  // number of the running call: a block given back carries it
    static constexpr size_t GRAIN = 1u << 20, HOLD_LIMIT = 192ull << 30;
    static size_t rounded(size_t bytes) { return (std::max<size_t>(bytes, 1) + GRAIN - 1) / GRAIN * GRAIN; }
    // the smallest kept block that holds `bytes` and is at most twice as large (a stage's arrays fit the blocks an earlier stage of
    // the same call gave back: the first call on a large graph allocates less from the driver, where a gigabyte costs ~50 ms on some
    // boxes); *block_bytes = the block's real size, to be handed to give()
    void *take(size_t bytes, size_t *block_bytes) {
        const size_t want = rounded(bytes);
        {
            std::lock_guard<std::mutex> lock(m);
            auto it = free_blocks.lower_bound(want);
            if (it != free_blocks.end() && it->first <= 2 * want) {
                void *p = it->second.p;
                *block_bytes = it->first;
                held -= it->first;
                free_blocks.erase(it);
                return p;
            }
        }
        *block_bytes = want;
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {  // out of memory with blocks of other sizes held back: give them up and try again
            (void)hipGetLastError();
            trim();
            HIP_CHECK(hipMalloc(&p, want));
        }
        return p;
    }
    void give(void *p, size_t block_bytes) {
        {
            std::lock_guard<std::mutex> lock(m);
            if (held + block_bytes <= HOLD_LIMIT) {
                free_blocks.emplace(block_bytes, Kept{p, gen});
                held += block_bytes;
                return;
            }
        }
        (void)hipFree(p);  // (more than HOLD_LIMIT kept already: calls of that size are trimmed at their end anyway)
    }
    // End of a call: what it did not touch goes back to the driver, so the cache never holds more than the last call's own arrays
    // (a driver that finishes graphs of different sizes would otherwise collect blocks of every size it has ever seen, out of
    // reach of the other allocators of the process).
    void end_call() {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(m);
            for (auto it = free_blocks.begin(); it != free_blocks.end();) {
                if (it->second.gen < gen) {
                    drop.push_back(it->second.p);
                    held -= it->first;
                    it = free_blocks.erase(it);
                } else ++it;
            }
            gen++;
        }
        for (void *p : drop) (void)hipFree(p);
    }
    void trim() {
        std::multimap<size_t, Kept> drop;
        {
            std::lock_guard<std::mutex> lock(m);
            drop.swap(free_blocks);
            held = 0;
        }
        for (auto &kv : drop) (void)hipFree(kv.second.p);
    }
    size_t held_bytes() {
        std::lock_guard<std::mutex> lock(m);
        return held;
    }
};
inline DeviceBlockCache &device_block_cache(int device_id) {
    static DeviceBlockCache *caches = new DeviceBlockCache[64];  // (never destroyed: buffers may be released during static destruction)
    if (device_id < 0 || device_id >= 64) MTG_DIE("device id %d out of range", device_id);
    return caches[device_id];
}

// hipMalloc for the engine's long-lived device arrays: when the device is full, the blocks the finish keeps for its next call are
// given up before the allocation is tried again.
template <typename T>
inline void device_malloc(T **p, size_t bytes) {
    if (hipMalloc((void **)p, bytes) == hipSuccess) return;
    (void)hipGetLastError();
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    device_block_cache(dev).trim();
    HIP_CHECK(hipMalloc((void **)p, bytes));
}

// One stream per device for the finishing stages, created on first use.
inline hipStream_t finish_stream(int device_id) {
    static hipStream_t streams[64] = {nullptr};
    static std::mutex mu;
    if (device_id < 0 || device_id >= 64) MTG_DIE("device id %d out of range", device_id);
    std::lock_guard<std::mutex> l(mu);
    if (!streams[device_id]) {
        HIP_CHECK(hipSetDevice(device_id));
        HIP_CHECK(hipStreamCreate(&streams[device_id]));
    }
    return streams[device_id];
}

struct Buf {
    void *p = nullptr;
    size_t bytes = 0, block_bytes = 0;
    int device = 0;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { release(); }
    void release() {
        if (p) device_block_cache(device).give(p, block_bytes);
        p = nullptr;
    }
    template <typename T>
    T *alloc(hipStream_t st, uint64_t n) {
        release();
        HIP_CHECK(hipGetDevice(&device));
        // a kept block is reused without an event in between: that is only ordered if every user works on the device's one finish stream
        if (st != finish_stream(device)) MTG_DIE("hu::Buf: work arrays of the finishing stages belong on the finish stream of device %d", device);
        bytes = (n ? n : 1) * sizeof(T);
        p = device_block_cache(device).take(bytes, &block_bytes);
        return (T *)p;
    }
    template <typename T>
    T *as() const { return (T *)p; }
};

// ---- device-resident original edges of a host graph (HostGraph::device_cache) ----
inline void edge_cache_free(DeviceEdgeCache *c) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_from);
    (void)hipFree(c->d_mirror);
    if (c->d_row0) (void)hipFree(c->d_row0);
    if (c->d_adj0) (void)hipFree(c->d_adj0);
    (void)hipSetDevice(cur);
}
// the cache of g on `device`, or null (another device, or a graph whose sizes changed: never the case after build)
inline std::mutex &edge_cache_mutex() {  // (mtg_compute_tigs_cfg builds its device copies from concurrent host threads)
    static std::mutex m;
    return m;
}
inline const DeviceEdgeCache *edge_cache_get(const HostGraph &g, int device) {
    std::lock_guard<std::mutex> l(edge_cache_mutex());
    const DeviceEdgeCache *c = g.device_cache.get();
    return (c && c->device == device && c->n_edges == g.n_original_edges && c->n_nodes == g.node_count()) ? c : nullptr;
}
// Takes ownership of two plain hipMalloc'd arrays on `device` (from-nodes of the original edges, mirror); keeps an existing
// cache of the same device and frees the offered arrays instead.
inline void edge_cache_put(const HostGraph &g, int device, uint32_t *d_from, uint32_t *d_mirror) {
    std::lock_guard<std::mutex> l(edge_cache_mutex());
    if (g.device_cache || std::getenv("MTG_NO_EDGE_CACHE")) {  // first come, first kept (one cache per graph)
        (void)hipFree(d_from);
        (void)hipFree(d_mirror);
        return;
    }
    auto *c = new DeviceEdgeCache();
    c->device = device;
    c->n_edges = g.n_original_edges;
    c->n_nodes = g.node_count();
    c->d_from = d_from;
    c->d_mirror = d_mirror;
    c->free_fn = edge_cache_free;
    g.device_cache.reset(c);
}

// Attaches the buckets of the original darts to the graph's cache on `device` (takes ownership; frees them if there is no such
// cache or it has buckets already).
inline void edge_cache_set_buckets(const HostGraph &g, int device, uint32_t *d_row0, uint32_t *d_adj0) {
    std::lock_guard<std::mutex> l(edge_cache_mutex());
    DeviceEdgeCache *c = g.device_cache.get();
    if (c && c->device == device && !c->d_row0) {
        c->d_row0 = d_row0;
        c->d_adj0 = d_adj0;
        return;
    }
    (void)hipFree(d_row0);
    (void)hipFree(d_adj0);
}

// Device -> pageable host memory through a pinned ring: while slice i + 1 crosses PCIe at full rate, host threads copy slice i out
// of the ring (a plain hipMemcpy into pageable memory is staged by the runtime on one thread: 15-20 GB/s here). Synchronises the
// stream. Small copies take the plain path.
constexpr int MAX_FINISH_DEVICES = 64;
// `put(dst_offset_bytes_of_the_source, src, n_bytes)` moves a piece of a slice from the ring to its place: a plain memcpy, or a
// conversion on the way (download_sliced_widen below)
constexpr size_t RING_SLICE = 16u << 20;
constexpr int RING_SLOTS = 4;
struct TransferRing {
    std::mutex m;  // one sliced transfer at a time per device
    char *slot[RING_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[RING_SLOTS];
    void ready() {  // (under the lock)
        if (slot[0]) return;
        for (int i = 0; i < RING_SLOTS; i++) {
            HIP_CHECK(hipHostMalloc((void **)&slot[i], RING_SLICE, hipHostMallocDefault));  // (coherent: read right after an event wait)
            HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventReleaseToSystem));
        }
    }
};
inline TransferRing &transfer_ring(int device_id) {
    static TransferRing rings[MAX_FINISH_DEVICES];
    if (device_id < 0 || device_id >= MAX_FINISH_DEVICES) MTG_DIE("sliced transfer: device id %d out of range", device_id);
    return rings[device_id];
}
// (progress, optional: called with the number of leading bytes that have reached their place, after every slice -- by one of the
// copying threads; slices complete in order)
template <typename Put>
inline void download_sliced_with(const void *d_src, size_t bytes, hipStream_t st, int device_id, Put &&put,
                                 const std::function<void(size_t)> *progress = nullptr) {
    constexpr size_t SLICE = RING_SLICE;
    constexpr int NS = RING_SLOTS;
    TransferRing &r = transfer_ring(device_id);
    std::lock_guard<std::mutex> lock(r.m);
    r.ready();
    const size_t n_slices = (bytes + SLICE - 1) / SLICE;
    const unsigned T = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<std::atomic<uint32_t>> done(n_slices);
    for (auto &x : done) x.store(0, std::memory_order_relaxed);
    std::atomic<long> recorded{-1};  // highest slice whose copy and event have been enqueued
    auto copier = [&](unsigned t) {
        for (size_t i = 0; i < n_slices; i++) {
            while (recorded.load(std::memory_order_acquire) < (long)i) std::this_thread::yield();
            HIP_CHECK(hipEventSynchronize(r.ev[i % NS]));
            const size_t off = i * SLICE, n = std::min(SLICE, bytes - off);
            const size_t a0 = (n / 8 * t / T) * 8, a1 = t + 1 == T ? n : (n / 8 * (t + 1) / T) * 8;  // (8-byte aligned shares)
            put(off + a0, r.slot[i % NS] + a0, a1 - a0);
            const uint32_t before = done[i].fetch_add(1, std::memory_order_acq_rel);
            if (progress && before + 1 == T) (*progress)(off + n);  // (the last share of slice i: every thread takes the slices in order)
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(copier, t);
    for (size_t i = 0; i < n_slices; i++) {
        if (i >= (size_t)NS)  // the slot is free again when every copier has taken its share of the slice that was in it
            while (done[i - NS].load(std::memory_order_acquire) < T) std::this_thread::yield();
        const size_t off = i * SLICE, n = std::min(SLICE, bytes - off);
        HIP_CHECK(hipMemcpyAsync(r.slot[i % NS], (const char *)d_src + off, n, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(r.ev[i % NS], st));
        recorded.store((long)i, std::memory_order_release);
    }
    for (auto &x : th) x.join();
    HIP_CHECK(hipStreamSynchronize(st));
}
inline void download_sliced(void *dst, const void *d_src, size_t bytes, hipStream_t st, int device_id) {
    if (bytes < (64u << 20)) {
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return;
    }
    download_sliced_with(d_src, bytes, st, device_id, [dst](size_t off, const char *src, size_t n) { std::memcpy((char *)dst + off, src, n); });
}
// n 32-bit words on the device -> n 64-bit words on the host: half the bytes over PCIe, widened by the threads that empty the ring
inline void download_sliced_widen(uint64_t *dst, const uint32_t *d_src, size_t n, hipStream_t st, int device_id) {
    if (n * 4 < (64u << 20)) {
        std::vector<uint32_t> tmp(n);
        if (n) HIP_CHECK(hipMemcpyAsync(tmp.data(), d_src, n * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (size_t i = 0; i < n; i++) dst[i] = tmp[i];
        return;
    }
    download_sliced_with(d_src, n * 4, st, device_id, [dst](size_t off, const char *src, size_t bytes) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
        uint64_t *out = dst + off / 4;
        for (size_t i = 0; i < bytes / 4; i++) out[i] = w[i];
    });
}

// Pageable host memory -> device through the same pinned ring, the other way round: host threads fill slice i + 1 while slice i
// crosses PCIe (a plain hipMemcpy from pageable memory is staged by the runtime on one thread: 9 GB/s measured for the 1.7 GB of a
// 2^27 graph's edge arrays -- 190 of the 360 ms of a cold device-graph build). Synchronises the stream.
inline void upload_sliced(void *d_dst, const void *src, size_t bytes, hipStream_t st, int device_id) {
    constexpr size_t SLICE = RING_SLICE;
    constexpr int NS = RING_SLOTS;
    if (bytes < 4 * SLICE) {
        if (bytes) HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return;
    }
    TransferRing &r = transfer_ring(device_id);
    std::lock_guard<std::mutex> lock(r.m);
    r.ready();
    const size_t n_slices = (bytes + SLICE - 1) / SLICE;
    const unsigned T = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<std::atomic<uint32_t>> filled(n_slices);
    for (auto &x : filled) x.store(0, std::memory_order_relaxed);
    std::atomic<long> sent{-1};  // highest slice whose copy and event have been enqueued
    auto filler = [&](unsigned t) {
        for (size_t i = 0; i < n_slices; i++) {
            if (i >= (size_t)NS) {  // the slot is free again when the DMA of the slice that was in it has finished
                while (sent.load(std::memory_order_acquire) < (long)(i - NS)) std::this_thread::yield();
                HIP_CHECK(hipEventSynchronize(r.ev[i % NS]));
            }
            const size_t off = i * SLICE, n = std::min(SLICE, bytes - off);
            const size_t a0 = (n / 8 * t / T) * 8, a1 = t + 1 == T ? n : (n / 8 * (t + 1) / T) * 8;
            std::memcpy(r.slot[i % NS] + a0, (const char *)src + off + a0, a1 - a0);
            filled[i].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(filler, t);
    for (size_t i = 0; i < n_slices; i++) {
        while (filled[i].load(std::memory_order_acquire) < T) std::this_thread::yield();
        const size_t off = i * SLICE, n = std::min(SLICE, bytes - off);
        HIP_CHECK(hipMemcpyAsync((char *)d_dst + off, r.slot[i % NS], n, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipEventRecord(r.ev[i % NS], st));
        sent.store((long)i, std::memory_order_release);
    }
    for (auto &x : th) x.join();
    HIP_CHECK(hipStreamSynchronize(st));
}

// End of a finishing call: its work arrays stay with the library for the next call (what the call did not touch goes back to the
// driver) while the call worked on less than a third of the device's memory -- BASELINE configs[3] at its nominal size (2^30: ~80 GB
// by this estimate) iterates without asking the driver for tens of gigabytes per call, which costs between 0.2 and 5 s there --;
// beyond that everything goes back. mtg_release_device_memory returns what is held at any time.
inline void finish_trim(int device_id, uint64_t bytes_used) {
    static uint64_t threshold[64] = {0};
    if (device_id >= 0 && device_id < 64 && !threshold[device_id]) {
        size_t free_b = 0, total_b = 0;
        threshold[device_id] = hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? (uint64_t)total_b / 3 : (32ull << 30);
    }
    const uint64_t limit = device_id >= 0 && device_id < 64 ? threshold[device_id] : (32ull << 30);
    if (bytes_used < limit) device_block_cache(device_id).end_call();  // (keeps this call's arrays, frees what it did not touch)
    else device_block_cache(device_id).trim();
}

}  // namespace hu
}  // namespace mtg
