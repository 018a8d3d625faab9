// hip_util.hpp -- small device-side building blocks shared by the streaming HIP stages (finish_device.hip,
// euler_device.hip, synth_device.hip): error check, stream-ordered buffers, a three-phase exclusive scan.
// Every kernel here is `static`: each translation unit that includes the header gets its own copy.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <future>
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

// Tuning of the finishing stages for measurements and tests (mtg_set_finish_tuning, include/mtg_engine.h): process-wide, read at
// the start of a call, never changes a result. The library reads no environment variable for any of it.
struct FinishTuning {
    std::atomic<int> records{0};           // walk records of the reference-order mode: 0 = the engine's choice, 1 = 32-byte, 2 = 128-byte, 3 = 256-byte
    std::atomic<int> flags{0};             // bit 0: the walk waits for all of its records; bit 1: never page-lock the record arena;
                                           // bit 2: keep nothing of the graph on the device between calls (edges, mirror, buckets)
                                           // bit 3: a trivial kernel every 2 ms while the host walks (measurement: what the GPU's idle state costs the next step)
    std::atomic<long> record_delay_us{0};  // tests: slows the arrival of the walk's records
};
enum : int { FT_NO_RECORD_OVERLAP = 1, FT_NO_PIN = 2, FT_NO_EDGE_CACHE = 4, FT_KEEP_AWAKE = 8, FT_NO_CUT_FIRST = 16 };
inline FinishTuning &finish_tuning() {
    static FinishTuning t;
    return t;
}

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

// ---- device memory: one arena per device ----------------------------------------------------------------------------------
// Every device array of the library -- device graph, search and replay work arrays, the finishing stages' arrays, the graph's kept
// edge arrays -- is a range of a few large hipMalloc'd CHUNKS, handed out by a best-fit free list with coalescing. Why: a hipMalloc
// normally takes 0.3 ms whatever its size, but single hipMalloc / hipFree calls sporadically stall for 0.5 to 5 s on the shared hosts
// of this pool (tools/alloc_probe.hip) -- and until round 4 a call of the path made ~200 of them, every stage taking
// its arrays from the driver and giving them back for the next stage to ask for again (0.44 s of waits in the driver's cold step of
// round 4), all of which a one-shot caller -- the only kind the reference has, clib.rs:291 -- pays. With the arena a call's stages
// reuse each other's memory (the device graph's blocks become the finish's dart arrays), the first chunk is sized for the whole call
// from (V, E) and reserved by ONE hipMalloc on a helper thread while the host still builds the graph (device_reserve_async), and a
// caller that iterates makes no driver call at all after its first call.
// Ordering: a range released by a hu::Buf may still be in use by work queued on the device's finish stream ("dirty"); a Buf
// allocation takes it as it is (same stream: ordered), any other allocation synchronises that stream first. device_free() keeps
// hipFree's meaning (the device is idle when it returns).
hipStream_t finish_stream(int device_id);
struct DeviceArena {
    struct Chunk { char *base; size_t bytes; size_t live = 0, peak_live = 0; };
    struct Range { size_t bytes; uint64_t dirty; };  // dirty: 0 = idle, else the generation in which it was released with work still queued
    uint64_t dirty_gen = 0;
    std::mutex m;
    std::vector<Chunk> chunks;
    std::map<char *, Range> free_ranges;                  // by address (coalescing)
    std::multimap<size_t, char *> free_by_size;           // best fit
    std::map<char *, size_t> live;                        // allocated ranges
    size_t live_bytes = 0, peak_bytes = 0, chunk_bytes = 0;
    uint64_t n_chunk_allocs = 0;
    int device = 0;
    size_t device_total = 0;           // bytes of HBM on the device (first use)
    std::shared_future<void> pending;  // a reservation under way on a helper thread (device_reserve_async): allocations wait for it instead of growing
    static constexpr size_t SMALL = 4096, BIG = 2u << 20;
    static size_t rounded(size_t bytes) {
        bytes = std::max<size_t>(bytes, 1);
        const size_t g = bytes >= BIG ? BIG : SMALL;  // (large arrays start and end on 2-MB boundaries: whole translation fragments)
        return (bytes + g - 1) / g * g;
    }
    void erase_size_entry(size_t bytes, char *p) {
        auto r = free_by_size.equal_range(bytes);
        for (auto it = r.first; it != r.second; ++it)
            if (it->second == p) { free_by_size.erase(it); return; }
    }
    void insert_free(char *p, size_t bytes, uint64_t dirty) {  // (under the lock) with coalescing inside the chunk
        Chunk *c = chunk_of(p);
        auto next = free_ranges.lower_bound(p);
        if (next != free_ranges.end() && p + bytes == next->first && chunk_of(next->first) == c) {
            bytes += next->second.bytes;
            dirty = std::max(dirty, next->second.dirty);
            erase_size_entry(next->second.bytes, next->first);
            next = free_ranges.erase(next);
        }
        if (next != free_ranges.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second.bytes == p && chunk_of(prev->first) == c) {
                p = prev->first;
                bytes += prev->second.bytes;
                dirty = std::max(dirty, prev->second.dirty);
                erase_size_entry(prev->second.bytes, prev->first);
                free_ranges.erase(prev);
            }
        }
        free_ranges[p] = Range{bytes, dirty};
        free_by_size.emplace(bytes, p);
    }
    Chunk *chunk_of(const void *p) {
        for (Chunk &c : chunks)
            if ((const char *)p >= c.base && (const char *)p < c.base + c.bytes) return &c;
        return nullptr;
    }
    // A chunk for a whole call is only taken while it leaves the device mostly free (up to a third of its memory: BASELINE configs[3]
    // at its nominal size, 2^30 edges, is 90 GB of 309): the stand-in of configs[4] (2^31 edges) fills most of the HBM, and a chunk sized
    // for the call's peak would sit half empty beside the other allocators of the process. Beyond that size every array gets a chunk
    // of its own, exactly as large as it is.
    size_t whole_call_limit() {
        if (!device_total) {  // (of THIS arena's device, whichever is current)
            size_t free_b = 0, total_b = 0;
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (cur != device) (void)hipSetDevice(device);
            device_total = hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? total_b : (size_t)(64ull << 30);
            if (cur != device) (void)hipSetDevice(cur);
        }
        return device_total / 3;
    }
    bool add_chunk(size_t bytes) {  // (under the lock)
        bytes = (bytes + BIG - 1) / BIG * BIG;
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        Chunk c;
        c.base = (char *)p;
        c.bytes = bytes;
        chunks.push_back(c);
        chunk_bytes += bytes;
        n_chunk_allocs++;
        free_ranges[(char *)p] = Range{bytes, 0};
        free_by_size.emplace(bytes, (char *)p);
        return true;
    }
    // entirely free chunks go back to the driver (all == false: only those the calls since the last end_call() used less than a
    // quarter of, so that a caller that moves on to smaller graphs does not sit on the largest one's memory)
    void release_free_chunks(bool all) {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t i = 0; i < chunks.size();) {
                Chunk &c = chunks[i];
                auto it = free_ranges.find(c.base);
                const bool empty = c.live == 0 && it != free_ranges.end() && it->second.bytes == c.bytes;
                if (empty && (all || c.peak_live < c.bytes / 4)) {
                    erase_size_entry(c.bytes, c.base);
                    free_ranges.erase(it);
                    drop.push_back(c.base);
                    chunk_bytes -= c.bytes;
                    chunks.erase(chunks.begin() + (long)i);
                } else i++;
            }
            if (!all) for (Chunk &c : chunks) c.peak_live = c.live;
        }
        if (all) { std::lock_guard<std::mutex> lock(m); peak_bytes = live_bytes; }
        if (drop.empty()) return;
        int cur = 0;  // (the arena's device, not whichever is current: the chunks may hold ranges released with work still queued there)
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (void *p : drop) (void)hipFree(p);
        if (cur != device) (void)hipSetDevice(cur);
    }
    // `for_finish_stream`: the caller works on the device's finish stream only (hu::Buf)
    void *alloc(size_t bytes, bool for_finish_stream, size_t grow_hint = 0) {
        const size_t want = rounded(bytes);
        bool sync_finish = false;
        uint64_t sync_mark = 0;
        void *out = nullptr;
        for (int attempt = 0; attempt < 3 && !out; attempt++) {
            {
                std::unique_lock<std::mutex> lock(m);
                auto it = free_by_size.lower_bound(want);
                if (it != free_by_size.end()) {
                    char *p = it->second;
                    const size_t have = it->first;
                    const Range r = free_ranges[p];
                    free_by_size.erase(it);
                    free_ranges.erase(p);
                    // a large request takes the END of the range when the range is much larger (long-lived small arrays collect at the
                    // front of a chunk, the big work arrays at its back)
                    char *mine = p;
                    if (have > want) {
                        const bool at_end = want >= BIG && ((size_t)(p + have - want) % BIG) == 0;
                        if (at_end) {
                            mine = p + have - want;
                            free_ranges[p] = Range{have - want, r.dirty};
                            free_by_size.emplace(have - want, p);
                        } else {
                            free_ranges[p + want] = Range{have - want, r.dirty};
                            free_by_size.emplace(have - want, p + want);
                        }
                    }
                    live[mine] = want;
                    live_bytes += want;
                    peak_bytes = std::max(peak_bytes, live_bytes);
                    Chunk *c = chunk_of(mine);
                    c->live += want;
                    c->peak_live = std::max(c->peak_live, c->live);
                    if (r.dirty && !for_finish_stream) { sync_finish = true; sync_mark = dirty_gen; }
                    out = mine;
                    break;
                }
                if (pending.valid()) {  // a chunk is on its way
                    std::shared_future<void> f = pending;
                    pending = std::shared_future<void>();
                    lock.unlock();
                    f.wait();
                    lock.lock();
                    attempt--;
                    continue;
                }
                if (grow_hint > whole_call_limit()) grow_hint = 0;
                if (attempt == 0 && add_chunk(std::max(want, grow_hint))) continue;
                if (attempt == 0 && grow_hint > want && add_chunk(want)) continue;
            }
            if (attempt == 0) { release_free_chunks(true); continue; }  // the device is full: give up what is held and try once more
            if (attempt == 1) {
                std::lock_guard<std::mutex> lock(m);
                if (!add_chunk(want)) MTG_DIE("out of device memory: %zu bytes on device %d (the library holds %zu in %zu chunks)", want, device, chunk_bytes, chunks.size());
            }
        }
        if (!out) MTG_DIE("out of device memory: %zu bytes on device %d", want, device);
        if (sync_finish) {
            HIP_CHECK(hipStreamSynchronize(finish_stream(device)));
            std::lock_guard<std::mutex> lock(m);  // what was released before the synchronisation began is idle now (later releases are not)
            for (auto &kv : free_ranges) if (kv.second.dirty <= sync_mark) kv.second.dirty = 0;
        }
        return out;
    }
    bool owns(const void *p) {
        std::lock_guard<std::mutex> lock(m);
        return live.count((char *)p) != 0;
    }
    void free(void *p, bool dirty) {
        std::lock_guard<std::mutex> lock(m);
        auto it = live.find((char *)p);
        if (it == live.end()) MTG_DIE("internal error: device range %p is not an allocation of the arena of device %d", p, device);
        const size_t bytes = it->second;
        live.erase(it);
        live_bytes -= bytes;
        chunk_of(p)->live -= bytes;
        insert_free((char *)p, bytes, dirty ? ++dirty_gen : 0);
    }
    // one chunk of `bytes` unless that much is free in one piece already
    void reserve(size_t bytes) {
        bytes = (bytes + BIG - 1) / BIG * BIG;
        std::lock_guard<std::mutex> lock(m);
        if (bytes > whole_call_limit()) return;
        if (!free_by_size.empty() && std::prev(free_by_size.end())->first >= bytes) return;
        (void)add_chunk(bytes);  // (failure is not an error here: the allocations themselves will ask again, in smaller pieces)
    }
    // at least `bytes` free in all (not necessarily in one piece), else one more chunk for what is missing: the head of a stage that
    // is about to take many arrays (instead of one small chunk per array). explicit_request: the caller asked for the memory by name
    // (mtg_device_create_opts(MTG_DEVICE_RESERVE_WORK)) -- the whole-call limit does not apply, `budget` (bytes the driver may still
    // be asked for) does.
    void ensure_free(size_t bytes, bool explicit_request = false, size_t budget = ~(size_t)0) {
        std::unique_lock<std::mutex> lock(m);
        if (pending.valid()) {
            std::shared_future<void> f = pending;
            pending = std::shared_future<void>();
            lock.unlock();
            f.wait();
            lock.lock();
        }
        size_t have = 0;
        for (auto &kv : free_ranges) have += kv.second.bytes;
        if (have >= bytes) return;
        if (!explicit_request && bytes > whole_call_limit()) return;
        const size_t more = bytes - have + (bytes - have) / 8;
        if (more > budget) return;
        (void)add_chunk(more);
    }
    size_t reclaimable_bytes() {  // what release_free_chunks(true) would give back right now
        std::lock_guard<std::mutex> lock(m);
        size_t n = 0;
        for (Chunk &c : chunks) {
            auto it = free_ranges.find(c.base);
            if (c.live == 0 && it != free_ranges.end() && it->second.bytes == c.bytes) n += c.bytes;
        }
        return n;
    }
};
inline DeviceArena &device_arena(int device_id) {
    static DeviceArena *arenas = [] {  // (never destroyed: buffers may be released during static destruction)
        DeviceArena *a = new DeviceArena[64];
        for (int i = 0; i < 64; i++) a[i].device = i;
        return a;
    }();
    if (device_id < 0 || device_id >= 64) MTG_DIE("device id %d out of range", device_id);
    return arenas[device_id];
}

// the library's hipMalloc / hipFree: ranges of the current device's arena (grow_hint: size of the chunk to add if none has room)
template <typename T>
inline void device_malloc(T **p, size_t bytes, size_t grow_hint = 0) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    *p = (T *)device_arena(dev).alloc(bytes, false, grow_hint);
}
// like hipFree: the device is idle when the range goes back (callers free right after a stage, when it is idle anyway)
inline void device_free(const void *p) {
    if (!p) return;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceSynchronize());
    device_arena(dev).free(const_cast<void *>(p), false);
}
// ... on a named device (destructors that may run on any thread)
inline void device_free_on(int device_id, const void *p) {
    if (!p) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device_id);
    (void)hipDeviceSynchronize();
    device_arena(device_id).free(const_cast<void *>(p), false);
    (void)hipSetDevice(cur);
}

// One stream per device for the finishing stages, created on first use.
inline hipStream_t finish_stream(int device_id) {
    static hipStream_t streams[64] = {nullptr};
    static std::mutex mu;
    if (device_id < 0 || device_id >= 64) MTG_DIE("device id %d out of range", device_id);
    std::lock_guard<std::mutex> l(mu);
    if (!streams[device_id]) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        HIP_CHECK(hipSetDevice(device_id));
        HIP_CHECK(hipStreamCreate(&streams[device_id]));
        (void)hipSetDevice(cur);
    }
    return streams[device_id];
}

// Work arrays of the finishing stages: every user runs on the device's one finish stream, so a range released earlier on that
// stream is free when it is used again -- no synchronisation between release and reuse.
struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { release(); }
    void release() {
        if (p) device_arena(device).free(p, true);
        p = nullptr;
    }
    template <typename T>
    T *alloc(hipStream_t st, uint64_t n) {
        release();
        HIP_CHECK(hipGetDevice(&device));
        if (st != finish_stream(device)) MTG_DIE("hu::Buf: work arrays of the finishing stages belong on the finish stream of device %d", device);
        bytes = (n ? n : 1) * sizeof(T);
        p = device_arena(device).alloc(bytes, true);
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
    (void)hipDeviceSynchronize();
    for (void *p : {c->d_from, c->d_mirror, c->d_row0, c->d_adj0})
        if (p) device_arena(c->device).free(p, false);
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
// Takes ownership of two arrays from hu::device_malloc on `device` (from-nodes of the original edges, mirror); keeps an existing
// cache of the same device and frees the offered arrays instead.
inline void edge_cache_put(const HostGraph &g, int device, uint32_t *d_from, uint32_t *d_mirror) {
    std::lock_guard<std::mutex> l(edge_cache_mutex());
    if (g.device_cache || (finish_tuning().flags.load() & FT_NO_EDGE_CACHE)) {  // first come, first kept (one cache per graph)
        device_free_on(device, d_from);
        device_free_on(device, d_mirror);
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
    device_free_on(device, d_row0);
    device_free_on(device, d_adj0);
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
// (max_threads: host threads that empty the ring -- 8 move a plain copy at PCIe speed; a `put` that writes several bytes per byte it
// reads, like the clib.rs flattening, takes more)
template <typename Put>
inline void download_sliced_with(const void *d_src, size_t bytes, hipStream_t st, int device_id, Put &&put,
                                 const std::function<void(size_t)> *progress = nullptr, unsigned max_threads = 8) {
    constexpr size_t SLICE = RING_SLICE;
    constexpr int NS = RING_SLOTS;
    TransferRing &r = transfer_ring(device_id);
    std::lock_guard<std::mutex> lock(r.m);
    r.ready();
    const size_t n_slices = (bytes + SLICE - 1) / SLICE;
    const unsigned T = std::max(1u, std::min(max_threads, std::thread::hardware_concurrency()));
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
// (a plain copy into pageable memory is staged by the runtime on one thread, ~9 GB/s: worth it only where starting the ring's threads
// costs more -- until round 5 the limit was 64 MB, 7 ms of a small graph's 40-ms call)
constexpr size_t PLAIN_COPY_LIMIT = 4u << 20;
inline void download_sliced(void *dst, const void *d_src, size_t bytes, hipStream_t st, int device_id) {
    if (bytes < PLAIN_COPY_LIMIT) {
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return;
    }
    download_sliced_with(d_src, bytes, st, device_id, [dst](size_t off, const char *src, size_t n) { std::memcpy((char *)dst + off, src, n); });
}
// n 32-bit words on the device -> n 64-bit words on the host: half the bytes over PCIe, widened by the threads that empty the ring
inline void download_sliced_widen(uint64_t *dst, const uint32_t *d_src, size_t n, hipStream_t st, int device_id, unsigned max_threads = 8) {
    if (n * 4 < PLAIN_COPY_LIMIT) {
        std::vector<uint32_t> tmp(n);
        if (n) HIP_CHECK(hipMemcpyAsync(tmp.data(), d_src, n * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (size_t i = 0; i < n; i++) dst[i] = tmp[i];
        return;
    }
    download_sliced_with(d_src, n * 4, st, device_id, [dst](size_t off, const char *src, size_t bytes) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
        uint64_t *out = dst + off / 4;
        for (size_t i = 0; i < bytes / 4; i++) __builtin_nontemporal_store((uint64_t)w[i], &out[i]);  // (written once, read by the caller later)
    }, nullptr, max_threads);
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

// End of a finishing call: the arena's chunks stay with the library for the next call (a caller that finishes graph after graph
// allocates nothing) while the call worked on less than a third of the device's memory -- BASELINE configs[3] at its nominal size
// (2^30: ~80 GB by this estimate) iterates without asking the driver for tens of gigabytes per call, which costs between 0.2 and 5 s
// there --; chunks the call hardly used (a caller that moved on to smaller graphs) and, beyond that size, every chunk that is free
// go back. mtg_release_device_memory returns what is free at any time.
inline void finish_trim(int device_id, uint64_t bytes_used) {
    static uint64_t threshold[64] = {0};
    if (device_id >= 0 && device_id < 64 && !threshold[device_id]) {
        size_t free_b = 0, total_b = 0;
        threshold[device_id] = hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? (uint64_t)total_b / 3 : (32ull << 30);
    }
    const uint64_t limit = device_id >= 0 && device_id < 64 ? threshold[device_id] : (32ull << 30);
    device_arena(device_id).release_free_chunks(bytes_used >= limit);
}

}  // namespace hu
}  // namespace mtg
