// device_pairs.hip -- the search + claim drivers above the stages: the one-shot form (engine-owned candidate buffers, the search's arrays
// given back before the replay takes its own) and SURVEY 8e inside the library (sources block-partitioned by work over several GPUs,
// candidate lists gathered on the first, replay there). Part of the device stage; shared types: device_internal.hpp.
#include "device_internal.hpp"

namespace mtg {

// instead of new memory beside them.
void device_set_single_use(Device *d) { d->single_use = true; }
void device_drop_search_arrays(Device *d) {
    HIP_CHECK(hipDeviceSynchronize());
    for (void **p : {(void **)&d->d_recs, (void **)&d->d_ext_col, (void **)&d->d_ext_w, (void **)&d->d_act_index, (void **)&d->d_act_node,
                     (void **)&d->d_ovf[0], (void **)&d->d_ovf[1], (void **)&d->d_fix, (void **)&d->d_fix_dense, (void **)&d->d_fix_val, (void **)&d->d_fix_dense_val}) {
        if (*p) hu::device_arena(d->dev).free(*p, false);
        *p = nullptr;
    }
    d->ovf_cap = 0;
    d->act_cap = 0;
}

uint64_t device_pairs(Device *d, void *stream, mtg_pair **pairs_out, int *rounds_out) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    d->last_wall_s[0] = d->last_wall_s[1] = d->last_wall_s[2] = 0;
    if (!S) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    unsigned long long *d_start = nullptr, *d_pool = nullptr;
    uint32_t *d_count = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    hu::device_malloc(&d_start, S * 8);
    hu::device_malloc(&d_count, S * 4);
    uint64_t cap = std::max<uint64_t>(S * 2 + std::min<uint64_t>((S + 63) / 64, (uint64_t)d->n_cu * 8) * ENUM_POOL_CHUNK, 1024);  // keys + per-wave chunk slack
    for (;;) {
        hu::device_malloc(&d_pool, cap * 8);
        uint64_t needed = 0;
        if (run_levels(d, st, 0, 0, S, d_pool, cap, d_start, d_count, &needed, nullptr) == 0) break;
        hu::device_free(d_pool);
        d_pool = nullptr;
        cap = needed + needed / 8 + 1024;
    }
    if (d->single_use) device_drop_search_arrays(d);
    d->last_wall_s[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    const uint64_t n = device_replay(d, stream, S, (const uint64_t *)d_start, d_count, (const uint64_t *)d_pool, pairs_out, rounds_out);
    hu::device_free(d_pool);
    hu::device_free(d_start);
    hu::device_free(d_count);
    return n;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU inside the library: the graph is replicated on every device, the ascending source list is cut into
// contiguous blocks of equal estimated work (1 + out-degree of the source), every device runs the SSSP stage over its block
// from its own host thread, the candidate lists are gathered on the first device with peer copies over xGMI (one copy of the
// counts, one of the starts, one of the keys per device: concatenation in device order IS the replay order), and the claim
// replay runs there. No other exchange exists on the path.
// ------------------------------------------------------------------------------------------------
__global__ void source_work_kernel(const uint32_t *out_nodes, const uint32_t *odeg, uint64_t n, uint32_t *work) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) work[i] = 1u + (odeg[out_nodes[i]] & ~ODEG_REACH);
}
// cut[r] = first source whose exclusive work prefix reaches total * r / parts (r = 1 .. parts-1)
__global__ void work_cuts_kernel(const unsigned long long *prefix, const uint32_t *work, uint64_t n, const unsigned long long *total, int parts,
                                 unsigned long long *cut) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long lo = prefix[i], hi = lo + work[i], tot = *total;
    for (int r = 1; r < parts; r++) {
        const unsigned long long target = tot / (unsigned)parts * (unsigned)r + tot % (unsigned)parts * (unsigned)r / (unsigned)parts;
        if (lo <= target && target < hi) cut[r] = i;
    }
}
__global__ void rebase_starts_kernel(unsigned long long *start, uint64_t n, unsigned long long offset) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) start[i] += offset;
}

std::vector<uint64_t> device_partition_sources(Device *d, void *stream, int parts) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    std::vector<uint64_t> cuts((size_t)parts + 1, 0);
    cuts[(size_t)parts] = S;
    if (parts <= 1 || S == 0) {
        for (int r = 1; r < parts; r++) cuts[(size_t)r] = S;
        return cuts;
    }
    uint32_t *d_work = nullptr;
    unsigned long long *d_prefix = nullptr, *d_cut = nullptr;
    hu::device_malloc(&d_work, S * 4);
    hu::device_malloc(&d_prefix, S * 8);
    hu::device_malloc(&d_cut, (size_t)parts * 8);
    HIP_CHECK(hipMemsetAsync(d_cut, 0, (size_t)parts * 8, st));
    hipLaunchKernelGGL(source_work_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d->d_out_nodes, d->d_odeg, S, d_work);
    scan_u32(d, st, d->replay, d_work, S, d_prefix, &d->d_counters[C_OVF_LIST]);
    hipLaunchKernelGGL(work_cuts_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d_prefix, d_work, S, &d->d_counters[C_OVF_LIST], parts, d_cut);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> h((size_t)parts);
    HIP_CHECK(hipMemcpyAsync(h.data(), d_cut, (size_t)parts * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (int r = 1; r < parts; r++) cuts[(size_t)r] = std::max<uint64_t>(cuts[(size_t)r - 1], h[(size_t)r]);
    hu::device_free(d_work); hu::device_free(d_prefix); hu::device_free(d_cut);
    return cuts;
}

// All devices must hold the same graph and be classified. Returns the pairs of the whole graph (host, malloc'd).
uint64_t device_pairs_multi(Device *const *devs, int n_dev, mtg_pair **pairs_out, int *rounds_out, double *gather_ms_out) {
    if (n_dev < 1) MTG_DIE("device_pairs_multi: no device");
    if (n_dev == 1) return device_pairs(devs[0], nullptr, pairs_out, rounds_out);
    Device *d0 = devs[0];
    const uint64_t S = d0->n_sources;
    for (int i = 1; i < n_dev; i++)
        if (devs[i]->V != d0->V || devs[i]->n_sources != S) MTG_DIE("device_pairs_multi: the devices hold different graphs");
    if (!S) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d0->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    const std::vector<uint64_t> cuts = device_partition_sources(d0, nullptr, n_dev);
    struct Part { unsigned long long *start = nullptr, *pool = nullptr; uint32_t *count = nullptr; uint64_t used = 0; };
    std::vector<Part> parts((size_t)n_dev);
    std::vector<std::thread> th;
    for (int i = 0; i < n_dev; i++) {
        th.emplace_back([&, i]() {  // one host thread per device: its block of sources, its own buffers
            Device *d = devs[i];
            Part &p = parts[(size_t)i];
            HIP_CHECK(hipSetDevice(d->dev));
            const uint64_t lo = cuts[(size_t)i], hi = cuts[(size_t)i + 1], n = hi - lo;
            if (!n) return;
            hu::device_malloc(&p.start, n * 8);
            hu::device_malloc(&p.count, n * 4);
            uint64_t cap = std::max<uint64_t>(n * 2 + std::min<uint64_t>((n + 63) / 64, (uint64_t)d->n_cu * 8) * ENUM_POOL_CHUNK, 1024);
            for (;;) {
                hu::device_malloc(&p.pool, cap * 8);
                uint64_t needed = 0;
                if (run_levels(d, nullptr, 0, lo, hi, p.pool, cap, p.start, p.count, &needed, nullptr) == 0) { p.used = needed; break; }
                hu::device_free(p.pool);
                p.pool = nullptr;
                cap = needed + needed / 8 + 1024;
            }
        });
    }
    for (auto &t : th) t.join();
    // gather on the first device
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipSetDevice(d0->dev));
    uint64_t pool_total = 0;
    for (const Part &p : parts) pool_total += p.used;
    unsigned long long *g_start = nullptr, *g_pool = nullptr;
    uint32_t *g_count = nullptr;
    hu::device_malloc(&g_start, S * 8);
    hu::device_malloc(&g_count, S * 4);
    hu::device_malloc(&g_pool, std::max<uint64_t>(pool_total, 1) * 8);
    uint64_t off = 0;
    for (int i = 0; i < n_dev; i++) {
        const Part &p = parts[(size_t)i];
        const uint64_t lo = cuts[(size_t)i], n = cuts[(size_t)i + 1] - lo;
        if (!n) continue;
        auto copy = [&](void *dst, const void *src, size_t bytes) {
            if (!bytes) return;
            if (devs[i]->dev == d0->dev) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, d0->dev, src, devs[i]->dev, bytes, nullptr));
        };
        copy(g_start + lo, p.start, n * 8);
        copy(g_count + lo, p.count, n * 4);
        copy(g_pool + off, p.pool, p.used * 8);
        if (off) hipLaunchKernelGGL(rebase_starts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, g_start + lo, n, (unsigned long long)off);
        off += p.used;
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(nullptr));
    if (gather_ms_out) *gather_ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n_dev; i++) {
        HIP_CHECK(hipSetDevice(devs[i]->dev));
        if (parts[(size_t)i].start) { hu::device_free(parts[(size_t)i].start); hu::device_free(parts[(size_t)i].count); hu::device_free(parts[(size_t)i].pool); }
    }
    HIP_CHECK(hipSetDevice(d0->dev));
    if (d0->single_use) device_drop_search_arrays(d0);
    d0->last_wall_s[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    const uint64_t n = device_replay(d0, nullptr, S, (const uint64_t *)g_start, g_count, (const uint64_t *)g_pool, pairs_out, rounds_out);
    hu::device_free(g_pool); hu::device_free(g_start); hu::device_free(g_count);
    return n;
}

// convenience for tests: allocates device buffers, grows the pool on demand, downloads to host vectors
void device_candidates_to_host(Device *d, void *stream, std::vector<uint64_t> &cand_start, std::vector<uint32_t> &cand_count,
                               std::vector<uint64_t> &pool) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    cand_start.assign(S, 0);
    cand_count.assign(S, 0);
    pool.clear();
    if (!S) return;
    unsigned long long *d_start = nullptr, *d_pool = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, S * 8);
    hu::device_malloc(&d_count, S * 4);
    uint64_t cap = std::max<uint64_t>(S * 4, 1024);
    for (;;) {
        hu::device_malloc(&d_pool, cap * 8);
        uint64_t needed = 0;
        const int rc = run_levels(d, st, 0, 0, S, d_pool, cap, d_start, d_count, &needed, nullptr);
        if (rc == 0) {
            pool.resize(needed);
            if (needed) HIP_CHECK(hipMemcpyAsync(pool.data(), d_pool, needed * 8, hipMemcpyDeviceToHost, st));
            break;
        }
        hu::device_free(d_pool);
        d_pool = nullptr;
        cap = needed + needed / 8 + 1024;
    }
    HIP_CHECK(hipMemcpyAsync(cand_start.data(), d_start, S * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(cand_count.data(), d_count, S * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hu::device_free(d_pool);
    hu::device_free(d_start);
    hu::device_free(d_count);
}


void device_warm_pairs_unit(hipFuncAttributes *a) { (void)hipFuncGetAttributes(a, reinterpret_cast<const void *>(source_work_kernel)); }

}  // namespace mtg
