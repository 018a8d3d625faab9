// device_classify.hip -- node classification on the GPU (greedytigs/mod.rs:222-255): out-degree and mirror -> multiplicity, class byte, the
// ascending out-node list and -- in the same compaction pass -- the list of sources that can reach an in-node at all (the only ones the
// search visits). Part of the device stage; shared types: device_internal.hpp.
#include "device_internal.hpp"

namespace mtg {

// ------------------------------------------------------------------------------------------------
// Classification kernels (greedytigs/mod.rs:229-245): compact arrays only (out-degree, mirror -> multiplicity, class byte)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(CLS_BLOCK) void classify_kernel(const uint32_t *odeg, const uint32_t *mirror, uint32_t n_nodes, int32_t *mult,
                                                             uint8_t *cls, uint32_t *block_counts, uint32_t *block_demand,
                                                             uint32_t *block_active) {
    __shared__ uint32_t wave_cnt[CLS_BLOCK / 64];
    __shared__ uint32_t wave_dem[CLS_BLOCK / 64];
    __shared__ uint32_t wave_act[CLS_BLOCK / 64];
    uint32_t cnt = 0, pos = 0, act = 0;
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {
        const uint64_t n64 = (uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x;
        if (n64 >= n_nodes) continue;
        const uint32_t n = (uint32_t)n64;
        const uint32_t m = mirror[n];
        const uint32_t on = odeg[n];
        const NodeClass c = classify_node(on & ~ODEG_REACH, m == n ? 0u : odeg[m] & ~ODEG_REACH, m == n);
        cnt += (c.cls & F_SOURCE) ? 1u : 0u;
        // (8:8 format) a source that can reach an in-node within the bound at all: the only ones the SSSP stage searches. The flag
        // is a function of the graph and rides in bit 31 of the out-degree word this kernel reads anyway.
        const bool reaches = (c.cls & F_SOURCE) && (on & ODEG_REACH);
        act += reaches ? 1u : 0u;
        mult[n] = c.diff;  // 0 for balanced nodes
        cls[n] = c.cls | (reaches ? F_REACH : 0);
        pos += c.diff > 0 ? (uint32_t)c.diff : 0u;
    }
    for (int dd = 32; dd >= 1; dd >>= 1) { pos += __shfl_down(pos, dd); cnt += __shfl_down(cnt, dd); act += __shfl_down(act, dd); }
    if ((threadIdx.x & 63) == 0) { wave_cnt[threadIdx.x >> 6] = cnt; wave_dem[threadIdx.x >> 6] = pos; wave_act[threadIdx.x >> 6] = act; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0, dm = 0, ac = 0;
        for (int i = 0; i < CLS_BLOCK / 64; i++) { s += wave_cnt[i]; dm += wave_dem[i]; ac += wave_act[i]; }
        block_counts[blockIdx.x] = s;
        block_demand[blockIdx.x] = dm;
        block_active[blockIdx.x] = ac;
    }
}

// single-workgroup exclusive scan of block_counts -> block_offsets (in place), total in *total_out; sum of block_demand in *demand_out.
// Launched with two workgroups, the second one scans a second array the same way (counts2 -> *total2_out).
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint32_t *counts, uint32_t n, unsigned long long *total_out,
                                                           const uint32_t *block_demand, unsigned long long *demand_out,
                                                           uint32_t *counts2 = nullptr, unsigned long long *total2_out = nullptr) {
    if (blockIdx.x == 1) { counts = counts2; total_out = total2_out; block_demand = nullptr; demand_out = nullptr; }
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    __shared__ unsigned long long dem_sum;
    if (threadIdx.x == 0) { carry = 0; dem_sum = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long dem = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? counts[i] : 0;
        dem += (block_demand && i < n) ? block_demand[i] : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int j = 0; j < wv; j++) wave_off += wave_tot[j];
        const uint32_t c = carry;
        if (i < n) counts[i] = c + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wave_off + incl;
        __syncthreads();
    }
    for (int dd = 32; dd >= 1; dd >>= 1) dem += __shfl_down(dem, dd);
    if (lane == 0 && dem) atomicAdd(&dem_sum, dem);
    __syncthreads();
    if (threadIdx.x == 0) { *total_out = carry; if (demand_out) *demand_out = dem_sum; }
}

// out_nodes = the sources, ascending; and the sources that can reach an in-node within the bound at all (F_REACH in their class
// byte; 8:8 format), in order: act_index = their positions in out_nodes, act_node = their nodes -- the only sources the SSSP stage
// searches (the others have an empty candidate list by construction). No gather and no pass of its own: the flag arrives with the
// class byte this pass reads anyway.
__global__ __launch_bounds__(CLS_BLOCK) void compact_sources_kernel(const uint8_t *cls, uint32_t n_nodes,
                                                                    const uint32_t *block_offsets, uint32_t *out_nodes,
                                                                    const uint32_t *block_act_offsets, uint32_t *act_index, uint32_t *act_node) {
    __shared__ uint32_t wave_cnt[CLS_PER][CLS_BLOCK / 64], wave_act[CLS_PER][CLS_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long bal[CLS_PER], abal[CLS_PER];
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {
        const uint64_t n = (uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x;
        const uint8_t c = n < n_nodes ? cls[n] : (uint8_t)0;
        bal[p] = __ballot((c & F_SOURCE) != 0);
        abal[p] = __ballot((c & F_REACH) != 0);
        if (lane == 0) { wave_cnt[p][wv] = (uint32_t)__popcll(bal[p]); wave_act[p][wv] = (uint32_t)__popcll(abal[p]); }
    }
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x], aoff = block_act_offsets[blockIdx.x];
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {  // ascending: lane order within a wave, wave order within a pass, pass order, block order
        for (int j = 0; j < CLS_BLOCK / 64; j++) {
            if (j == wv && ((bal[p] >> lane) & 1ull)) {
                const uint32_t node = (uint32_t)((uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x);
                const uint32_t idx = off + (uint32_t)__popcll(bal[p] & ((1ull << lane) - 1ull));
                out_nodes[idx] = node;
                if ((abal[p] >> lane) & 1ull) {
                    const uint32_t pos = aoff + (uint32_t)__popcll(abal[p] & ((1ull << lane) - 1ull));
                    act_index[pos] = idx;
                    act_node[pos] = node;
                }
            }
            off += wave_cnt[p][j];
            aoff += wave_act[p][j];
        }
    }
}

__global__ void export_live_kernel(const uint8_t *cls, uint32_t n_nodes, uint8_t *live) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) live[n] = (cls[n] & F_TARGET) ? 1 : 0;
}

// The searched sources of a launch over the sources [src_begin, src_end): the part of the classification's list of sources that can
// reach an in-node (act_index, ascending) inside that range -- first entry and number, by a wave-wide 64-ary search (one wave; each
// round probes 64 evenly spaced entries: four rounds for ten million). *act_begin = first entry, *act_count = their number.
uint64_t device_classify(Device *d, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    d->n_sources = 0;
    if (!d->d_cls) {
        hu::device_malloc(&d->d_cls, std::max<uint64_t>(d->V, 1));
        hu::device_malloc(&d->d_mult, std::max<uint64_t>(d->V, 1) * 4);
        hu::device_malloc(&d->d_out_nodes, std::max<uint64_t>(d->V, 1) * 4);
    }
    if (d->V) {
        static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(classify_kernel, dim3((unsigned)d->n_cls_blocks), dim3(CLS_BLOCK), 0, st, d->d_odeg, d->d_mirror, (uint32_t)d->V,
                           d->d_mult, d->d_cls, d->d_block_counts, d->d_block_counts + d->n_cls_blocks, d->d_act_blocks);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(2), dim3(1024), 0, st, d->d_block_counts, (uint32_t)d->n_cls_blocks,
                           &d->d_counters[C_OVF_LIST], d->d_block_counts + d->n_cls_blocks, &d->d_counters[C_DEMAND],
                           d->d_act_blocks, d->d_act_total);
        HIP_CHECK(hipGetLastError());
        const auto t1 = std::chrono::steady_clock::now();
        read_counters(d, st);  // (the number of sources sizes the lists the compaction writes)
        if (dbg) std::fprintf(stderr, "[mtg] classify: two launches issued in %.3f ms, their results back after %.3f ms more\n",
                              std::chrono::duration<double, std::milli>(t1 - t0).count(),
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        d->n_sources = d->h_counters[C_OVF_LIST];
        d->total_demand = d->h_counters[C_DEMAND];
        const uint64_t act_need = d->w8 ? d->n_sources : 0;  // (without the 8:8 format no source carries the flag: the lists stay empty)
        if (d->act_cap < act_need || !d->d_act_index) {
            for (void *p : {(void *)d->d_act_index, (void *)d->d_act_node}) if (p) hu::device_free(p);
            hu::device_malloc(&d->d_act_index, std::max<uint64_t>(act_need, 1) * 4);
            hu::device_malloc(&d->d_act_node, std::max<uint64_t>(act_need, 1) * 4);
            d->act_cap = act_need;
        }
        hipLaunchKernelGGL(compact_sources_kernel, dim3((unsigned)d->n_cls_blocks), dim3(CLS_BLOCK), 0, st, d->d_cls,
                           (uint32_t)d->V, d->d_block_counts, d->d_out_nodes, d->d_act_blocks, d->d_act_index, d->d_act_node);
        HIP_CHECK(hipGetLastError());
    }
    d->classified = true;
    return d->n_sources;
}

void device_classify_download(Device *d, void *stream, uint32_t *out_nodes, int32_t *mult, uint8_t *live) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    if (!d->classified) MTG_DIE("mtg_classify_download: call mtg_classify first");
    if (out_nodes && d->n_sources)
        HIP_CHECK(hipMemcpyAsync(out_nodes, d->d_out_nodes, d->n_sources * 4, hipMemcpyDeviceToHost, st));
    if (mult && d->V) HIP_CHECK(hipMemcpyAsync(mult, d->d_mult, d->V * 4, hipMemcpyDeviceToHost, st));
    if (live && d->V) {
        uint8_t *d_live = nullptr;
        hu::device_malloc(&d_live, d->V);
        hipLaunchKernelGGL(export_live_kernel, dim3((unsigned)((d->V + 255) / 256)), dim3(256), 0, st, d->d_cls, (uint32_t)d->V, d_live);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(live, d_live, d->V, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        hu::device_free(d_live);
    }
    HIP_CHECK(hipStreamSynchronize(st));
}

const uint32_t *device_d_out_nodes(const Device *d) { return d->d_out_nodes; }
uint64_t device_n_sources(const Device *d) { return d->n_sources; }

void device_warm_classify_unit(hipFuncAttributes *a) { (void)hipFuncGetAttributes(a, reinterpret_cast<const void *>(classify_kernel)); }

}  // namespace mtg
