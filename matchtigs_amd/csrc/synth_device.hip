// synth_device.hip -- the G-csr benchmark / test input generated on the MI355X.
//
// Twin of matchtigs_amd/synth.py g_csr (numpy): the same counter-based splitmix64 streams, the same rule ("a unitig whose
// two directed edges would be the 5th out-edge of their from-nodes, counting ALL generated edges in edge order, is dropped"),
// hence the same graph -- but in seconds instead of minutes at the human-like and pangenome-like sizes (2^30 / 2^31 nominal
// edges), where the numpy argsort dominates everything else. No reference counterpart: the reference reads real genomes
// (bin.rs:902-912), which cannot be shipped. Test infrastructure that lives in the library because it needs the GPU.
//
// rank(e) < max_degree  <=>  e is among the max_degree smallest edge ids leaving from(e): max_degree rounds of "atomicMin of
// the ids above the previous round's minimum" per node give the max_degree-th smallest id per node; no sort.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>

#include "device.hpp"
#include "hip_util.hpp"
#include "parallel.hpp"

namespace mtg {

using namespace hu;

namespace {

struct SynthParams {
    uint64_t base_a, base_b, base_w;  // seed + (stream << 40) for streams 1, 2, 3
    uint64_t n_unitigs;
    uint64_t n_nodes;
    uint64_t n_paired;  // 2 * n_binodes: ids below are in mirror pairs (n ^ 1), ids from here on are self-mirror
};

__device__ __forceinline__ uint64_t splitmix(uint64_t base, uint64_t counter) {
    uint64_t z = base + counter * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t mirror_of(const SynthParams &p, uint32_t n) { return n < p.n_paired ? (n ^ 1u) : n; }
// from-nodes of the two directed edges of unitig u: 2u = a -> b, 2u + 1 = m(b) -> m(a)
__device__ __forceinline__ void endpoints(const SynthParams &p, uint64_t u, uint32_t &a, uint32_t &b) {
    a = (uint32_t)(splitmix(p.base_a, u + 1) % p.n_nodes);
    b = (uint32_t)(splitmix(p.base_b, u + 1) % p.n_nodes);
}

// keys are edge id + 1 (0 = "no previous minimum"), 0xFFFFFFFF = "fewer edges than rounds"
__global__ __launch_bounds__(EB) void round_kernel(SynthParams p, const uint32_t *prev, uint32_t *cur) {
    const uint64_t u = gid();
    if (u >= p.n_unitigs) return;
    uint32_t a, b;
    endpoints(p, u, a, b);
    const uint32_t f0 = a, f1 = mirror_of(p, b);
    const uint32_t k0 = (uint32_t)(2 * u + 1), k1 = (uint32_t)(2 * u + 2);
    if (k0 > prev[f0]) atomicMin(&cur[f0], k0);
    if (k1 > prev[f1]) atomicMin(&cur[f1], k1);
}
__global__ __launch_bounds__(EB) void keep_kernel(SynthParams p, const uint32_t *thr, uint32_t *keep) {
    const uint64_t u = gid();
    if (u >= p.n_unitigs) return;
    uint32_t a, b;
    endpoints(p, u, a, b);
    const uint32_t k0 = (uint32_t)(2 * u + 1), k1 = (uint32_t)(2 * u + 2);
    keep[u] = (k0 <= thr[a] && k1 <= thr[mirror_of(p, b)]) ? 1u : 0u;
}
// weight = 1 + #{j : x <= T_j}, T descending (the integer form of 1 + floor(log(x / 2^53) / log1p(-p)) clipped to k)
__global__ __launch_bounds__(EB) void emit_kernel(SynthParams p, const uint32_t *keep, const uint32_t *pos, const uint64_t *thresholds,
                                                 uint32_t n_thresholds, uint32_t *e_from, uint32_t *e_to, uint16_t *w16) {
    const uint64_t u = gid();
    if (u >= p.n_unitigs || !keep[u]) return;
    uint32_t a, b;
    endpoints(p, u, a, b);
    const uint64_t x = (splitmix(p.base_w, u + 1) >> 11) + 1;
    uint32_t lo = 0, hi = n_thresholds;  // first index with T < x
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (thresholds[mid] >= x) lo = mid + 1;
        else hi = mid;
    }
    const uint64_t j = pos[u];
    e_from[2 * j] = a;
    e_to[2 * j] = b;
    e_from[2 * j + 1] = mirror_of(p, b);
    e_to[2 * j + 1] = mirror_of(p, a);
    w16[j] = (uint16_t)(1 + lo);
}

}  // namespace

HostGraph *device_synth_g_csr(uint64_t n_binodes, uint64_t n_self_mirrors, uint64_t n_unitigs, uint64_t seed, uint64_t k,
                              const uint64_t *thresholds, uint64_t n_thresholds, int max_degree, int device_id) {
    const uint64_t V = 2 * n_binodes + n_self_mirrors;
    if (V == 0 || V >= NONE) MTG_DIE("mtg_synth_g_csr: %llu nodes do not fit 32-bit ids", (unsigned long long)V);
    if (2 * n_unitigs + 2 >= NONE) MTG_DIE("mtg_synth_g_csr: %llu directed edges do not fit 32-bit ids", (unsigned long long)(2 * n_unitigs));
    if (k < 1 || k > 65535 || n_thresholds + 1 > k) MTG_DIE("mtg_synth_g_csr: bad k / threshold count");
    if (max_degree < 1 || max_degree > 64) MTG_DIE("mtg_synth_g_csr: max_degree out of range");
    if (device_count() <= device_id) MTG_DIE("mtg_synth_g_csr: no MI355X/HIP device %d (the generator has no CPU form in the library; numpy twin: synth.g_csr)", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    device_reserve_async(V, 2 * n_unitigs, device_id);  // (the device memory of the call that will follow on this graph; the generator's own arrays are ranges of it)
    hipStream_t st = finish_stream(device_id);
    static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] synth_g_csr: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };

    SynthParams p;
    p.base_a = seed + (1ull << 40);
    p.base_b = seed + (2ull << 40);
    p.base_w = seed + (3ull << 40);
    p.n_unitigs = n_unitigs;
    p.n_nodes = V;
    p.n_paired = 2 * n_binodes;

    Buf b_thr0, b_thr1, b_keep, b_pos, b_bsum, b_tot, b_T;
    uint32_t *thr[2] = {b_thr0.alloc<uint32_t>(st, V), b_thr1.alloc<uint32_t>(st, V)};
    uint32_t *d_keep = b_keep.alloc<uint32_t>(st, n_unitigs);
    uint32_t *d_pos = b_pos.alloc<uint32_t>(st, n_unitigs);
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(n_unitigs) + 1);
    uint32_t *d_tot = b_tot.alloc<uint32_t>(st, 2);
    uint64_t *d_T = b_T.alloc<uint64_t>(st, n_thresholds);
    if (n_thresholds) HIP_CHECK(hipMemcpyAsync(d_T, thresholds, n_thresholds * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(thr[0], 0, V * 4, st));
    int cur = 0;
    if (n_unitigs)
        for (int r = 0; r < max_degree; r++) {
            HIP_CHECK(hipMemsetAsync(thr[cur ^ 1], 0xFF, V * 4, st));
            round_kernel<<<grid_for(n_unitigs), EB, 0, st>>>(p, thr[cur], thr[cur ^ 1]);
            cur ^= 1;
        }
    else HIP_CHECK(hipMemsetAsync(thr[cur], 0xFF, V * 4, st));
    uint32_t kept = 0;
    if (n_unitigs) {
        keep_kernel<<<grid_for(n_unitigs), EB, 0, st>>>(p, thr[cur], d_keep);
        scan_u32<uint32_t>(st, d_keep, n_unitigs, d_pos, d_bsum, d_tot);
        HIP_CHECK(hipMemcpyAsync(&kept, d_tot, 4, hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
    b_thr0.release();
    b_thr1.release();
    lap("degree cap + count");

    const uint64_t E = 2 * (uint64_t)kept;
    Buf b_from, b_to, b_w;
    uint32_t *d_from = b_from.alloc<uint32_t>(st, E);
    uint32_t *d_to = b_to.alloc<uint32_t>(st, E);
    uint16_t *d_w = b_w.alloc<uint16_t>(st, kept);
    if (n_unitigs) emit_kernel<<<grid_for(n_unitigs), EB, 0, st>>>(p, d_keep, d_pos, d_T, (uint32_t)n_thresholds, d_from, d_to, d_w);
    HIP_CHECK(hipGetLastError());

    // ---- host graph, filled in place (what graph_from_edges builds from caller arrays) ----
    HostGraph *g = new HostGraph();
    g->init_nodes(V);
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++) g->mirror[n] = n < 2 * n_binodes ? (uint32_t)(n ^ 1) : (uint32_t)n;
    });
    g->reserve_edges(E + E / 2 + 1024);
    g->append_unlinked(E);
    PodVec<uint16_t> w16(kept);
    if (E) {
        HIP_CHECK(hipMemcpyAsync(g->e_from.data(), d_from, E * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(g->e_to.data(), d_to, E * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(w16.data(), d_w, (uint64_t)kept * 2, hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    lap("emit + download");
    parallel_ranges(kept, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t u = lo; u < hi; u++) {
            g->w_biedge[u] = w16[u];  // (edge 2u = unitig u forwards, 2u + 1 its mirror: host_graph.hpp)
        }
    });
    g->n_original_edges = E;  // (host adjacency: linked on demand, like mtg_graph_from_edges)
    g->built = true;
    lap("host graph");
    b_from.release(); b_to.release(); b_w.release(); b_keep.release(); b_pos.release(); b_bsum.release(); b_tot.release(); b_T.release();
    HIP_CHECK(hipStreamSynchronize(st));
    return g;
}

}  // namespace mtg
