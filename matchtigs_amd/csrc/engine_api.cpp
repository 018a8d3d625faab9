// engine_api.cpp -- extern "C" surface of libmatchtigs: include/mtg_engine.h and include/matchtigs.h.
//
// matchtigs_* are the drop-in replacements of /root/reference/src/clib.rs:87-410; mtg_* is the
// engine layer underneath (see the header for which reference lines each stage replaces).
#include <spawn.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/matchtigs.h"
#include "../../include/mtg_engine.h"
#include "../../include/mtg_policy.h"
#include "device.hpp"
#include "euler_lean.hpp"
#include "host_graph.hpp"
#include "hugebuf.hpp"
#include "parallel.hpp"

using namespace mtg;

extern char **environ;

struct mtg_graph { HostGraph g; };
struct mtg_device { Device *d; };
// The tigs of a finish on the GPU stay in HBM (`dev`) until somebody asks for them on the host: host() downloads them once. Counts
// come from `dev` without a copy; a clib.rs caller's tigs never get here (TigSink).
struct mtg_walks {
    mutable Walks w;
    mutable std::unique_ptr<ResidentTigs> dev;
    mutable std::mutex m;
    mtg_walks() = default;
    explicit mtg_walks(Walks &&walks, ResidentTigs *resident = nullptr) : w(std::move(walks)), dev(resident) {}
    const Walks &host() const {
        std::lock_guard<std::mutex> lock(m);
        if (dev) {
            dev->download(w);
            dev.reset();
        }
        return w;
    }
    Walks &host() { return const_cast<Walks &>(static_cast<const mtg_walks *>(this)->host()); }
    // (the device arrays, if the tigs are still on that GPU; valid until a call brings them to the host -- one thread per handle)
    const ResidentTigs *resident_on(int device_id) const {
        std::lock_guard<std::mutex> lock(m);
        return dev && dev->device == device_id ? dev.get() : nullptr;
    }
    uint64_t count() const {
        std::lock_guard<std::mutex> lock(m);
        return dev ? dev->n_tigs : w.limits.size();
    }
    uint64_t total_edges() const {
        std::lock_guard<std::mutex> lock(m);
        return dev ? dev->n_edges : w.edges.size();
    }
};
// The clib.rs handle is the same object as the engine graph.
struct MatchtigsData { mtg_graph graph; };

static thread_local double g_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static bool g_log_initialised = false;
static thread_local double g_last_euler_kernel_ms = 0;
static thread_local double g_last_gather_ms = 0;
static thread_local mtg_dijkstra_performance_data g_last_perf = {};
static thread_local double g_last_finish_times[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// mtg_compute_tigs_clib: where a device finish on this thread delivers its tigs (the caller's clib.rs arrays) instead of a Walks object
static thread_local TigSink *g_clib_sink = nullptr;
static thread_local bool g_clib_sink_used = false;

static void check_config(const mtg_config *cfg, const char *who) {
    if (!cfg) MTG_DIE("%s: null configuration", who);
    // fields are only ever appended: an older caller's (shorter) struct is valid, a newer or a garbage one is not
    static_assert(sizeof(mtg_config) == MTG_CONFIG_MIN_SIZE, "a field was appended to mtg_config: read it only when cfg->struct_size covers it, and default it otherwise");
    if (cfg->struct_size < MTG_CONFIG_MIN_SIZE || cfg->struct_size > sizeof(mtg_config))
        MTG_DIE("%s: mtg_config.struct_size = %llu, this library understands %d to %zu bytes: fill the configuration with mtg_config_init, and "
                "build against an mtg_engine.h no newer than this library", who, (unsigned long long)cfg->struct_size, MTG_CONFIG_MIN_SIZE, sizeof(mtg_config));
    if (cfg->k < 1) MTG_DIE("%s: k must be >= 1", who);
    if (cfg->n_devices < 1 || cfg->n_devices > MTG_MAX_DEVICES) MTG_DIE("%s: n_devices = %d is out of range [1, %d]", who, cfg->n_devices, MTG_MAX_DEVICES);
    if (cfg->euler_mode != MTG_EULER_HOST_REFERENCE_ORDER && cfg->euler_mode != MTG_EULER_DEVICE) MTG_DIE("%s: unknown euler_mode %d", who, cfg->euler_mode);
    if (cfg->finish_stage != MTG_FINISH_AUTO && cfg->finish_stage != MTG_FINISH_HOST && cfg->finish_stage != MTG_FINISH_DEVICE)
        MTG_DIE("%s: unknown finish_stage %d", who, cfg->finish_stage);
    if (cfg->node_weight_array_type != MTG_NODE_WEIGHT_EPOCH_ARRAY && cfg->node_weight_array_type != MTG_NODE_WEIGHT_HASHBROWN_HASH_MAP)
        MTG_DIE("Unknown node weight array type: %d", cfg->node_weight_array_type);       // implementation/mod.rs:78
    if (cfg->heap_type != MTG_HEAP_STD_BINARY_HEAP) MTG_DIE("Unknown heap type: %d", cfg->heap_type);  // implementation/mod.rs:99
    if (cfg->performance_data_type != MTG_PERFORMANCE_DATA_NONE && cfg->performance_data_type != MTG_PERFORMANCE_DATA_COMPLETE)
        MTG_DIE("Unknown performance data type: %d", cfg->performance_data_type);          // implementation/mod.rs:123
}

static Walks euler_cycles_by_mode(const HostGraph &g, const mtg_config &cfg) {
    if (cfg.euler_mode == MTG_EULER_DEVICE) return device_euler_cycles(g, cfg.device_ids[0], &g_last_euler_kernel_ms);
    return euler_cycles(g);
}

static size_t tig_count(const mtg_walks *tigs) {  // (a device finish with a sink delivers no walks)
    return g_clib_sink && g_clib_sink_used ? (size_t)g_clib_sink->n_tigs : (size_t)tigs->count();
}
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void log_info(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static void log_info(const char *fmt, ...) {
    if (!g_log_initialised) return;  // the reference logs nothing before matchtigs_initialise either
    va_list ap;
    va_start(ap, fmt);
    std::fprintf(stderr, "[INFO] ");
    std::vfprintf(stderr, fmt, ap);
    std::fprintf(stderr, "\n");
    va_end(ap);
}

// Giving a large graph back takes as long as unmapping its memory does (0.05 s for the edge arrays of the 2^27 graph, 1 s with the
// 23 GB of walk records a reference-order finish leaves in its arena): the device copy goes at once, the host memory on a thread
// of its own -- the caller (clib.rs:291 consumes the handle inside matchtigs_compute_tigs) has its result and does not wait for munmap.
template <typename T>
static void free_graph_object(T *obj, HostGraph &g) {
    if (!obj) return;
    if (g.edge_count() < (1u << 22)) {
        delete obj;
        return;
    }
    device_release_graph_cache(g);
    g.arena.unpin_all();  // (what is left for the thread makes no HIP call)
    std::thread([obj]() { delete obj; }).detach();
}

extern "C" {

const char *mtg_version(void) { return "matchtigs-amd 0.1 (gfx950; reference algbio/matchtigs 2.1.9)"; }
int mtg_device_count(void) { return device_count(); }
unsigned mtg_policies(void) { return MTG_POLICY_MASK; }

// ---- host graph ----
mtg_graph *mtg_graph_from_edges(uint64_t n_nodes, const uint32_t *mirror, uint64_t n_edges, const uint32_t *edge_from,
                                const uint32_t *edge_to, const uint64_t *edge_weight) {
    HostGraph *h = graph_from_edges(n_nodes, mirror, n_edges, edge_from, edge_to, edge_weight);
    mtg_graph *g = new mtg_graph{std::move(*h)};
    delete h;
    return g;
}
mtg_graph *mtg_graph_builder_new(uint64_t unitig_amount) {
    HostGraph *h = builder_new(unitig_amount);
    mtg_graph *g = new mtg_graph{std::move(*h)};
    delete h;
    return g;
}
void mtg_graph_builder_merge(mtg_graph *g, uint64_t ua, int sa, uint64_t ub, int sb) {
    if (!g) MTG_DIE("mtg_graph_builder_merge: null graph");
    builder_merge(&g->g, ua, sa != 0, ub, sb != 0);
}
void mtg_graph_builder_merge_links(mtg_graph *g, uint64_t n_links, const int64_t *links) {
    if (!g || (n_links && !links)) MTG_DIE("mtg_graph_builder_merge_links: null argument");
    for (uint64_t i = 0; i < n_links; i++) {
        if (links[4 * i] < 0 || links[4 * i + 2] < 0) MTG_DIE("mtg_graph_builder_merge_links: negative unitig id in row %llu", (unsigned long long)i);
        builder_merge(&g->g, (uint64_t)links[4 * i], links[4 * i + 1] != 0, (uint64_t)links[4 * i + 2], links[4 * i + 3] != 0);
    }
}
void mtg_graph_builder_build(mtg_graph *g, const uint64_t *unitig_weights) {
    if (!g) MTG_DIE("mtg_graph_builder_build: null graph");
    builder_build(&g->g, unitig_weights);
}
void mtg_graph_free(mtg_graph *g) {
    if (g) free_graph_object(g, g->g);
}
void mtg_graph_reset(mtg_graph *g) {
    if (!g || !g->g.built) MTG_DIE("mtg_graph_reset: graph is not built");
    g->g.reset_to_original();
}
uint64_t mtg_graph_node_count(const mtg_graph *g) { return g->g.node_count(); }
uint64_t mtg_graph_edge_count(const mtg_graph *g) { return g->g.edge_count(); }
// (the payload of an edge is kept per biedge, and only what is not arithmetic: host_graph.hpp)
static void export_payload(const HostGraph &h, uint64_t first, uint64_t n, uint64_t *edge_weight, uint64_t *edge_dummy_id, uint64_t *edge_unitig,
                           uint8_t *edge_forwards) {
    if (!edge_weight && !edge_dummy_id && !edge_unitig && !edge_forwards) return;
    parallel_ranges(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t e = first + i;
            if (edge_weight) edge_weight[i] = h.weight(e);
            if (edge_dummy_id) edge_dummy_id[i] = h.dummy_id(e);
            if (edge_unitig) edge_unitig[i] = h.unitig(e);
            if (edge_forwards) edge_forwards[i] = h.forwards(e) ? 1 : 0;
        }
    });
}
void mtg_graph_export(const mtg_graph *g, uint32_t *mirror, uint32_t *edge_from, uint32_t *edge_to, uint64_t *edge_weight,
                      uint64_t *edge_dummy_id, uint64_t *edge_unitig, uint8_t *edge_forwards) {
    const HostGraph &h = g->g;
    const size_t V = h.node_count(), E = h.edge_count();
    if (mirror && V) std::memcpy(mirror, h.mirror.data(), V * 4);
    if (edge_from && E) std::memcpy(edge_from, h.e_from.data(), E * 4);
    if (edge_to && E) std::memcpy(edge_to, h.e_to.data(), E * 4);
    export_payload(h, 0, E, edge_weight, edge_dummy_id, edge_unitig, edge_forwards);
}

void mtg_graph_export_range(const mtg_graph *g, uint64_t first_edge, uint64_t n_edges, uint32_t *edge_from, uint32_t *edge_to,
                            uint64_t *edge_weight, uint64_t *edge_dummy_id, uint64_t *edge_unitig, uint8_t *edge_forwards) {
    const HostGraph &h = g->g;
    if (first_edge > h.edge_count() || n_edges > h.edge_count() - first_edge) MTG_DIE("mtg_graph_export_range: range exceeds the %llu edges", (unsigned long long)h.edge_count());
    if (!n_edges) return;
    const size_t o = first_edge, n = n_edges;
    if (edge_from) std::memcpy(edge_from, h.e_from.data() + o, n * 4);
    if (edge_to) std::memcpy(edge_to, h.e_to.data() + o, n * 4);
    export_payload(h, o, n, edge_weight, edge_dummy_id, edge_unitig, edge_forwards);
}
uint64_t mtg_graph_original_edge_count(const mtg_graph *g) { return g->g.n_original_edges; }

// ---- device stage ----
mtg_device *mtg_device_create(const mtg_graph *g, uint64_t k, int device_id) {
    if (!g || !g->g.built) MTG_DIE("mtg_device_create: graph is not built");
    // (a unitig of weight 0 aborts inside device_create, in the pass that clamps the weights: the bounded search and its lower
    // bounds need weights >= 1 -- the reference computes weight = len + 1 - k >= 1, bin.rs:369-376)
    return new mtg_device{device_create(g->g, k, device_id)};
}
mtg_device *mtg_device_create_opts(const mtg_graph *g, uint64_t k, int device_id, int flags) {
    if (!g || !g->g.built) MTG_DIE("mtg_device_create_opts: graph is not built");
    if (flags & ~(MTG_DEVICE_NO_LOWER_BOUNDS | MTG_DEVICE_RESERVE_WORK)) MTG_DIE("mtg_device_create_opts: unknown flags %d", flags);
    mtg_device *d = new mtg_device{device_create(g->g, k, device_id, !(flags & MTG_DEVICE_NO_LOWER_BOUNDS))};
    if (flags & MTG_DEVICE_RESERVE_WORK) device_reserve_step_work(d->d);
    const int used[1] = {device_id};
    device_drop_foreign_reservation(used, 1);  // (a constructor's provisional chunk on another GPU goes back)
    return d;
}
void mtg_device_build_lower_bounds(mtg_device *d, void *stream) {
    if (!d) MTG_DIE("mtg_device_build_lower_bounds: null device");
    device_build_lower_bounds(d->d, stream);
}
double mtg_device_lower_bounds_ms(const mtg_device *d) { return d ? device_lower_bounds_ms(d->d) : 0.0; }
void mtg_device_free(mtg_device *d) {
    if (!d) return;
    device_free(d->d);
    delete d;
}
uint64_t mtg_device_graph_bytes(const mtg_device *d) { return device_graph_bytes(d->d); }
uint64_t mtg_classify(mtg_device *d, void *stream) { return device_classify(d->d, stream); }
void mtg_classify_download(mtg_device *d, void *stream, uint32_t *out_nodes, int32_t *multiplicity, uint8_t *is_in_node) {
    device_classify_download(d->d, stream, out_nodes, multiplicity, is_in_node);
}
const uint32_t *mtg_classify_d_out_nodes(const mtg_device *d) { return device_d_out_nodes(d->d); }
int mtg_sssp_candidates(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end, uint64_t *d_pool,
                        uint64_t pool_capacity, uint64_t *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed) {
    return device_sssp(d->d, stream, src_begin, src_end, d_pool, pool_capacity, d_cand_start, d_cand_count, pool_needed);
}
double mtg_last_sssp_kernel_ms(const mtg_device *d) { return device_last_kernel_ms(d->d); }
int mtg_last_sssp_levels(const mtg_device *d, double *ms_out, uint64_t *sources_out, int capacity) {
    return device_last_levels(d->d, ms_out, sources_out, capacity);
}
const char *mtg_last_sssp_level_name(const mtg_device *d, int level) { return device_last_level_name(d->d, level); }
void mtg_sssp_count(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    device_sssp_count(d->d, stream, src_begin, src_end, stats);
}
void mtg_sssp_count_visited(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    device_sssp_count_visited(d->d, stream, src_begin, src_end, stats);
}
int mtg_sssp_prunes(const mtg_device *d) { return device_prunes(d->d) ? 1 : 0; }
uint64_t mtg_last_sssp_searched_sources(const mtg_device *d) { return device_last_active_sources(d->d); }
int mtg_set_sssp_plan(mtg_device *d, int plan) { return device_set_plan(d->d, plan); }
uint64_t mtg_replay_claims_device(mtg_device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start,
                                  const uint32_t *d_cand_count, const uint64_t *d_pool, mtg_pair **pairs_out) {
    if (!d || !pairs_out) MTG_DIE("mtg_replay_claims_device: null argument");
    return device_replay(d->d, stream, n_sources, d_cand_start, d_cand_count, d_pool, pairs_out, nullptr);
}
void mtg_last_replay_ms(const mtg_device *d, double out[2]) { device_last_replay_ms(d->d, out); }
void mtg_set_replay_tuning(mtg_device *d, uint64_t windows, int block, int grid, int role_mod, int plain_barrier) {
    if (!d) MTG_DIE("mtg_set_replay_tuning: null device");
    device_set_replay_tuning(d->d, windows, block, grid, role_mod, plain_barrier);
}
uint64_t mtg_replay_claims_resident(mtg_device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start,
                                    const uint32_t *d_cand_count, const uint64_t *d_pool) {
    if (!d) MTG_DIE("mtg_replay_claims_resident: null argument");
    return device_replay(d->d, stream, n_sources, d_cand_start, d_cand_count, d_pool, nullptr, nullptr);
}
const mtg_pair *mtg_resident_pairs(const mtg_device *d, uint64_t *n_pairs_out) {
    if (!d) MTG_DIE("mtg_resident_pairs: null argument");
    return device_resident_pairs(d->d, n_pairs_out);
}
uint64_t mtg_download_resident_pairs(mtg_device *d, mtg_pair **pairs_out) {
    if (!d || !pairs_out) MTG_DIE("mtg_download_resident_pairs: null argument");
    return device_download_pairs(d->d, pairs_out);
}
int mtg_last_replay_rounds(const mtg_device *d) { return device_last_replay_rounds(d->d); }
uint64_t mtg_last_replay_visits(const mtg_device *d) { return device_last_replay_visits(d->d); }
uint64_t mtg_compute_pairs(mtg_device *const *devices, int n_devices, mtg_pair **pairs_out) {
    if (!devices || n_devices < 1 || n_devices > MTG_MAX_DEVICES || !pairs_out) MTG_DIE("mtg_compute_pairs: bad argument");
    std::vector<Device *> dv;
    for (int i = 0; i < n_devices; i++) {
        if (!devices[i]) MTG_DIE("mtg_compute_pairs: null device %d", i);
        dv.push_back(devices[i]->d);
    }
    return device_pairs_multi(dv.data(), n_devices, pairs_out, nullptr, &g_last_gather_ms);
}
double mtg_last_gather_ms(void) { return g_last_gather_ms; }
void mtg_partition_sources(mtg_device *d, int parts, uint64_t *cuts_out) {
    if (!d || parts < 1 || !cuts_out) MTG_DIE("mtg_partition_sources: bad argument");
    const std::vector<uint64_t> c = device_partition_sources(d->d, nullptr, parts);
    for (int i = 0; i <= parts; i++) cuts_out[i] = c[(size_t)i];
}

// ---- host stages ----
uint64_t mtg_replay_claims(const mtg_graph *g, uint64_t n_sources, const uint32_t *out_nodes, const int32_t *multiplicity,
                           const uint8_t *is_in_node, const uint64_t *cand_start, const uint32_t *cand_count,
                           const uint64_t *pool, mtg_pair **pairs_out) {
    std::vector<Pair> p = replay_claims(g->g, n_sources, out_nodes, multiplicity, is_in_node, cand_start, cand_count, pool);
    static_assert(sizeof(Pair) == sizeof(mtg_pair), "pair layout");
    mtg_pair *out = (mtg_pair *)std::malloc(std::max<size_t>(p.size(), 1) * sizeof(mtg_pair));
    if (!out) MTG_DIE("out of memory");
    if (!p.empty()) std::memcpy(out, p.data(), p.size() * sizeof(mtg_pair));
    *pairs_out = out;
    return p.size();
}
void mtg_free(void *p) { std::free(p); }

uint64_t mtg_insert_pair_edges(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs) {
    return insert_pair_edges(g->g, reinterpret_cast<const Pair *>(pairs), n_pairs);
}
uint64_t mtg_make_eulerian(mtg_graph *g, uint64_t dummy_edge_id, uint64_t k) { return make_eulerian(g->g, dummy_edge_id, k); }
mtg_walks *mtg_euler_cycles(const mtg_graph *g) { return new mtg_walks{euler_cycles(g->g)}; }
mtg_walks *mtg_euler_cycles_records(const mtg_graph *g, int record_format) {
    const HostGraph &h = g->g;
    if (record_format == 0) return new mtg_walks{euler_cycles(h)};
    if (record_format < 1 || record_format > 3) MTG_DIE("mtg_euler_cycles_records: unknown record format %d", record_format);
    h.ensure_linked();
    const uint64_t V = h.node_count(), E = h.edge_count();
    std::vector<LeanNode> lean(V);
    std::vector<uint32_t> ext_eid, ext_to;
    for (uint64_t n = 0; n < V; n++) {  // what lean_build_kernel writes (finish_device.hip): own adjacency, newest edge first
        LeanNode &r = lean[n];
        if (h.out_deg[n] > 65535) MTG_DIE("mtg_euler_cycles_records: out-degree beyond 65535");
        r.deg = (uint16_t)h.out_deg[n];
        r.pos = 0;
        r.ext_begin = (uint32_t)ext_eid.size();
        for (int i = 0; i < 3; i++) r.eid[i] = r.to[i] = NONE;
        uint32_t i = 0;
        for (uint32_t e = h.head_out[n]; e != NONE; e = h.e_next_out[e], i++) {
            if (i < 3) { r.eid[i] = e; r.to[i] = h.e_to[e]; }
            else { ext_eid.push_back(e); ext_to.push_back(h.e_to[e]); }
        }
    }
    if (record_format == 1)
        return new mtg_walks{euler_cycles_lean(lean.data(), V, ext_eid.data(), ext_to.data(), h.e_from.data(), h.e_to.data(), E, &h.arena)};
    if (record_format == 3) {
        HugeBuf<EulerNode2> mid(V, &h.arena);
        return new mtg_walks{euler_cycles_from_lean_mid(lean.data(), mid.p, V, ext_eid.data(), ext_to.data(), h.e_from.data(), h.e_to.data(), E, &h.arena)};
    }
    HugeBuf<EulerNode3> wide(V, &h.arena);
    return new mtg_walks{euler_cycles_from_lean(lean.data(), wide.p, V, ext_eid.data(), ext_to.data(), h.e_from.data(), h.e_to.data(), E, &h.arena)};
}
mtg_walks *mtg_euler_cycles_device(const mtg_graph *g, int device_id) {
    return new mtg_walks{device_euler_cycles(g->g, device_id, &g_last_euler_kernel_ms)};
}
double mtg_last_euler_kernel_ms(void) { return g_last_euler_kernel_ms; }
void mtg_set_euler_device_tuning(int splitter_bitmap) { device_euler_force_bitmap(splitter_bitmap); }
mtg_walks *mtg_cut_cycles(const mtg_graph *g, const mtg_walks *cycles, uint64_t k) {
    return new mtg_walks{cut_cycles(g->g, cycles->host(), k)};
}

// The device finish takes a graph that holds only its original edges and matched pairs shorter than k (what the claim loop
// produces); anything else (a graph Eulerised before, pairs from elsewhere) goes through the generic host stages.
static bool use_device_finish(const HostGraph &g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config &cfg) {
    if (cfg.finish_stage == MTG_FINISH_HOST) return false;
    const bool forced = cfg.finish_stage == MTG_FINISH_DEVICE;
    if (device_count() <= cfg.device_ids[0]) {
        if (forced) MTG_DIE("finish_stage = MTG_FINISH_DEVICE, but no MI355X/HIP device %d is visible", cfg.device_ids[0]);
        return false;
    }
    bool ok = g.edge_count() == g.n_original_edges && g.first_breaking_edge == UINT64_MAX && cfg.k <= 0xFFFFFFFFull;
    for (uint64_t i = 0; i < n_pairs && ok; i++) ok = pairs[i].distance < cfg.k && pairs[i].out_node < g.node_count() && pairs[i].in_node < g.node_count();
    if (!ok && forced) MTG_DIE("finish_stage = MTG_FINISH_DEVICE needs a graph without dummy edges and matched pairs shorter than k");
    return ok;
}
static mtg_walks *finish_on_device(HostGraph &g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config &cfg, const mtg_pair *d_pairs_resident = nullptr) {
    log_info("Making graph Eulerian by adding breaking dummy edges");
    log_info("Finding Eulerian bicycle");
    double *t = g_last_finish_times;
    if (g_clib_sink) g_clib_sink_used = true;
    ResidentTigs *resident = nullptr;
    Walks host_walks = device_finish(g, reinterpret_cast<const Pair *>(pairs), n_pairs, cfg.k, cfg.device_ids[0], cfg.euler_mode, t, d_pairs_resident, g_clib_sink,
                                     g_clib_sink ? nullptr : &resident);
    mtg_walks *tigs = new mtg_walks(std::move(host_walks), resident);
    g_phase[5] = t[0] + t[1];
    g_phase[6] = t[2];
    g_phase[7] = t[3];
    g_last_euler_kernel_ms = t[4];
    return tigs;
}

static mtg_walks *eulerise_and_cut(HostGraph &g, uint64_t dummy_edge_id, const mtg_config &cfg) {
    const uint64_t k = cfg.k;
    double t0 = now_s();
    log_info("Making graph Eulerian by adding breaking dummy edges");
    make_eulerian(g, dummy_edge_id, k);
    if (!is_eulerian(g)) MTG_DIE("Failed to make the graph Eulerian. (greedytigs/mod.rs:714)");
    double t1 = now_s();
    g_phase[5] += t1 - t0;
    log_info("Finding Eulerian bicycle");
    Walks cycles = euler_cycles_by_mode(g, cfg);
    double t2 = now_s();
    g_phase[6] = t2 - t1;
    log_info("Found %zu Eulerian bicycles", cycles.limits.size());
    mtg_walks *tigs = new mtg_walks{cut_cycles(g, cycles, k)};
    g_phase[7] = now_s() - t2;
    return tigs;
}

void mtg_config_init(mtg_config *cfg, uint64_t threads, uint64_t k) {  // GreedytigAlgorithmConfiguration::new, greedytigs/mod.rs:62-72
    if (!cfg) MTG_DIE("mtg_config_init: null configuration");
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->threads = threads;
    cfg->k = k;
    cfg->staged_parallelism_divisor = 0.0;
    cfg->resource_limit_factor = 0;
    cfg->node_weight_array_type = MTG_NODE_WEIGHT_HASHBROWN_HASH_MAP;
    cfg->heap_type = MTG_HEAP_STD_BINARY_HEAP;
    cfg->performance_data_type = MTG_PERFORMANCE_DATA_NONE;
    cfg->euler_mode = MTG_EULER_HOST_REFERENCE_ORDER;
    cfg->finish_stage = MTG_FINISH_AUTO;
    cfg->n_devices = 1;
    cfg->device_ids[0] = 0;
}

mtg_walks *mtg_finish_greedytigs_cfg(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg) {
    check_config(cfg, "mtg_finish_greedytigs_cfg");
    if (!g || !g->g.built) MTG_DIE("mtg_finish_greedytigs_cfg: graph is not built");
    if (n_pairs && !pairs) MTG_DIE("mtg_finish_greedytigs_cfg: null pairs");
    if (use_device_finish(g->g, pairs, n_pairs, *cfg)) {
        mtg_walks *tigs = finish_on_device(g->g, pairs, n_pairs, *cfg);
        log_info("Found %zu greedytigs", tig_count(tigs));
        return tigs;
    }
    double t0 = now_s();
    const uint64_t dummy_edge_id = insert_pair_edges(g->g, reinterpret_cast<const Pair *>(pairs), n_pairs);
    g_phase[5] = now_s() - t0;
    mtg_walks *tigs = eulerise_and_cut(g->g, dummy_edge_id, *cfg);
    log_info("Found %zu greedytigs", tig_count(tigs));
    return tigs;
}
mtg_walks *mtg_finish_device(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg) {
    check_config(cfg, "mtg_finish_device");
    if (!g || !g->g.built) MTG_DIE("mtg_finish_device: graph is not built");
    if (n_pairs && !pairs) MTG_DIE("mtg_finish_device: null pairs");
    mtg_config forced = *cfg;
    forced.finish_stage = MTG_FINISH_DEVICE;
    (void)use_device_finish(g->g, pairs, n_pairs, forced);
    return finish_on_device(g->g, pairs, n_pairs, forced);
}
mtg_walks *mtg_finish_greedytigs_resident(mtg_graph *g, mtg_device *d, const mtg_config *cfg) {
    check_config(cfg, "mtg_finish_greedytigs_resident");
    if (!g || !g->g.built || !d) MTG_DIE("mtg_finish_greedytigs_resident: graph is not built / null device");
    // (the resident pairs index the graph d was built from, and their distances are below d's k: the host path checks every pair
    // it is handed, here the identity of the graph stands for that)
    if (!device_matches(d->d, g->g, cfg->k))
        MTG_DIE("mtg_finish_greedytigs_resident: the device copy was built from another graph (node / edge counts differ) or for another k than cfg->k = %llu",
                (unsigned long long)cfg->k);
    uint64_t n_pairs = 0;
    const mtg_pair *d_pairs = device_resident_pairs(d->d, &n_pairs);
    if (use_device_finish(g->g, nullptr, 0, *cfg) && device_id_of(d->d) == cfg->device_ids[0]) {
        mtg_walks *tigs = finish_on_device(g->g, nullptr, n_pairs, *cfg, d_pairs);
        log_info("Found %zu greedytigs", tig_count(tigs));
        return tigs;
    }
    // host finish, or a finish on another GPU: through the host, like mtg_finish_greedytigs_cfg
    mtg_pair *pairs = nullptr;
    n_pairs = device_download_pairs(d->d, &pairs);
    mtg_walks *tigs = mtg_finish_greedytigs_cfg(g, pairs, n_pairs, cfg);
    std::free(pairs);
    return tigs;
}
void mtg_release_device_memory(int device_id) { device_release_memory(device_id); }
void mtg_set_default_device(int device_id) {
    if (device_id < 0 || device_id >= 64) MTG_DIE("mtg_set_default_device: device id %d out of range", device_id);
    device_set_default(device_id);
}
void mtg_set_reserve_ahead(int on) { device_set_reserve_ahead(on); }
void mtg_set_finish_tuning(int records, int flags, int64_t record_delay_us) {
    if (records < 0 || records > 3) MTG_DIE("mtg_set_finish_tuning: unknown record format %d", records);
    device_set_finish_tuning(records, flags, (long)record_delay_us);
}
uint64_t mtg_device_memory_held(int device_id) { return device_memory_held(device_id); }
void mtg_device_arena_stats(int device_id, uint64_t out[4], int reset_peak) {
    if (device_id < 0 || device_id >= 64) MTG_DIE("mtg_device_arena_stats: device id %d out of range", device_id);
    device_arena_stats(device_id, out);
    if (reset_peak) device_arena_reset_peak(device_id);
}
void mtg_graph_release_device_cache(mtg_graph *g) {
    if (g) device_release_graph_cache(g->g);
}
void mtg_last_finish_device_times(double out[6]) {
    for (int i = 0; i < 6; i++) out[i] = g_last_finish_times[i];
}
void mtg_last_finish_device_stage_ms(double out[6]) {
    const double *t = g_last_finish_times;
    out[0] = t[6]; out[1] = t[7]; out[2] = t[4]; out[3] = t[8]; out[4] = t[9]; out[5] = t[10];
}
mtg_graph *mtg_synth_g_csr(uint64_t n_binodes, uint64_t n_self_mirrors, uint64_t n_unitigs, uint64_t seed, uint64_t k,
                           const uint64_t *weight_thresholds, uint64_t n_thresholds, int max_degree, int device_id) {
    if (n_thresholds && !weight_thresholds) MTG_DIE("mtg_synth_g_csr: null thresholds");
    HostGraph *h = device_synth_g_csr(n_binodes, n_self_mirrors, n_unitigs, seed, k, weight_thresholds, n_thresholds, max_degree, device_id);
    mtg_graph *g = new mtg_graph{std::move(*h)};
    delete h;
    return g;
}
mtg_walks *mtg_finish_greedytigs(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, uint64_t k) {
    mtg_config cfg;
    mtg_config_init(&cfg, 1, k);
    return mtg_finish_greedytigs_cfg(g, pairs, n_pairs, &cfg);
}
mtg_walks *mtg_compute_eulertigs_cfg(mtg_graph *g, const mtg_config *cfg) {
    check_config(cfg, "mtg_compute_eulertigs_cfg");
    if (!g || !g->g.built) MTG_DIE("mtg_compute_eulertigs_cfg: graph is not built");
    g_phase[5] = 0;
    mtg_walks *tigs = use_device_finish(g->g, nullptr, 0, *cfg) ? finish_on_device(g->g, nullptr, 0, *cfg)
                                                                : eulerise_and_cut(g->g, 0, *cfg);  // eulertigs/mod.rs:101-102
    log_info("Found %zu eulertigs", tig_count(tigs));
    return tigs;
}
mtg_walks *mtg_compute_eulertigs(mtg_graph *g, uint64_t k) {
    mtg_config cfg;
    mtg_config_init(&cfg, 1, k);
    return mtg_compute_eulertigs_cfg(g, &cfg);
}

uint64_t mtg_walks_count(const mtg_walks *w) { return w->count(); }
uint64_t mtg_walks_total_edges(const mtg_walks *w) { return w->total_edges(); }
void mtg_walks_export(const mtg_walks *w, uint64_t *limits, uint32_t *edges) {
    const Walks &h = w->host();
    if (limits && !h.limits.empty()) std::memcpy(limits, h.limits.data(), h.limits.size() * 8);
    if (edges && !h.edges.empty()) std::memcpy(edges, h.edges.data(), h.edges.size() * 4);
}
void mtg_walks_data(const mtg_walks *w, const uint64_t **limits, const uint32_t **edges) {
    const Walks &h = w->host();
    if (limits) *limits = h.limits.data();
    if (edges) *edges = h.edges.data();
}
void mtg_walks_free(mtg_walks *w) { delete w; }
mtg_walks *mtg_walks_from_arrays(uint64_t n_walks, const uint64_t *limits, const uint32_t *edges) {
    if (n_walks && (!limits || !edges)) MTG_DIE("mtg_walks_from_arrays: null argument");
    mtg_walks *w = new mtg_walks();
    uint64_t prev = 0;
    for (uint64_t i = 0; i < n_walks; i++) {
        if (limits[i] < prev) MTG_DIE("mtg_walks_from_arrays: limits must be non-decreasing");
        prev = limits[i];
    }
    w->w.limits.assign(limits, limits + n_walks);
    w->w.edges.assign(edges, edges + prev);
    return w;
}

uint64_t mtg_flatten_clib(const mtg_graph *g, const mtg_walks *tigs, int64_t *tigs_edge_out, uint64_t *tigs_insert_out,
                          uint64_t *tigs_out_limits) {
    return flatten_clib(g->g, tigs->host(), tigs_edge_out, tigs_insert_out, tigs_out_limits);
}

uint64_t mtg_write_walks_fasta(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                               const char *unitig_seqs, const uint64_t *seq_offsets, char **fasta_out) {
    if (!g || !fasta_out || (n_walks && (!limits || !edges)) || !unitig_seqs || !seq_offsets)
        MTG_DIE("mtg_write_walks_fasta: null argument");
    return write_walks_fasta(g->g, n_walks, limits, edges, k, unitig_seqs, seq_offsets, fasta_out);
}

uint64_t mtg_write_walks_gfa(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                             const char *unitig_seqs, const uint64_t *seq_offsets, const char *header, char **gfa_out) {
    if (!g || !gfa_out || (n_walks && (!limits || !edges)) || !unitig_seqs || !seq_offsets)
        MTG_DIE("mtg_write_walks_gfa: null argument");
    return write_walks_text(g->g, n_walks, limits, edges, k, unitig_seqs, seq_offsets, true, header, gfa_out);
}

// f-1 on the device: the same text from spell_device.hip (2-bit packed store, output-centric kernel)
static thread_local double g_last_spell_kernel_ms = 0;
static thread_local uint64_t g_last_spell_bytes = 0;
uint64_t mtg_write_walks_text_device(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                                     const char *unitig_seqs, const uint64_t *seq_offsets, int gfa, const char *gfa_header, int device_id,
                                     char **text_out) {
    if (!g || !text_out || (n_walks && (!limits || !edges)) || !unitig_seqs || !seq_offsets)
        MTG_DIE("mtg_write_walks_text_device: null argument");
    return device_write_walks_text(g->g, n_walks, limits, edges, k, unitig_seqs, seq_offsets, gfa != 0, gfa_header, device_id, text_out,
                                   &g_last_spell_kernel_ms, &g_last_spell_bytes);
}
double mtg_last_spell_kernel_ms(void) { return g_last_spell_kernel_ms; }
uint64_t mtg_last_spell_bytes(void) { return g_last_spell_bytes; }

uint64_t mtg_write_duplication_bitvector(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges,
                                         char **text_out) {
    if (!g || !text_out || (n_walks && (!limits || !edges))) MTG_DIE("mtg_write_duplication_bitvector: null argument");
    return write_duplication_bitvector(g->g, n_walks, limits, edges, text_out);
}
uint64_t mtg_write_tigs_duplication_bitvector_file(const mtg_graph *g, const mtg_walks *tigs, const char *path) {
    if (!g || !tigs || !path) MTG_DIE("mtg_write_tigs_duplication_bitvector_file: null argument");
    char *buf = nullptr;
    const uint64_t n = write_duplication_bitvector(g->g, tigs->host().limits.size(), tigs->host().limits.data(), tigs->host().edges.data(), &buf);
    write_file(path, buf, n, 0);  // the reference writes this file uncompressed (implementation/mod.rs:665)
    std::free(buf);
    return n;
}

// ---- f-2: BCALM2 input route + FASTA file output ----
struct mtg_unitigs { UnitigStore *s; };

mtg_graph *mtg_read_bcalm2(const char *path, uint64_t k, mtg_unitigs **unitigs_out) {
    if (!unitigs_out) MTG_DIE("mtg_read_bcalm2: null argument");
    UnitigStore *st = nullptr;
    HostGraph *h = read_bcalm2(path, k, &st);
    mtg_graph *g = new mtg_graph{std::move(*h)};
    delete h;
    *unitigs_out = new mtg_unitigs{st};
    return g;
}
uint64_t mtg_unitigs_count(const mtg_unitigs *u) { return u->s->off.size() - 1; }
const char *mtg_unitigs_data(const mtg_unitigs *u) { return u->s->data.data(); }
const uint64_t *mtg_unitigs_offsets(const mtg_unitigs *u) { return u->s->off.data(); }
void mtg_unitigs_free(mtg_unitigs *u) {
    if (!u) return;
    delete u->s;
    delete u;
}
uint64_t mtg_write_tigs_text_file_device(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs, int gfa,
                                         const char *gfa_header, const char *path, int compression_level, int device_id) {
    if (!g || !tigs || !unitigs || !path) MTG_DIE("mtg_write_tigs_text_file_device: null argument");
    char *buf = nullptr;
    // (tigs a finish on the same GPU left in HBM are spelled from there: no download, no upload)
    const ResidentTigs *res = tigs->resident_on(device_id);
    const uint64_t n = res ? device_write_walks_text(g->g, 0, nullptr, nullptr, k, unitigs->s->data.data(), unitigs->s->off.data(), gfa != 0, gfa_header, device_id, &buf,
                                                     &g_last_spell_kernel_ms, &g_last_spell_bytes, res)
                       : device_count() > device_id && device_id >= 0
                           ? device_write_walks_text(g->g, tigs->host().limits.size(), tigs->host().limits.data(), tigs->host().edges.data(), k, unitigs->s->data.data(),
                                                     unitigs->s->off.data(), gfa != 0, gfa_header, device_id, &buf, &g_last_spell_kernel_ms, &g_last_spell_bytes)
                           : write_walks_text(g->g, tigs->host().limits.size(), tigs->host().limits.data(), tigs->host().edges.data(), k,
                                              unitigs->s->data.data(), unitigs->s->off.data(), gfa != 0, gfa_header, &buf);
    write_file(path, buf, n, compression_level);
    std::free(buf);
    return n;
}
uint64_t mtg_write_tigs_fasta_file(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs,
                                   const char *path, int compression_level) {
    return mtg_write_tigs_text_file_device(g, tigs, k, unitigs, 0, nullptr, path, compression_level, 0);
}
uint64_t mtg_write_tigs_gfa_file(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs,
                                 const char *header, const char *path, int compression_level) {
    return mtg_write_tigs_text_file_device(g, tigs, k, unitigs, 1, header, path, compression_level, 0);
}

// ---- optimal matchtigs around the external matcher (matchtigs/mod.rs:150-940) ----
struct mtg_matching {
    MatchingInstance *m;
};
mtg_matching *mtg_matching_instance(const mtg_graph *g, const mtg_config *cfg) {
    check_config(cfg, "mtg_matching_instance");
    if (!g || !g->g.built) MTG_DIE("mtg_matching_instance: graph is not built");
    const uint64_t k = cfg->k;
    double t0 = now_s();
    mtg_device *dev = mtg_device_create(g, k, cfg->device_ids[0]);
    log_info("Collecting nodes with missing incoming or outgoing edges");
    const uint64_t S = mtg_classify(dev, nullptr);
    std::vector<uint32_t> out_nodes(S);
    std::vector<int32_t> mult(g->g.node_count());
    mtg_classify_download(dev, nullptr, out_nodes.data(), mult.data(), nullptr);
    log_info("Found %llu nodes with missing outgoing edges", (unsigned long long)S);
    log_info("Computing shortest paths between nodes with missing outgoing and nodes with missing incoming edges");
    std::vector<uint64_t> cand_start, pool;
    std::vector<uint32_t> cand_count;
    device_candidates_to_host(dev->d, nullptr, cand_start, cand_count, pool);
    mtg_device_free(dev);
    double t1 = now_s();
    MatchingInstance *m = build_matching_instance(g->g, k, S, out_nodes.data(), mult.data(), cand_start.data(), cand_count.data(), pool.data());
    log_info("Took %.6fs for computing paths and getting edges, of this %.6fs are from dijkstra", now_s() - t0, t1 - t0);
    log_info("Found %llu nodes and %zu edges", (unsigned long long)m->transformed_node_count, m->edge_n2.size());
    log_info("Matching problem contains %llu edges that originate from %llu mirror biedges", (unsigned long long)m->mirror_expanded_biedges,
             (unsigned long long)m->mirror_biedges);
    log_info("Found %llu relevant WCCs", (unsigned long long)m->wcc_amount);
    return new mtg_matching{m};
}
mtg_matching *mtg_matching_instance_from_lists(const mtg_graph *g, uint64_t k, uint64_t n_sources, const uint32_t *out_nodes,
                                               const int32_t *multiplicity, const uint64_t *cand_start, const uint32_t *cand_count,
                                               const uint64_t *pool) {
    if (!g || !g->g.built) MTG_DIE("mtg_matching_instance_from_lists: graph is not built");
    if (k < 1) MTG_DIE("mtg_matching_instance_from_lists: k must be >= 1");
    if (!multiplicity || (n_sources && (!out_nodes || !cand_start || !cand_count))) MTG_DIE("mtg_matching_instance_from_lists: null argument");
    return new mtg_matching{build_matching_instance(g->g, k, n_sources, out_nodes, multiplicity, cand_start, cand_count, pool)};
}
void mtg_matching_get_stats(const mtg_matching *m, mtg_matching_stats *out) {
    if (!m || !out) MTG_DIE("mtg_matching_get_stats: null argument");
    out->transformed_node_count = m->m->transformed_node_count;
    out->edge_count = m->m->edge_n2.size();
    out->wcc_amount = m->m->wcc_amount;
    out->matching_node_count = m->m->matching_node_count;
    out->matching_edge_count = m->m->matching_edge_count;
    out->mirror_biedges = m->m->mirror_biedges;
    out->mirror_expanded_biedges = m->m->mirror_expanded_biedges;
}
uint64_t mtg_matching_write(const mtg_matching *m, const char *path) {
    if (!m || !path) MTG_DIE("mtg_matching_write: null argument");
    log_info("Outputting matching problem to \"%s\"", path);
    return write_matching_instance(*m->m, path);
}
uint64_t mtg_matching_read_solution(const mtg_matching *m, const char *solution_path, mtg_pair **pairs_out) {
    if (!m || !solution_path || !pairs_out) MTG_DIE("mtg_matching_read_solution: null argument");
    std::vector<Pair> p = read_matching_solution(*m->m, solution_path);
    mtg_pair *out = (mtg_pair *)std::malloc(std::max<size_t>(p.size(), 1) * sizeof(mtg_pair));
    if (!out) MTG_DIE("out of memory");
    if (!p.empty()) std::memcpy(out, p.data(), p.size() * sizeof(mtg_pair));
    *pairs_out = out;
    return p.size();
}
void mtg_matching_free(mtg_matching *m) {
    if (!m) return;
    delete m->m;
    delete m;
}
mtg_walks *mtg_finish_matchtigs_cfg(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg) {
    check_config(cfg, "mtg_finish_matchtigs_cfg");
    if (!g || !g->g.built) MTG_DIE("mtg_finish_matchtigs_cfg: graph is not built");
    const uint64_t k = cfg->k;
    uint64_t bidirected = 0;
    for (uint64_t i = 0; i < n_pairs; i++) bidirected += pairs[i].out_node == g->g.mirror[pairs[i].in_node] ? 2 : 0;
    const uint64_t dummy_edge_id = insert_pair_edges(g->g, reinterpret_cast<const Pair *>(pairs), n_pairs);  // :797-808
    log_info("Inserted %llu matched edges", (unsigned long long)(2 * n_pairs));
    if (bidirected) log_info("Inserted %llu bidirected loops", (unsigned long long)bidirected);
    log_info("Making graph Eulerian by completing unmatched nodes");
    make_eulerian(g->g, dummy_edge_id, k);  // :831
    if (!is_eulerian(g->g)) MTG_DIE("Failed to make the graph Eulerian. (matchtigs/mod.rs:849)");
    log_info("Finding Eulerian bicycle");
    Walks cycles = euler_cycles_by_mode(g->g, *cfg);
    log_info("Found %zu Eulerian bicycles", cycles.limits.size());
    uint64_t begin = 0;
    for (uint64_t c = 0; c < cycles.limits.size(); c++) {  // :870-886
        uint64_t longest = 0;
        for (uint64_t i = begin; i < cycles.limits[c]; i++)
            if (g->g.is_dummy(cycles.edges[i])) longest = std::max<uint64_t>(longest, g->g.weight(cycles.edges[i]));
        if (longest > 0 && longest < k) MTG_DIE("Eulerian bicycle contains at least one dummy edge, but no breaking edge (matchtigs/mod.rs:883)");
        begin = cycles.limits[c];
    }
    mtg_walks *tigs = new mtg_walks{cut_cycles(g->g, cycles, k)};
    log_info("Found %zu matchtigs", (size_t)tigs->count());
    return tigs;
}
mtg_walks *mtg_compute_matchtigs_cfg(mtg_graph *g, const mtg_config *cfg) {
    check_config(cfg, "mtg_compute_matchtigs_cfg");
    if (!cfg->matching_file_prefix) MTG_DIE("mtg_compute_matchtigs_cfg: matching_file_prefix is null");
    if (!cfg->matcher_path) MTG_DIE("mtg_compute_matchtigs_cfg: matcher_path is null");
    mtg_matching *m = mtg_matching_instance(g, cfg);
    const std::string instance_path = std::string(cfg->matching_file_prefix) + ".minimalperfectmatching";  // :592-593
    const std::string solution_path = instance_path + ".solution";                                          // :722
    mtg_matching_write(m, instance_path.c_str());
    if (m->m->transformed_node_count != 0) {  // :724-741
        log_info("Running matcher at \"%s\"", cfg->matcher_path);
        const char *argv[] = {cfg->matcher_path, "-e", instance_path.c_str(), "-w", solution_path.c_str(), nullptr};
        pid_t pid = 0;
        const int rc = posix_spawnp(&pid, cfg->matcher_path, nullptr, nullptr, const_cast<char *const *>(argv), environ);  // searches PATH like Command::new (matchtigs/mod.rs:727)
        if (rc != 0) MTG_DIE("cannot start the matcher %s: %s", cfg->matcher_path, std::strerror(rc));
        int status = 0;
        while (waitpid(pid, &status, 0) < 0)
            if (errno != EINTR) MTG_DIE("waitpid on the matcher failed: %s", std::strerror(errno));
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) MTG_DIE("Matcher was unsuccessful: wait status %d (matchtigs/mod.rs:735)", status);
    } else {
        log_info("Nothing to match, generating empty output file");  // :742-746
        FILE *f = std::fopen(solution_path.c_str(), "w");
        if (!f) MTG_DIE("cannot create %s", solution_path.c_str());
        std::fputs("0 0\n", f);
        std::fclose(f);
    }
    log_info("Applying matcher result to graph");
    mtg_pair *pairs = nullptr;
    const uint64_t n_pairs = mtg_matching_read_solution(m, solution_path.c_str(), &pairs);
    mtg_matching_free(m);
    mtg_walks *tigs = mtg_finish_matchtigs_cfg(g, pairs, n_pairs, cfg);
    std::free(pairs);
    return tigs;
}

static mtg_walks *compute_tigs_cfg_body(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg);
mtg_walks *mtg_compute_tigs_cfg(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg) {
    mtg_walks *w = compute_tigs_cfg_body(g, tig_algorithm, cfg);
    // what the graph's constructor reserved ahead on its default GPU is given back if this call ran elsewhere
    if (tig_algorithm >= 3) device_drop_foreign_reservation(cfg->device_ids, cfg->n_devices);
    return w;
}
static mtg_walks *compute_tigs_cfg_body(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg) {
    check_config(cfg, "mtg_compute_tigs_cfg");
    if (!g || !g->g.built) MTG_DIE("mtg_compute_tigs: graph is not built");
    for (double &p : g_phase) p = 0;
    g_last_perf = mtg_dijkstra_performance_data{};
    const uint64_t k = cfg->k;
    switch (tig_algorithm) {
        case 1: {  // clib.rs:351-361: one walk per forward unitig edge
            mtg_walks *w = new mtg_walks();
            for (uint64_t e = 0; e < g->g.edge_count(); e += 2) {
                w->w.edges.push_back((uint32_t)e);
                w->w.limits.push_back(w->w.edges.size());
            }
            return w;
        }
        case 3:
            return mtg_compute_eulertigs_cfg(g, cfg);
        case 5: {
            double t0 = now_s();
            // one resident copy of the graph per configured GPU (replicas + source sharding), built concurrently
            const int n_dev = cfg->n_devices;
            std::vector<mtg_device *> devs((size_t)n_dev, nullptr);
            {
                std::vector<std::thread> th;
                // (searched once: no goal-directed lower bounds -- their precompute costs several times what it saves one search)
                for (int i = 1; i < n_dev; i++) th.emplace_back([&, i]() { devs[(size_t)i] = mtg_device_create_opts(g, k, cfg->device_ids[i], MTG_DEVICE_NO_LOWER_BOUNDS); });
                devs[0] = mtg_device_create_opts(g, k, cfg->device_ids[0], MTG_DEVICE_NO_LOWER_BOUNDS);
                for (auto &t : th) t.join();
            }
            mtg_device *dev = devs[0];
            if (cfg->performance_data_type != MTG_PERFORMANCE_DATA_COMPLETE)  // (the counters are a second search)
                for (mtg_device *x : devs) device_set_single_use(x->d);
            double t1 = now_s();
            g_phase[0] = t1 - t0;
            log_info("Collecting nodes with missing incoming or outgoing edges");
            uint64_t S = 0;
            {
                std::vector<std::thread> th;
                for (int i = 1; i < n_dev; i++) th.emplace_back([&, i]() { (void)mtg_classify(devs[(size_t)i], nullptr); });
                S = mtg_classify(dev, nullptr);
                for (auto &t : th) t.join();
            }
            double t2 = now_s();
            g_phase[1] = t2 - t1;
            log_info("Found %llu nodes with missing outgoing edges", (unsigned long long)S);
            // SSSP candidates (sharded over the devices) and the claim loop both run on the GPU; only the matched pairs come back
            // ... and stay in the HBM of the first GPU when the finish runs there too (the default)
            const bool resident = use_device_finish(g->g, nullptr, 0, *cfg) && device_id_of(dev->d) == cfg->device_ids[0];
            mtg_pair *pairs = nullptr;
            uint64_t n_pairs = 0;
            {
                std::vector<Device *> dd((size_t)n_dev);
                for (int i = 0; i < n_dev; i++) dd[(size_t)i] = devs[(size_t)i]->d;
                double gather_ms = 0;
                n_pairs = device_pairs_multi(dd.data(), n_dev, resident ? nullptr : &pairs, nullptr, &gather_ms);
                g_last_gather_ms = gather_ms;
            }
            double t3 = now_s();
            {   // [2] SSSP stage (+ gather over the devices), [4] claim replay, [3] pair download (0 when the pairs stay in HBM)
                double w[3];
                device_last_pairs_wall_s(dev->d, w);
                g_phase[4] = w[1];
                g_phase[3] = w[2];
                g_phase[2] = std::max(0.0, (t3 - t2) - w[1] - w[2]);
            }
            for (int i = 1; i < n_dev; i++) mtg_device_free(devs[(size_t)i]);
            if (cfg->performance_data_type == MTG_PERFORMANCE_DATA_COMPLETE) {  // greedytigs/mod.rs:647-673
                device_performance_data(dev->d, nullptr, &g_last_perf);
                const mtg_dijkstra_performance_data &p = g_last_perf;
                if (p.iterations)
                    log_info("Dijkstras had a factor of %.3f unnecessary heap elements", (double)p.unnecessary_heap_elements / (double)p.iterations);
                log_info("Dijktras had a maximum maximum heap size of %llu", (unsigned long long)p.max_max_heap_size);
                log_info("Dijktras had a maximum maximum distance array size of %llu", (unsigned long long)p.max_max_distance_array_size);
                if (p.dijkstras) {
                    log_info("Dijktras had an average maximum heap size of %.0f", (double)p.sum_max_heap_size / (double)p.dijkstras);
                    log_info("Dijktras had an average maximum heap size of %.0f", (double)p.sum_max_distance_array_size / (double)p.dijkstras);  // sic, :669-671
                }
            }
            mtg_pair *d_pairs = resident ? device_take_pairs(dev->d, nullptr) : nullptr;  // (ours now: the device graph can go before the finish)
            mtg_device_free(dev);
            log_info("Found %llu shortest paths", (unsigned long long)n_pairs);
            mtg_walks *tigs;
            if (resident) {
                tigs = finish_on_device(g->g, nullptr, n_pairs, *cfg, d_pairs);
                log_info("Found %zu greedytigs", tig_count(tigs));
                device_free_array(cfg->device_ids[0], d_pairs);
                if (std::getenv("MTG_DEBUG")) {
                    uint64_t a[4];
                    device_arena_stats(cfg->device_ids[0], a);
                    std::fprintf(stderr, "[mtg] device arena: %.2f GB in chunks (%llu taken from the driver so far), peak of live arrays %.2f GB, estimate for this graph %.2f GB\n",
                                 a[0] / 1e9, (unsigned long long)a[3], a[2] / 1e9, device_call_bytes_estimate(g->g.node_count(), g->g.n_original_edges, k) / 1e9);
                }
            } else {
                tigs = mtg_finish_greedytigs_cfg(g, pairs, n_pairs, cfg);
                std::free(pairs);
            }
            return tigs;
        }
        case 2:
            MTG_DIE("tig algorithm 2 (pathtigs) is outside the scope of the MI355X engine (SURVEY.md 2, row 11)");
        case 4:  // clib.rs:362-376: needs the external matcher the configuration names
            return mtg_compute_matchtigs_cfg(g, cfg);
        default:
            MTG_DIE("Unknown tigs algorithm identifier %llu", (unsigned long long)tig_algorithm);  // clib.rs:390
    }
    return nullptr;
}

// The whole path into a caller's clib.rs output arrays (sized as clib.rs:332-348: 2 E, 2 E and E entries for E original edges):
// what matchtigs_compute_tigs does after it has built its configuration. With a finish on the GPU the tigs never exist as walks on
// the host: the host threads that empty the download ring write the flattened form (clib.rs:393-407) straight into the caller's
// arrays, whose pages a helper thread touches while the GPU stages run.
uint64_t mtg_compute_tigs_clib(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg, int64_t *tigs_edge_out, uint64_t *tigs_insert_out,
                               uint64_t *tigs_out_limits) {
    check_config(cfg, "mtg_compute_tigs_clib");
    if (!g || !g->g.built) MTG_DIE("mtg_compute_tigs_clib: graph is not built");
    if (!tigs_edge_out || !tigs_insert_out || !tigs_out_limits) MTG_DIE("mtg_compute_tigs_clib: null output array");
    TigSink sink;
    sink.edge_out = tigs_edge_out;
    sink.insert_out = tigs_insert_out;
    sink.limits_out = tigs_out_limits;
    const uint64_t E0 = g->g.n_original_edges;
    std::thread toucher;
    if ((tig_algorithm == 5 || tig_algorithm == 3) && E0 >= (1u << 22) && use_device_finish(g->g, nullptr, 0, *cfg)) {
        // fresh output arrays cost a page fault per 4 KB when they are first written: a few threads take those faults now, beside the
        // device-graph build and the search (tigs hold one edge per unitig and a few matched dummies; a tig per four unitigs is plenty)
        const uint64_t n_e = E0 / 2 + E0 / 8, n_l = E0 / 4;
        // (advice only: where the caller's arrays are anonymous memory and the system hands out huge pages on request, a fault brings 2 MB)
        for (auto r : {std::make_pair((char *)tigs_edge_out, n_e * 8), std::make_pair((char *)tigs_insert_out, n_e * 8), std::make_pair((char *)tigs_out_limits, n_l * 8)}) {
            const uintptr_t lo = ((uintptr_t)r.first + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), hi = ((uintptr_t)r.first + r.second) & ~(uintptr_t)((2u << 20) - 1);
            if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
        }
        toucher = std::thread([=]() {
            const double t_touch = now_s();
            parallel_ranges((n_e * 8 + 4095) / 4096, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t pg = lo; pg < hi; pg++) {
                    *((volatile char *)tigs_edge_out + pg * 4096) = 0;
                    *((volatile char *)tigs_insert_out + pg * 4096) = 0;
                }
            }, 6);
            parallel_ranges((n_l * 8 + 4095) / 4096, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t pg = lo; pg < hi; pg++) *((volatile char *)tigs_out_limits + pg * 4096) = 0;
            }, 6);
            if (std::getenv("MTG_DEBUG")) std::fprintf(stderr, "[mtg] output arrays touched (%.2f GB) %.1f ms after the call began\n", (2 * n_e + n_l) * 8 / 1e9, (now_s() - t_touch) * 1e3);
        });
    }
    if (toucher.joinable()) sink.pretoucher = &toucher;  // joined by the finish before its first write into the arrays
    g_clib_sink = &sink;
    g_clib_sink_used = false;
    mtg_walks *tigs = mtg_compute_tigs_cfg(g, tig_algorithm, cfg);
    g_clib_sink = nullptr;
    if (toucher.joinable()) toucher.join();
    const uint64_t n = g_clib_sink_used ? sink.n_tigs : flatten_clib(g->g, tigs->host(), tigs_edge_out, tigs_insert_out, tigs_out_limits);
    delete tigs;
    return n;
}

mtg_walks *mtg_compute_tigs(mtg_graph *g, uint64_t tig_algorithm, uint64_t k, int device_id) {
    mtg_config cfg;
    mtg_config_init(&cfg, 1, k);
    cfg.device_ids[0] = device_id;
    return mtg_compute_tigs_cfg(g, tig_algorithm, &cfg);
}

void mtg_last_performance_data(mtg_dijkstra_performance_data *out) {
    if (out) *out = g_last_perf;
}

void mtg_last_phase_seconds(double out[8]) {
    for (int i = 0; i < 8; i++) out[i] = g_phase[i];
}

// ------------------------------------------------------------------------------------------------
// matchtigs.h: the reference's C-ABI (src/clib.rs)
// ------------------------------------------------------------------------------------------------
void matchtigs_initialise(void) {  // clib.rs:87-92
    g_log_initialised = true;
    log_info("Logging initialised successfully");
}

MatchtigsData *matchtigs_initialise_graph(size_t unitig_amount) {  // clib.rs:94-102
    HostGraph *h = builder_new(unitig_amount);
    MatchtigsData *d = new MatchtigsData{mtg_graph{std::move(*h)}};
    delete h;
    return d;
}

void matchtigs_merge_nodes(MatchtigsData *data, size_t unitig_a, bool strand_a, size_t unitig_b, bool strand_b) {  // clib.rs:135-170
    if (!data) MTG_DIE("matchtigs_merge_nodes: matchtigs_data is null");
    builder_merge(&data->graph.g, unitig_a, strand_a, unitig_b, strand_b);
}

void matchtigs_build_graph(MatchtigsData *data, const size_t *unitig_weights) {  // clib.rs:180-259
    if (!data) MTG_DIE("matchtigs_build_graph: matchtigs_data is null");
    static_assert(sizeof(size_t) == sizeof(uint64_t), "64-bit only");
    double t0 = now_s();
    builder_build(&data->graph.g, reinterpret_cast<const uint64_t *>(unitig_weights));
    log_info("Took %.6fs to build the tig graph", now_s() - t0);
}

size_t matchtigs_compute_tigs(MatchtigsData *data, size_t tig_algorithm, size_t threads, size_t k,
                              const char *matching_file_prefix, const char *matcher_path, ptrdiff_t *tigs_edge_out,
                              size_t *tigs_insert_out, size_t *tigs_out_limits) {  // clib.rs:280-410
    if (!data) MTG_DIE("matchtigs_compute_tigs: matchtigs_data is null");
    log_info("Computing tigs for k = %zu and %zu threads", k, threads);
    log_info("Graph has %llu nodes and %llu edges", (unsigned long long)data->graph.g.node_count(),
             (unsigned long long)data->graph.g.edge_count());
    if (!matching_file_prefix) MTG_DIE("assertion failed: !matching_file_prefix.is_null() (clib.rs:300)");
    if (!matcher_path) MTG_DIE("assertion failed: !matcher_path.is_null() (clib.rs:316)");
    if (!tigs_edge_out) MTG_DIE("assertion failed: !tigs_edge_out.is_null() (clib.rs:333)");
    if (!tigs_insert_out) MTG_DIE("assertion failed: !tigs_insert_out.is_null() (clib.rs:339)");
    if (!tigs_out_limits) MTG_DIE("assertion failed: !tigs_out_limits.is_null() (clib.rs:345)");
    static_assert(sizeof(ptrdiff_t) == sizeof(int64_t), "64-bit only");
    mtg_config cfg;  // clib.rs:378-389: staged None, factor 1, StdBinaryHeap, EpochNodeWeightArray, performance data None
    mtg_config_init(&cfg, threads, k);
    cfg.resource_limit_factor = 1;
    cfg.node_weight_array_type = MTG_NODE_WEIGHT_EPOCH_ARRAY;
    cfg.matching_file_prefix = matching_file_prefix;  // clib.rs:362-376
    cfg.matcher_path = matcher_path;
    const uint64_t n = mtg_compute_tigs_clib(&data->graph, tig_algorithm, &cfg, reinterpret_cast<int64_t *>(tigs_edge_out),
                                             reinterpret_cast<uint64_t *>(tigs_insert_out), reinterpret_cast<uint64_t *>(tigs_out_limits));
    free_graph_object(data, data->graph.g);  // Box::from_raw at clib.rs:291: the handle is consumed
    return n;
}

}  // extern "C"
