// host_graph.hpp -- host-side edge-centric bigraph of the MI355X greedy-matchtigs engine.
//
// Replaces the container the reference gets from bigraph/traitgraph/petgraph
// (`NodeBigraphWrapper<PetGraph<(), EdgeData>>`, /root/reference/src/clib.rs:37, src/bin.rs:349-355).
// Layout is flat structure-of-arrays; edges are always appended as (edge, mirror edge) pairs, so the
// mirror of edge e is e ^ 1 (the reference looks it up by data equality).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <sys/mman.h>

#include "huge_arena.hpp"

#define MTG_DIE(...)                                  \
    do {                                              \
        std::fprintf(stderr, "libmatchtigs: ");       \
        std::fprintf(stderr, __VA_ARGS__);            \
        std::fprintf(stderr, "\n");                   \
        std::abort();                                 \
    } while (0)

namespace mtg {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// malloc for the large result arrays (matched pairs, tigs, edge arrays): from 8 MB on, 2-MB aligned and advised to sit on transparent
// huge pages. Fresh 4-KB pages cost a fault per page on first touch, and the first touch of these arrays is a copy: filling 73 MB
// of pairs ran at 2 GB/s per thread (4.3 of the 4.6 ms the pair download took at 2^27) -- one fault per 2 MB instead of 512.
// free() releases either kind.
inline void advise_huge(void *p, size_t bytes) { (void)madvise(p, bytes, MADV_HUGEPAGE); }  // (advice only: failure is harmless)
inline void *big_malloc(size_t bytes) {
    constexpr size_t HUGE = 2u << 20;
    if (bytes < (8u << 20)) return std::malloc(bytes ? bytes : 1);
    const size_t rounded = (bytes + HUGE - 1) / HUGE * HUGE;
    void *p = std::aligned_alloc(HUGE, rounded);
    static const bool no_thp = std::getenv("MTG_NO_THP") != nullptr;  // (measurements only)
    if (p && !no_thp) advise_huge(p, rounded);
    return p;
}

// A few recently released large blocks are kept for the next request of about the same size: a caller that iterates (one tig set
// after the other) otherwise pays the page faults of every fresh array and the unmapping of every released one -- at 2^27 the two
// tig arrays (0.56 GB) cost ~8 ms to fault in and 21 ms to give back, of a 160-ms device-mode step. At most 8 blocks / 2 GB.
struct BigBlockCache {
    struct Entry { void *p; size_t bytes; };
    std::mutex m;
    Entry e[8];
    int n = 0;
    size_t total = 0;
    static constexpr size_t LIMIT = 2ull << 30, HUGE = 2u << 20, SMALL = 8u << 20;
    static size_t rounded(size_t bytes) { return (bytes + HUGE - 1) / HUGE * HUGE; }
    void *take(size_t bytes) {  // best fit among the kept blocks, at most a quarter larger than asked for
        const size_t want = rounded(bytes);
        std::lock_guard<std::mutex> lock(m);
        int best = -1;
        for (int i = 0; i < n; i++)
            if (e[i].bytes >= want && e[i].bytes <= want + want / 4 && (best < 0 || e[i].bytes < e[best].bytes)) best = i;
        if (best < 0) return nullptr;
        void *p = e[best].p;
        total -= e[best].bytes;
        e[best] = e[--n];
        return p;
    }
    void give(void *p, size_t bytes) {  // (bytes = what the block was asked for: its rounded size is a lower bound of its real one)
        const size_t have = rounded(bytes);
        void *drop[9];
        int n_drop = 0;
        {
            std::lock_guard<std::mutex> lock(m);
            if (have > LIMIT) drop[n_drop++] = p;
            else {
                while (n == 8 || total + have > LIMIT) {  // oldest first
                    drop[n_drop++] = e[0].p;
                    total -= e[0].bytes;
                    for (int i = 1; i < n; i++) e[i - 1] = e[i];
                    n--;
                }
                e[n++] = Entry{p, have};
                total += have;
            }
        }
        for (int i = 0; i < n_drop; i++) std::free(drop[i]);
    }
};
inline BigBlockCache &big_block_cache() {
    static BigBlockCache *c = new BigBlockCache;  // (never destroyed: blocks may be released during static destruction)
    return *c;
}

// std::vector whose resize() leaves new elements uninitialised: the bulk edge insertion fills them from host threads,
// and a sequential zero-fill of hundreds of MB first would cost as much as the fill itself. Large blocks come from big_malloc
// or from the cache of recently released ones.
template <typename T>
struct DefaultInitAlloc : std::allocator<T> {
    template <typename U>
    struct rebind { using other = DefaultInitAlloc<U>; };
    using std::allocator<T>::allocator;
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void *p = bytes >= BigBlockCache::SMALL ? big_block_cache().take(bytes) : nullptr;
        if (!p) p = big_malloc(bytes);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n) noexcept {
        const size_t bytes = n * sizeof(T);
        if (bytes >= BigBlockCache::SMALL) big_block_cache().give(p, bytes);
        else std::free(p);
    }
    template <typename U>
    void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A>
    void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <typename T>
using PodVec = std::vector<T, DefaultInitAlloc<T>>;

struct Pair {
    uint32_t out_node;
    uint32_t in_node;
    uint64_t distance;
};

struct Walks {
    PodVec<uint64_t> limits;  // exclusive end of walk i
    PodVec<uint32_t> edges;   // edge ids
};

// Device-resident copy of what never changes after the graph is built -- the from-node of every ORIGINAL edge and the mirror
// array -- left behind by the first stage that uploaded them (device.hip / finish_device.hip) for the stages and steps that
// follow on the same GPU. Owned by the graph, freed with it (free_fn is set by the HIP translation unit that made it).
struct DeviceEdgeCache {
    int device = -1;
    uint64_t n_edges = 0, n_nodes = 0;
    void *d_from = nullptr, *d_mirror = nullptr;
    // the original darts bucketed by from-node (row0[V + 1], adj0[n_edges] in ascending dart id): a static function of the graph,
    // built by the first device finish on it; the buckets of an Eulerised dart set are this plus the dummy darts' (whose ids are all larger)
    void *d_row0 = nullptr, *d_adj0 = nullptr;
    void (*free_fn)(DeviceEdgeCache *) = nullptr;
};
struct DeviceEdgeCacheDeleter {
    void operator()(DeviceEdgeCache *c) const {
        if (c && c->free_fn) c->free_fn(c);
        delete c;
    }
};

struct HostGraph {
    // ---- clib.rs builder state (src/clib.rs:97-170) ----
    bool has_builder = false;
    bool built = false;
    uint64_t unitig_amount = 0;
    std::vector<PodVec<uint64_t>> link_chunks;  // the links reported so far, packed (graph_build.cpp): united by matchtigs_build_graph, in order

    // ---- graph ----
    PodVec<uint32_t> mirror;         // [V] (filled by host threads)
    // Per-node adjacency lists (newest edge first, the petgraph iteration order) cover the edges [0, linked_edges): the device
    // finish appends its dummy edges to the edge arrays only, and a host stage that walks adjacency calls ensure_linked() first.
    mutable std::vector<uint32_t> head_out;  // [V] newest outgoing edge among the linked ones
    mutable std::vector<uint32_t> out_deg;   // [V] over the linked edges; in_deg(n) == out_deg(mirror(n)) by the mirror property
    mutable std::vector<uint32_t> tail_out;  // [V] last edge of the list (only under the oldest-first adjacency policy, mtg_policy.h P3)
    PodVec<uint32_t> e_from, e_to;              // [E]
    mutable PodVec<uint32_t> e_next_out;        // [E]
    mutable uint64_t linked_edges = 0;
    mutable bool adjacency_ready = false;  // head_out / out_deg exist (sized by the first ensure_linked)
    mutable std::unique_ptr<std::mutex> link_mutex{new std::mutex};  // ensure_linked() (a pointer: the graph stays movable)
    // Edge payload (MatchtigEdgeData, implementation/mod.rs:287-317: sequence handle, forwards, weight, dummy id). Edges only ever
    // come as (edge, mirror edge) pairs 2b, 2b + 1 -- "biedge" b -- that share weight and dummy id; edge 2b is the forwards one;
    // original biedge b IS unitig b (clib.rs:239-248); a dummy has no unitig. So the payload is kept per BIEDGE, and only the part
    // that is not arithmetic: 8 bytes per biedge instead of the 25 bytes per EDGE of four arrays (4.3 -> 0.5 GB of first-touch
    // writes for the 2^27 graph: 0.14 of the 0.24 s a one-shot caller spent in mtg_graph_from_edges).
    PodVec<uint64_t> w_biedge;                  // [E / 2] weight (k-mers of the unitig; length of a dummy)
    PodVec<uint64_t> dummy_tail;                // [(E - n_original_edges) / 2] dummy id of the dummy biedges, in edge order
    uint64_t n_original_edges = 0;
    uint64_t weight(uint64_t e) const { return w_biedge[e >> 1]; }
    uint64_t dummy_id(uint64_t e) const { return e < n_original_edges ? 0 : dummy_tail[(e - n_original_edges) >> 1]; }  // 0 = original (implementation/mod.rs:291-293)
    uint64_t unitig(uint64_t e) const { return e < n_original_edges ? e >> 1 : 0; }
    bool forwards(uint64_t e) const { return !(e & 1); }
    // bookkeeping for the streaming cutter: ids >= first_breaking_edge are breaking dummies of weight breaking_weight
    uint64_t first_breaking_edge = UINT64_MAX;
    uint64_t breaking_weight = 0;
    bool dummies_canonical = true;
    mutable HugeArena arena;  // large scratch mappings of the host stages, kept between calls on this graph (huge_arena.hpp)
    mutable std::unique_ptr<DeviceEdgeCache, DeviceEdgeCacheDeleter> device_cache;

    uint64_t node_count() const { return mirror.size(); }
    uint64_t edge_count() const { return e_from.size(); }
    bool is_dummy(uint32_t e) const { return e >= n_original_edges; }  // (dummy ids are >= 1: greedytigs/mod.rs:681, implementation/mod.rs:573)
    bool self_mirror(uint32_t n) const { return mirror[n] == n; }
    uint32_t in_deg(uint32_t n) const { return out_deg[mirror[n]]; }

    void reserve_edges(uint64_t n);
    // n dummy biedges at once -- `out[i] -> in[i]` and its mirror `mirror(in[i]) -> mirror(out[i])`, ids e, e + 1 -- with the adjacency
    // lists as n one-by-one insertions would leave them (linked in parallel by node range unless !link)
    void add_biedges_bulk(const uint32_t *out, const uint32_t *in, const uint64_t *weight, uint64_t first_dummy_id, uint64_t n,
                          bool link = true);
    // Grows the edge arrays by n_new (uninitialised) entries WITHOUT linking them; the caller fills e_from, e_to, w_biedge and --
    // for dummy edges -- dummy_tail.
    void append_unlinked(uint64_t n_new);
    // Links the edges [linked_edges, E) into the adjacency lists, in ascending id (== one-by-one insertion order).
    void ensure_linked() const;
    void init_nodes(uint64_t n);
    void validate_pairing() const;
    // Pops all edges beyond the original ones (newest first), restoring head_out / out_deg.
    void reset_to_original();
};

// ---- construction (graph_build.cpp) ----
HostGraph *graph_from_edges(uint64_t n_nodes, const uint32_t *mirror, uint64_t n_edges, const uint32_t *from,
                            const uint32_t *to, const uint64_t *weight);
HostGraph *builder_new(uint64_t unitig_amount);
void builder_merge(HostGraph *g, uint64_t ua, bool sa, uint64_t ub, bool sb);
void builder_build(HostGraph *g, const uint64_t *unitig_weights);

// ---- host stages (host_pipeline.cpp) ----
std::vector<Pair> replay_claims(const HostGraph &g, uint64_t n_sources, const uint32_t *out_nodes,
                                const int32_t *multiplicity, const uint8_t *is_in_node, const uint64_t *cand_start,
                                const uint32_t *cand_count, const uint64_t *pool);
uint64_t insert_pair_edges(HostGraph &g, const Pair *pairs, uint64_t n_pairs);
uint64_t make_eulerian(HostGraph &g, uint64_t dummy_edge_id, uint64_t k);
bool is_eulerian(const HostGraph &g);
Walks euler_cycles(const HostGraph &g);
Walks cut_cycles(const HostGraph &g, const Walks &cycles, uint64_t k);
// matching_instance.cpp: optimal matchtigs around the external matcher (matchtigs/mod.rs:150-812)
struct MatchingInstance {
    uint64_t k = 0;
    uint64_t transformed_node_count = 0;  // GraphMatchingNodeMap::node_count()
    uint64_t wcc_amount = 0, matching_node_count = 0, matching_edge_count = 0;
    uint64_t mirror_biedges = 0, mirror_expanded_biedges = 0;
    std::vector<uint32_t> first_id, id_count;  // [V] node_id_map[n] = first_id[n] .. + id_count[n]
    // the edge map, sorted by (n1, n2): edges of n1 are [edge_begin[n1], edge_begin[n1 + 1])
    std::vector<uint64_t> edge_begin;                            // [transformed_node_count + 1]
    PodVec<uint32_t> edge_n2, edge_weight, edge_out, edge_target;  // value = (weight, out_node, target_node) of the last insert
    std::vector<uint64_t> extra_offset;                          // matching_node_extra_offset [transformed_node_count]
};
MatchingInstance *build_matching_instance(const HostGraph &g, uint64_t k, uint64_t n_sources, const uint32_t *out_nodes,
                                          const int32_t *multiplicity, const uint64_t *cand_start, const uint32_t *cand_count,
                                          const uint64_t *pool);
uint64_t write_matching_instance(const MatchingInstance &m, const char *path);  // returns bytes written
std::vector<Pair> read_matching_solution(const MatchingInstance &m, const char *path);
// bcalm2.cpp: bin.rs:902-912 input route, FASTA file output
struct UnitigStore {
    std::string data;           // concatenated ASCII sequences
    std::vector<uint64_t> off;  // count + 1 offsets
};
HostGraph *read_bcalm2(const char *path, uint64_t k, UnitigStore **store_out);
void write_file(const char *path, const char *data, uint64_t len, int compression_level);
// spell.cpp: bin.rs:466-606
uint64_t write_walks_fasta(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                           const char *seqs, const uint64_t *seq_off, char **out_buf);
// bin.rs:667-818 when gfa (header = the input file's header line, or null for "H\tKL:Z:{k}"), else bin.rs:466-606
uint64_t write_walks_text(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                          const char *seqs, const uint64_t *seq_off, bool gfa, const char *gfa_header, char **out_buf);
uint64_t write_duplication_bitvector(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges,
                                     char **out_buf);  // implementation/mod.rs:668-702
uint64_t flatten_clib(const HostGraph &g, const Walks &tigs, int64_t *edge_out, uint64_t *insert_out, uint64_t *limits);

}  // namespace mtg
