// graph_build.cpp -- building the host bigraph: from flat edge arrays, or through the reference's
// node-centric unitig-link interface (/root/reference/src/clib.rs:94-259).
#include "host_graph.hpp"

#include <chrono>

#include <algorithm>
#include <atomic>
#include <thread>

#include "../../include/mtg_policy.h"
#include "device.hpp"
#include "parallel.hpp"

namespace mtg {

void HostGraph::init_nodes(uint64_t n) {
    if (n >= NONE) MTG_DIE("graph has %llu nodes; node ids are 32-bit", (unsigned long long)n);
    mirror.resize(n);  // (uninitialised: graph_from_edges copies into it from host threads, builder_build fills it with NONE first)
    // (head_out / out_deg: sized by the first ensure_linked() -- the device path never walks host adjacency, and 8 bytes per node
    // of first-touch plus the linking itself were 0.4 of the 0.65 s a one-shot caller spent building the 2^27 graph)
    head_out.clear();
    out_deg.clear();
    adjacency_ready = false;
    linked_edges = 0;
}

void HostGraph::reserve_edges(uint64_t n) {
    e_from.reserve(n); e_to.reserve(n); e_next_out.reserve(n);
    w_biedge.reserve(n / 2 + 1);
}

// Links the edges [lo, hi) into the per-node adjacency lists in ascending id, i.e. exactly as one-by-one insertion would
// (newest first, like petgraph's per-node edge list). One pass partitions the edge ids by node range (a thread per edge range,
// one bucket per node range), then every node range prepends its edges, edge ranges in ascending order: O(E) reads instead of
// the O(threads x E) of letting every node-range thread scan all edges.
// One edge joins its from-node's list; edges arrive in ascending id. Iteration order of a list = policy P3 (mtg_policy.h): newest
// edge first (petgraph: the new edge becomes the head) or oldest first (the new edge goes to the tail: `tail_out` keeps it).
static inline void link_one(const HostGraph &g, uint32_t e) {
    const uint32_t f = g.e_from[e];
    if (!MTG_POLICY_ADJACENCY_OLDEST_FIRST) {
        g.e_next_out[e] = g.head_out[f];
        g.head_out[f] = e;
    } else {
        g.e_next_out[e] = NONE;
        if (g.head_out[f] == NONE) g.head_out[f] = e;
        else g.e_next_out[g.tail_out[f]] = e;
        g.tail_out[f] = e;
    }
    g.out_deg[f]++;
}
static void link_range(const HostGraph &g, uint64_t lo, uint64_t hi) {
    const uint64_t n = hi - lo, V = g.node_count();
    if (!n) return;
    unsigned T = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n < (1u << 16) || V < 1024) T = 1;
    if (T == 1) {
        for (uint64_t e = lo; e < hi; e++) link_one(g, (uint32_t)e);
        return;
    }
    unsigned shift = 0;
    while ((((V - 1) >> shift) + 1) > T) shift++;
    const unsigned NB = (unsigned)(((V - 1) >> shift) + 1);  // node ranges of 2^shift nodes
    const uint64_t chunk = (n + T - 1) / T;
    std::vector<std::vector<PodVec<uint32_t>>> bucket(T, std::vector<PodVec<uint32_t>>(NB));
    parallel_tasks(T, [&](uint64_t i) {
        const uint64_t a = std::min(hi, lo + i * chunk), b = std::min(hi, a + chunk);
        auto &mine = bucket[i];
        for (unsigned j = 0; j < NB; j++) mine[j].reserve((b - a) / NB + ((b - a) / NB) / 4 + 16);
        for (uint64_t e = a; e < b; e++) mine[g.e_from[e] >> shift].push_back((uint32_t)e);
    }, T);
    parallel_tasks(NB, [&](uint64_t j) {
        for (unsigned i = 0; i < T; i++)
            for (const uint32_t e : bucket[i][j]) link_one(g, e);
    }, T);
}

// (const, but it writes the adjacency lists the first time a read-only stage needs them after a device finish: two threads that
// call read-only functions on one graph meet at the lock; the fast path is one acquire load)
void HostGraph::ensure_linked() const {
    const uint64_t total = e_from.size();
    if (__atomic_load_n(&adjacency_ready, __ATOMIC_ACQUIRE) && __atomic_load_n(&linked_edges, __ATOMIC_ACQUIRE) >= total) return;
    std::lock_guard<std::mutex> lock(*link_mutex);
    if (!adjacency_ready) {  // first use of the host adjacency on this graph: the per-node heads come into being now
        head_out.assign(node_count(), NONE);
        if (MTG_POLICY_ADJACENCY_OLDEST_FIRST) tail_out.assign(node_count(), NONE);
        out_deg.assign(node_count(), 0);
        linked_edges = 0;
        __atomic_store_n(&adjacency_ready, true, __ATOMIC_RELEASE);
    }
    if (linked_edges >= total) return;
    link_range(*this, linked_edges, total);
    __atomic_store_n(&linked_edges, total, __ATOMIC_RELEASE);
}

void HostGraph::append_unlinked(uint64_t n_new) {
    const uint64_t total = e_from.size() + n_new;
    if (total >= NONE - 1) MTG_DIE("edge ids are 32-bit; too many edges");
    if ((e_from.size() | n_new) & 1) MTG_DIE("internal error: edges come in (edge, mirror edge) pairs");
    e_from.resize(total); e_to.resize(total); e_next_out.resize(total);
    w_biedge.resize(total / 2);
    if (built) dummy_tail.resize((total - n_original_edges) / 2);
}

// Appends n dummy biedges (out[i] -> in[i] with weight[i] and dummy id first_dummy_id + 1 + i, each followed by its
// mirror) exactly as n add_biedge calls would: the edge arrays are filled by host threads, the per-node adjacency lists
// are then linked in ascending edge id (link_range), so the newest-first iteration order is the same as with one-by-one insertion.
void HostGraph::add_biedges_bulk(const uint32_t *out, const uint32_t *in, const uint64_t *weight, uint64_t first_dummy_id,
                                 uint64_t n, bool link) {
    if (!n) return;
    const uint64_t base = e_from.size();
    append_unlinked(2 * n);
    parallel_ranges(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t e = base + 2 * i;
            e_from[e] = out[i]; e_to[e] = in[i];
            e_from[e + 1] = mirror[in[i]]; e_to[e + 1] = mirror[out[i]];
            w_biedge[e >> 1] = weight[i];
            dummy_tail[(e - n_original_edges) >> 1] = first_dummy_id + 1 + i;
        }
    });
    if (link) ensure_linked();
}

void HostGraph::reset_to_original() {
    first_breaking_edge = UINT64_MAX;
    breaking_weight = 0;
    dummies_canonical = true;
    const uint64_t total = e_from.size(), keep = n_original_edges;
    if (total == keep) return;
    // pop the linked dummy edges: the newest edge of a node unwinds the node's list down to its newest original edge
    // (exactly one edge per touched node is the head, so every node is handled by one thread)
    const uint64_t linked = std::min(linked_edges, total);
    if (MTG_POLICY_ADJACENCY_OLDEST_FIRST && linked > keep) {
        // (the dummy edges sit at the tails of the lists under this policy: the lists are simply linked again when next needed)
        adjacency_ready = false;
        linked_edges = 0;
    } else if (linked > keep)
        parallel_ranges(linked - keep, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t e = keep + lo; e < keep + hi; e++) {
                const uint32_t f = e_from[e];
                if (__atomic_load_n(&head_out[f], __ATOMIC_RELAXED) != e) continue;
                uint32_t x = (uint32_t)e, c = 0;
                while (x != NONE && x >= keep) {
                    x = e_next_out[x];
                    c++;
                }
                __atomic_store_n(&head_out[f], x, __ATOMIC_RELAXED);
                out_deg[f] -= c;
            }
        });
    linked_edges = std::min(linked_edges, keep);
    e_from.resize(keep); e_to.resize(keep); e_next_out.resize(keep);
    w_biedge.resize(keep / 2);
    dummy_tail.resize(0);
}

void HostGraph::validate_pairing() const {
    // graph.verify_node_pairing(), clib.rs:251
    const uint64_t n = mirror.size();
    parallel_ranges(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint32_t m = mirror[i];
            if (m == NONE || m >= n || mirror[m] != i)
                MTG_DIE("assertion failed: graph.verify_node_pairing() (node %llu)", (unsigned long long)i);
        }
    });
}

// The per-node adjacency lists over the original edges (insertion order, newest first, like petgraph's per-node edge list) are
// linked LAZILY, by the first host stage that walks them (ensure_linked): nothing on the device path does.
static void link_adjacency(HostGraph &g, uint64_t n_edges) {
    g.linked_edges = 0;
    (void)n_edges;
}

HostGraph *graph_from_edges(uint64_t n_nodes, const uint32_t *mirror, uint64_t n_edges, const uint32_t *from,
                            const uint32_t *to, const uint64_t *weight) {
    if ((n_nodes && !mirror) || (n_edges && (!from || !to || !weight))) MTG_DIE("mtg_graph_from_edges: null array");
    if (n_edges % 2) MTG_DIE("mtg_graph_from_edges: edges must come in (forward, mirror) pairs");
    if (n_edges >= NONE - 1) MTG_DIE("edge ids are 32-bit; too many edges");
    // (a helper thread starts the HIP runtime and reserves the device memory of the call that will follow, beside the work below)
    device_reserve_async(n_nodes, n_edges);
    HostGraph *g = new HostGraph();
    static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] graph_from_edges: %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    g->init_nodes(n_nodes);
    lap("init_nodes");
    // In the numbering most producers use, the mirror of node n is n ^ 1 (self-mirror nodes apart). One bit per node says where that
    // holds (written with the copy of the mirror array): the edge check below then needs a random bit of an 11-MB table
    // (cache-resident) instead of a random word of the 4 V-byte mirror array for all but the odd nodes -- the same comparison, half the
    // time on the bench graph (44-52 -> 24-31 ms). Any other numbering (clib.rs's own: representatives in slot order) takes the
    // lookups as before.
    PodVec<uint64_t> adjacent_v((n_nodes + 63) / 64 + 1);
    uint64_t *adjacent = adjacent_v.data();
    parallel_ranges((n_nodes + 63) / 64, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t w = lo; w < hi; w++) {
            uint64_t bits = 0;
            const uint64_t n0 = w * 64, n1 = std::min<uint64_t>(n_nodes, n0 + 64);
            for (uint64_t n = n0; n < n1; n++) {
                const uint32_t m = mirror[n];
                g->mirror[n] = m;
                bits |= (uint64_t)(m == (uint32_t)(n ^ 1u)) << (n - n0);
            }
            adjacent[w] = bits;
        }
    });
    lap("mirror");
    g->validate_pairing();
    lap("validate_pairing");
    // bulk form of n_edges / 2 add_biedge calls: arrays filled by host threads, adjacency linked by node range. Room for the dummy
    // edges the algorithms append later (matched pairs + breaking edges: 43 % of the original edges on the bench graph) is reserved
    // now -- address space only, untouched pages cost nothing --, so that the first insertion does not copy 5 GB of edge arrays.
    g->reserve_edges(n_edges + n_edges / 2 + 1024);
    g->e_from.resize(n_edges); g->e_to.resize(n_edges); g->e_next_out.resize(n_edges);
    g->w_biedge.resize(n_edges / 2);
    parallel_ranges(n_edges / 2, [&](uint64_t lo, uint64_t hi) {
        // (measured and dropped: MADV_POPULATE_WRITE on the thread's parts of the three fresh arrays before it fills them -- no faster
        // at 2^27, 20 % slower at 2^30 -- and streaming stores, see below)
        constexpr uint64_t AHEAD = 24;  // biedges: the two mirror[] lookups of the check below are random reads into 4 V bytes
        const uint32_t *mir = g->mirror.data();
        for (uint64_t u = lo; u < hi; u++) {
            const uint64_t e = 2 * u;
            if (u + AHEAD < hi) {
                const uint32_t pf = from[e + 2 * AHEAD], pt = to[e + 2 * AHEAD];
                if (pf < n_nodes && !((adjacent[pf >> 6] >> (pf & 63)) & 1u)) __builtin_prefetch(&mir[pf]);
                if (pt < n_nodes && !((adjacent[pt >> 6] >> (pt & 63)) & 1u)) __builtin_prefetch(&mir[pt]);
            }
            const uint32_t f = from[e], t = to[e];
            if (f >= n_nodes || t >= n_nodes) MTG_DIE("edge %llu: node id out of range", (unsigned long long)e);
            // graph.verify_edge_mirror_property(), clib.rs:252: the partner must be mirror(to) -> mirror(from)
            const uint32_t mirror_t = ((adjacent[t >> 6] >> (t & 63)) & 1u) ? (t ^ 1u) : mir[t];
            const uint32_t mirror_f = ((adjacent[f >> 6] >> (f & 63)) & 1u) ? (f ^ 1u) : mir[f];
            if (from[e + 1] != mirror_t || to[e + 1] != mirror_f || weight[e + 1] != weight[e])
                MTG_DIE("assertion failed: graph.verify_edge_mirror_property() (edge %llu)", (unsigned long long)e);
            // (plain stores: streaming stores into these freshly mapped arrays measured 1.5-2 x slower, 0.15-0.22 s against 0.07-0.13 s
            // alternating on one box -- the page faults of first touch do not mix with write-combining)
            g->e_from[e] = f; g->e_to[e] = t;
            g->e_from[e + 1] = from[e + 1]; g->e_to[e + 1] = to[e + 1];
            g->w_biedge[u] = weight[e];  // (edge 2u = unitig u forwards, edge 2u + 1 its mirror: host_graph.hpp)
        }
    });
    lap("edge arrays");
    link_adjacency(*g, n_edges);
    lap("link_adjacency");
    g->n_original_edges = n_edges;
    g->built = true;
    return g;
}

// ---------------------------------------------------------------------------------------------
// The clib.rs builder (clib.rs:94-259). matchtigs_merge_nodes is called once per link of the unitig graph -- hundreds of millions
// of times for a human genome -- and each call is two unions in a union-find over 4 slots per unitig: two to four dependent random
// accesses into gigabytes, ~90 ns per call when done on the spot (measured, 24 M links over 8 M unitigs). Since round 5 a call only
// RECORDS its link (8 bytes, after the same argument checks) and matchtigs_build_graph performs the unions -- in the order of the
// calls, so that every root is the one the reference's union-find (disjoint-sets 0.4.2: union by rank, ties by policy P4) ends
// with -- in one loop that knows the future: the slots of the links a few iterations ahead are prefetched while the current one
// is united, which hides part of the DRAM latency the per-call form had to wait for (137 M links over 48 M unitigs: 13 -> 6.6-8 s
// for all calls + the build; what remains is the chain of dependent loads and stores from one union to the next).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t LINK_CHUNK = 1ull << 24;  // links per chunk of the record (128 MB)

HostGraph *builder_new(uint64_t unitig_amount) {
    if (unitig_amount >= (1ull << 31)) MTG_DIE("matchtigs_initialise_graph: %llu unitigs: edge ids are 32-bit", (unsigned long long)unitig_amount);
    // (a helper thread starts the HIP runtime and reserves the device memory of the call that will follow, while the caller reports
    // its links: a compacted de Bruijn graph has about 1.4 nodes per unitig)
    device_reserve_async(unitig_amount + unitig_amount / 2, 2 * unitig_amount);
    HostGraph *g = new HostGraph();
    g->has_builder = true;
    g->unitig_amount = unitig_amount;
    return g;
}

void builder_merge(HostGraph *g, uint64_t ua, bool sa, uint64_t ub, bool sb) {
    if (!g || !g->has_builder || g->built) MTG_DIE("matchtigs_merge_nodes: graph is not in the building state");
    if (ua >= g->unitig_amount || ub >= g->unitig_amount) MTG_DIE("matchtigs_merge_nodes: unitig id out of range");
    if (g->link_chunks.empty() || g->link_chunks.back().size() == LINK_CHUNK) {
        g->link_chunks.emplace_back();
        g->link_chunks.back().reserve(LINK_CHUNK);
    }
    // (unitig ids are below 2^31: 32 bits hold an id and its strand)
    g->link_chunks.back().push_back(((ua << 1 | (sa ? 1u : 0u)) << 32) | (ub << 1 | (sb ? 1u : 0u)));
}

namespace {
// Union-find with the representative choice of disjoint-sets 0.4.2 (Cargo.lock:412-416): union by rank, ties
// by policy P4 (mtg_policy.h: the first argument's root goes below the second's). A root keeps its RANK in its own parent word
// (top bit set + rank) instead of pointing at itself, so that a union touches one array, not two -- a rank array would cost one more
// cache miss per union. Index = uint32_t while the 4 U slots fit below its top bit.
template <typename Index>
struct UnionFind {
    static constexpr Index ROOT = (Index)1 << (sizeof(Index) * 8 - 1);
    Index *parent;
    // Written without data-dependent branches for the chains that occur (a slot is a root, points at one, or at a slot that does):
    // whether a slot is a root is a coin flip to the branch predictor, and a mispredicted branch costs more here than the loads,
    // which the caller has prefetched. Full compression; does not change which element is the root.
    inline Index root(Index x) {
        const Index p0 = parent[x];
        const bool is0 = p0 & ROOT;
        const Index r1 = is0 ? x : p0;
        const Index p1 = parent[r1];
        const bool is1 = p1 & ROOT;
        Index r = is1 ? r1 : p1;
        Index p = parent[r];
        if (__builtin_expect(!(p & ROOT), 0)) {  // (deeper than two: rare after compression)
            do { r = p; p = parent[r]; } while (!(p & ROOT));
            Index y = r1;
            while (y != r) {
                const Index ny = parent[y];
                parent[y] = r;
                y = ny;
            }
        }
        parent[x] = is0 ? p0 : r;  // (a root keeps its own word; depth one rewrites the same value)
        return r;
    }
    inline void unite(Index x, Index y) {
        const Index a = root(x), b = root(y);
        const Index ra = parent[a], rb = parent[b];  // ROOT | rank
        // equal ranks: policy P4 decides who goes below (mtg_policy.h). Already united (a == b): both stores rewrite the root's word.
        const bool same = a == b;
        const bool a_stays = mtg_policy_union_tie_first_goes_below() ? ra > rb : ra >= rb;
        const Index top = a_stays ? a : b, below = a_stays ? b : a;
        const Index top_word = (a_stays ? ra : rb) + (ra == rb ? 1 : 0);
        parent[top] = same ? ra : top_word;
        parent[below] = same ? ra : top;
    }
};
// slots (clib.rs:104-122): fwd-in 4u, bwd-out 4u+1, fwd-out 4u+2, bwd-in 4u+3
struct LinkSlots { uint64_t out_a, in_b, mirror_in_a, mirror_out_b; };
inline LinkSlots slots_of(uint64_t link) {
    const uint64_t a = link >> 32, b = link & 0xFFFFFFFFull;
    const uint64_t ua = a >> 1, ub = b >> 1;
    const bool sa = a & 1u, sb = b & 1u;
    return LinkSlots{sa ? ua * 4 + 2 : ua * 4 + 1, sb ? ub * 4 : ub * 4 + 3, sa ? ua * 4 + 3 : ua * 4, sb ? ub * 4 + 1 : ub * 4 + 2};
}
}  // namespace

template <typename Index>
static void build_from_links(HostGraph *g, const uint64_t *unitig_weights);

void builder_build(HostGraph *g, const uint64_t *unitig_weights) {
    if (!g || !g->has_builder || g->built) MTG_DIE("matchtigs_build_graph: graph is not in the building state");
    if (!unitig_weights) MTG_DIE("assertion failed: !unitig_weights.is_null() (clib.rs:188)");
    if (g->unitig_amount * 4 < (1ull << 31)) build_from_links<uint32_t>(g, unitig_weights);
    else build_from_links<uint64_t>(g, unitig_weights);
}

template <typename Index>
static void build_from_links(HostGraph *g, const uint64_t *unitig_weights) {
    const uint64_t slots = g->unitig_amount * 4;
    const uint64_t U = g->unitig_amount, n_edges = 2 * U;
    if (n_edges >= NONE - 1) MTG_DIE("edge ids are 32-bit; too many edges");
    static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] build_graph: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    // (0) the unions, in the order of the matchtigs_merge_nodes calls (clib.rs:168-169: (out_a, in_b), then (mirror_in_a, mirror_out_b))
    PodVec<Index> parent_v(slots);
    Index *parent = parent_v.data();
    constexpr Index ROOT = UnionFind<Index>::ROOT;
    parallel_ranges(slots, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) parent[i] = ROOT;  // every slot a root of rank 0
    });
    {
        UnionFind<Index> uf{parent};
        // the slots' own words are requested AHEAD links before their union (a unitig's four slots share a cache line: two requests
        // per link). Measured and dropped: a second stage that requests the words those point at -- its four extra loads per link cost
        // more than the misses it hides (8 M unitigs: 1.0 s with it, 0.72 s without; 48 M: 6.2 against 5.3-6.3 s).
        constexpr uint64_t AHEAD = 32;
        for (size_t c = 0; c < g->link_chunks.size(); c++) {
            const PodVec<uint64_t> &chunk = g->link_chunks[c];
            const uint64_t n = chunk.size();
            for (uint64_t i = 0; i < n; i++) {
                if (i + AHEAD < n) {
                    const LinkSlots f = slots_of(chunk[i + AHEAD]);
                    __builtin_prefetch(&parent[f.out_a]);
                    __builtin_prefetch(&parent[f.in_b]);
                }
                const LinkSlots s = slots_of(chunk[i]);
                uf.unite((Index)s.out_a, (Index)s.in_b);
                uf.unite((Index)s.mirror_in_a, (Index)s.mirror_out_b);
            }
        }
        std::vector<PodVec<uint64_t>>().swap(g->link_chunks);
    }
    // (from here on a root points at itself, as the passes below expect)
    parallel_ranges(slots, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++)
            if (parent[i] & ROOT) parent[i] = (Index)i;
    });
    lap("unions");
    // Every pass over the 4 U slots below runs on host threads (until round 5 they were sequential loops of ~30 ns per slot: 8 s for
    // a human-sized unitig set, more than the whole computation that follows). The union-find is final here, so everything below only
    // reads it, except the compression of pass 1.
    // (1) full compression: every slot points at its root. A thread writes only its own slots; a chase may pass through a slot another
    // thread has just pointed at its root -- still an ancestor --, and roots never change.
    parallel_ranges(slots, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            Index r = (Index)i, p;
            while ((p = __atomic_load_n(&parent[r], __ATOMIC_RELAXED)) != r) r = p;
            __atomic_store_n(&parent[i], r, __ATOMIC_RELAXED);
        }
    });
    lap("compression");
    // (2) node id = rank of the representative among the sorted distinct representatives (clib.rs:193-234): roots counted per range,
    // ranges prefix-summed, roots numbered
    PodVec<uint32_t> node_of_root(slots);  // (only read at roots)
    uint64_t n_nodes = 0;
    {
        constexpr unsigned MAXR = 64;
        uint64_t range_lo[MAXR], range_cnt[MAXR];
        std::atomic<unsigned> n_ranges{0};
        parallel_ranges(slots, [&](uint64_t lo, uint64_t hi) {
            uint64_t c = 0;
            for (uint64_t i = lo; i < hi; i++) c += parent[i] == i;
            const unsigned id = n_ranges++;
            if (id >= MAXR) MTG_DIE("matchtigs_build_graph: too many host threads");
            range_lo[id] = lo;
            range_cnt[id] = c;
        });
        const unsigned R = n_ranges.load();
        unsigned order[MAXR];
        for (unsigned i = 0; i < R; i++) order[i] = i;
        std::sort(order, order + R, [&](unsigned x, unsigned y) { return range_lo[x] < range_lo[y]; });
        uint64_t base_of_lo[MAXR], lo_sorted[MAXR];
        for (unsigned i = 0; i < R; i++) {
            lo_sorted[i] = range_lo[order[i]];
            base_of_lo[i] = n_nodes;
            n_nodes += range_cnt[order[i]];
        }
        if (n_nodes >= NONE - 1) MTG_DIE("too many nodes for 32-bit ids");
        parallel_ranges(slots, [&](uint64_t lo, uint64_t hi) {  // (the same ranges: parallel_ranges cuts [0, n) the same way for the same n)
            unsigned k = 0;
            while (k < R && lo_sorted[k] != lo) k++;
            if (k == R) MTG_DIE("matchtigs_build_graph: internal error (range split changed)");
            uint64_t next = base_of_lo[k];
            for (uint64_t i = lo; i < hi; i++)
                if (parent[i] == i) node_of_root[i] = (uint32_t)next++;
        });
    }
    lap("node numbers");
    g->init_nodes(n_nodes);
    g->reserve_edges(n_edges + n_edges / 2 + 1024);  // (room for the dummy edges: see graph_from_edges)
    g->e_from.resize(n_edges); g->e_to.resize(n_edges); g->e_next_out.resize(n_edges);
    g->w_biedge.resize(n_edges / 2);
    parallel_ranges(n_nodes, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) g->mirror[i] = NONE;
    });
    lap("arrays");
    auto node_of = [&](uint64_t slot) { return node_of_root[parent[slot]]; };
    // (3) set_mirror_nodes (clib.rs:236-237) is an assignment in unitig order: a later unitig re-pairs a node an earlier one paired
    // differently (then verify_node_pairing fails below, as in the reference). For a consistent input every node only ever receives
    // one value, and the assignments commute: they run on host threads, each one an exchange that notices a node that had ANOTHER
    // value before -- then the pass is repeated in the reference's order, on one thread.
    {
        uint32_t *mir = g->mirror.data();
        std::atomic<bool> conflict{false};
        parallel_ranges(U, [&](uint64_t lo, uint64_t hi) {
            bool bad = false;
            auto put = [&](uint32_t x, uint32_t v) {
                const uint32_t old = __atomic_exchange_n(&mir[x], v, __ATOMIC_RELAXED);
                bad |= old != NONE && old != v;
            };
            for (uint64_t u = lo; u < hi; u++) {
                const uint32_t n1 = node_of(u * 4), n2 = node_of(u * 4 + 2), mirror_n2 = node_of(u * 4 + 3), mirror_n1 = node_of(u * 4 + 1);
                put(n1, mirror_n1); put(mirror_n1, n1);   // clib.rs:236
                put(n2, mirror_n2); put(mirror_n2, n2);   // clib.rs:237
            }
            if (bad) conflict.store(true);
        });
        if (conflict.load())
            for (uint64_t u = 0; u < U; u++) {
                const uint32_t n1 = node_of(u * 4), n2 = node_of(u * 4 + 2), mirror_n2 = node_of(u * 4 + 3), mirror_n1 = node_of(u * 4 + 1);
                mir[n1] = mirror_n1; mir[mirror_n1] = n1;
                mir[n2] = mirror_n2; mir[mirror_n2] = n2;
            }
    }
    lap("mirror nodes");
    parallel_ranges(U, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t u = lo; u < hi; u++) {
            const uint64_t e = 2 * u;
            g->e_from[e] = node_of(u * 4); g->e_to[e] = node_of(u * 4 + 2);              // clib.rs:239-243
            g->e_from[e + 1] = node_of(u * 4 + 3); g->e_to[e + 1] = node_of(u * 4 + 1);  // clib.rs:244-248
            g->w_biedge[u] = unitig_weights[u];
        }
    });
    lap("edges");
    link_adjacency(*g, n_edges);
    g->validate_pairing();  // clib.rs:251
    lap("verify_node_pairing");
    // clib.rs:252 verify_edge_mirror_property: with a valid pairing the two edges of a unitig are mirrors of
    // each other by construction only if later set_mirror_nodes calls did not re-pair their endpoints.
    parallel_ranges(U, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t u = lo; u < hi; u++) {
            const uint64_t e = 2 * u;
            if (g->e_from[e + 1] != g->mirror[g->e_to[e]] || g->e_to[e + 1] != g->mirror[g->e_from[e]])
                MTG_DIE("assertion failed: graph.verify_edge_mirror_property() (unitig %llu)", (unsigned long long)u);
        }
    });
    lap("verify_edge_mirror");
    g->n_original_edges = g->edge_count();
    g->built = true;
}

}  // namespace mtg
