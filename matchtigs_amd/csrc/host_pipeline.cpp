// host_pipeline.cpp -- the sequential stages of the greedy-matchtigs path that follow the GPU SSSP stage.
//
//   replay_claims      greedytigs/mod.rs:301-523  (claim loop, 1-thread order) over GPU candidate lists
//   insert_pair_edges  greedytigs/mod.rs:678-689
//   make_eulerian      implementation/mod.rs:392-649 (+ choose_in_node_from_iterator :252-285)
//   euler_cycles       bigraph 5.0.1 compute_minimum_bidirected_eulerian_cycle_decomposition (call :722)
//   cut_cycles         greedytigs/mod.rs:726-789 == eulertigs/mod.rs:123-186
//   flatten_clib       clib.rs:393-407
//
// All citations are into /root/reference/src/. These are O(V+E) formulations that must produce the
// same sequences as the reference's (BTreeMap / Vec::rotate_left based) code; see DESIGN.md.
#include "../../include/mtg_policy.h"
#include "host_graph.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>

#include "parallel.hpp"

namespace mtg {

namespace {
struct StageTimer {  // MTG_DEBUG=1: per-stage host timings on stderr
    const bool on = std::getenv("MTG_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

// ---------------------------------------------------------------------------------------------
// Claim loop over precomputed candidate lists.
//
// The reference runs, per source o, `while demand > 0 { D = truncated Dijkstra against the live
// bitmap; claim from D }`. Because original weights are >= 1 and the graph does not change during
// this phase (dummy edges are added afterwards, :678), the answer of each query is "the first
// target_amount entries of L(o) that are still live", where L(o) (all *initial* in-nodes within
// k-1 of o, o excluded, sorted by (distance, node)) is what the GPU stage produced; live bits are
// only ever cleared (:456, :494, :499).
// ---------------------------------------------------------------------------------------------
std::vector<Pair> replay_claims(const HostGraph &g, uint64_t n_sources, const uint32_t *out_nodes,
                                const int32_t *multiplicity, const uint8_t *is_in_node, const uint64_t *cand_start,
                                const uint32_t *cand_count, const uint64_t *pool) {
    const uint64_t V = g.node_count();
    std::vector<int32_t> mult(multiplicity, multiplicity + V);
    std::vector<uint8_t> live(is_in_node, is_in_node + V);
    std::vector<Pair> result;
    std::vector<uint64_t> found(8);

    for (uint64_t i = 0; i < n_sources; i++) {
        const uint32_t o = out_nodes[i];
        const uint32_t om = g.mirror[o];
        const bool o_sm = (om == o);
        int32_t demand = mult[om];                                    // :306-311
        // :313-316 is a debug_assert (0..=4) only; a release build of the reference accepts any degree.
        if (demand == 0) continue;                                    // :318-320
        const uint64_t *list = pool + cand_start[i];
        const uint32_t len = cand_count[i];
        while (demand > 0) {                                          // :322
            const int target_amount = demand + 1;                     // :323
            if ((size_t)target_amount > found.size()) found.resize((size_t)target_amount);
            int nf = 0;
            for (uint32_t c = 0; c < len && nf < target_amount; c++)  // the query :324-335
                if (live[(uint32_t)list[c]]) found[nf++] = list[c];
            if (nf == 0) break;                                       // :338-346
            const bool abort_after_this = nf < target_amount;         // :348
            for (int c = 0; c < nf; c++) {                            // :350
                const uint32_t t = (uint32_t)found[c];
                const uint64_t dist = found[c] >> 32;
                bool self_edge = false;
                if (t == om) {                                        // :352-358
                    if (demand < 2) continue;
                    self_edge = true;
                }
                const uint32_t tm = g.mirror[t];
                const bool t_sm = (tm == t);
                demand = o_sm ? mult[o] : -mult[o];                   // :401-410
                if (demand == 0) break;                               // :412-414
                if (!self_edge && mult[t] == 0) {                     // :454-458
                    live[t] = 0;
                    continue;
                }
                result.push_back(Pair{o, t, dist});                   // :461
                const int red = self_edge ? 2 : 1;                    // :399
                if (o_sm) mult[o] -= 1;                               // :463-466
                else { mult[o] += red; mult[om] -= red; }             // :467-473
                demand = -mult[o];                                    // :474
                if (!self_edge) {                                     // :476-491
                    mult[t] -= 1;
                    if (!t_sm) mult[tm] += 1;
                }
                if (demand == 0) live[om] = 0;                        // :493-495
                if (!self_edge && mult[t] == 0) live[t] = 0;          // :497-501
            }
            if (abort_after_this) break;                              // :504-511
        }
    }
    return result;
}

// greedytigs/mod.rs:678-689
uint64_t insert_pair_edges(HostGraph &g, const Pair *pairs, uint64_t n_pairs) {
    StageTimer tm;
    if (n_pairs && g.first_breaking_edge != UINT64_MAX) g.dummies_canonical = false;  // matched edges after breaking edges
    PodVec<uint32_t> out(n_pairs), in(n_pairs);
    PodVec<uint64_t> w(n_pairs);
    parallel_ranges(n_pairs, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) { out[i] = pairs[i].out_node; in[i] = pairs[i].in_node; w[i] = pairs[i].distance; }
    });
    tm.lap("insert_pairs: unpack");
    g.add_biedges_bulk(out.data(), in.data(), w.data(), 0, n_pairs);  // dummy ids 1..n_pairs, :681
    tm.lap("insert_pairs: bulk add");
    return n_pairs;
}

bool is_eulerian(const HostGraph &g) {  // bigraph decomposes_into_eulerian_bicycles (call :708)
    g.ensure_linked();
    std::atomic<bool> ok{true};
    parallel_ranges(g.node_count(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++) {
            if (g.self_mirror((uint32_t)n)) {
                if (g.out_deg[n] & 1) ok = false;
            } else if (g.out_deg[n] != g.in_deg((uint32_t)n)) ok = false;
        }
    });
    return ok;
}

// ---------------------------------------------------------------------------------------------
// make_graph_eulerian_with_breaking_edges, implementation/mod.rs:392-649.
//
// The reference keeps two BTreeMaps (out-nodes keyed by Reverse(node), in-nodes keyed by node) that
// only lose entries after construction, always takes their first keys, and looks entries up by
// node. Dense formulation: need[n] < 0 for a present out-entry, > 0 for a present in-entry, 0 for
// "not in either map"; the maps' key sets are two static sorted node lists (built by host threads) and
// their first keys two monotone cursors into those lists that skip entries whose need has reached 0.
// ---------------------------------------------------------------------------------------------
uint64_t make_eulerian(HostGraph &g, uint64_t dummy_edge_id, uint64_t k) {
    StageTimer tm;
    g.ensure_linked();
    const uint64_t V = g.node_count();
    PodVec<int32_t> need(V);
    // find_non_eulerian_binodes_with_differences, :408: per node range, then concatenated in node order
    constexpr unsigned MAX_PARTS = 64;
    std::vector<uint32_t> part_sm[MAX_PARTS], part_in[MAX_PARTS], part_out[MAX_PARTS];
    std::atomic<unsigned> n_parts{0};
    std::vector<std::pair<uint64_t, unsigned>> part_of;  // (range begin, part id)
    std::mutex part_mutex;
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        const unsigned id = n_parts++;
        if (id >= MAX_PARTS) MTG_DIE("make_eulerian: too many host threads");
        {
            std::lock_guard<std::mutex> l(part_mutex);
            part_of.emplace_back(lo, id);
        }
        std::vector<uint32_t> sm, in, out;  // thread-local while growing (the shared headers would false-share)
        for (uint64_t n = lo; n < hi; n++) {
            int32_t d = 0;
            if (g.self_mirror((uint32_t)n)) {
                if (g.out_deg[n] & 1) sm.push_back((uint32_t)n);  // difference 0 entries, :424-427
            } else {
                d = (int32_t)((int64_t)g.out_deg[n] - (int64_t)g.in_deg((uint32_t)n));
                if (d > 0) in.push_back((uint32_t)n);
                else if (d < 0) out.push_back((uint32_t)n);
            }
            need[n] = d;
        }
        part_sm[id] = std::move(sm);
        part_in[id] = std::move(in);
        part_out[id] = std::move(out);
    });
    std::sort(part_of.begin(), part_of.end());
    std::vector<uint32_t> self_mirrors, in_list, out_list;  // ascending node order
    uint64_t total_in = 0;
    for (auto &po : part_of) {
        self_mirrors.insert(self_mirrors.end(), part_sm[po.second].begin(), part_sm[po.second].end());
        in_list.insert(in_list.end(), part_in[po.second].begin(), part_in[po.second].end());
        out_list.insert(out_list.end(), part_out[po.second].begin(), part_out[po.second].end());
    }
    for (uint32_t n : in_list) total_in += (uint64_t)need[n];
    tm.lap("make_eulerian: need[]");
    const uint64_t NI = in_list.size();
    uint64_t in_cur = 0;                         // index into in_list: smallest node with need > 0
    int64_t out_cur = (int64_t)out_list.size() - 1;  // index into out_list: largest node with need < 0
    auto first_in = [&](uint64_t from) -> uint64_t {
        while (from < NI && need[in_list[from]] <= 0) from++;
        return from;
    };
    if (g.first_breaking_edge == UINT64_MAX) {
        g.first_breaking_edge = g.edge_count();
        g.breaking_weight = k;
        for (uint64_t e = g.n_original_edges; e < g.edge_count(); e++)
            if (g.weight(e) >= k) g.dummies_canonical = false;  // a matched dummy as long as a breaking edge
    } else g.dummies_canonical = false;  // Eulerised twice
    // the pairing below only needs `need[]`; the breaking biedges are collected and appended in one bulk call
    std::vector<uint32_t> brk_out, brk_in;
    brk_out.reserve(total_in + self_mirrors.size());
    brk_in.reserve(total_in + self_mirrors.size());
    auto breaking = [&](uint32_t out_node, uint32_t in_node) {  // :489-493, :506-510, :572-577
        brk_out.push_back(out_node);
        brk_in.push_back(in_node);
    };

    for (size_t p = 0; p < self_mirrors.size(); p += 2) {  // :481-524
        if (p + 1 < self_mirrors.size()) {
            breaking(self_mirrors[p], self_mirrors[p + 1]);
        } else {
            in_cur = first_in(in_cur);
            if (in_cur >= NI)
                MTG_DIE("Have an uneven number of self-mirrors, but no other nodes with missing in edges. "
                        "(implementation/mod.rs:496-498)");
            const uint32_t in_node = in_list[in_cur];
            breaking(self_mirrors[p], in_node);
            const uint32_t m = g.mirror[in_node];
            if (need[m] >= 0) MTG_DIE("Mirror of in_node not found (implementation/mod.rs:517)");
            need[in_node] -= 1;  // :512
            need[m] += 1;        // removal at 0 (:513-517) / increment (:519-521) coincide in the dense form
        }
    }

    // (the loop is one thread chasing need[] / mirror[] of nodes it knows long in advance: both lists are walked monotonically,
    // so the entries a few positions ahead are fetched early)
    constexpr int64_t AHEAD = 12;
    for (;;) {  // :526
        while (out_cur >= 0 && need[out_list[out_cur]] >= 0) out_cur--;
        if (out_cur < 0) break;
        if (out_cur >= AHEAD) {
            const uint32_t po = out_list[out_cur - AHEAD];
            __builtin_prefetch(&need[po]);
            __builtin_prefetch(&g.mirror[po]);
        }
        if (out_cur >= AHEAD / 2) __builtin_prefetch(&need[g.mirror[out_list[out_cur - AHEAD / 2]]]);
        if (in_cur + (uint64_t)AHEAD < NI) {
            const uint32_t pi = in_list[in_cur + (uint64_t)AHEAD];
            __builtin_prefetch(&need[pi]);
            __builtin_prefetch(&g.mirror[pi]);
        }
        if (in_cur + (uint64_t)AHEAD / 2 < NI) __builtin_prefetch(&need[g.mirror[in_list[in_cur + (uint64_t)AHEAD / 2]]]);
        const uint32_t out_node = out_list[out_cur];
        const int32_t out_diff = need[out_node];
        // choose_in_node_from_iterator, :252-285
        in_cur = first_in(in_cur);
        if (in_cur >= NI) MTG_DIE("in_node_iterator.next().unwrap() on an empty map (implementation/mod.rs:262)");
        uint64_t pick = in_cur;
        if ((in_list[pick] == g.mirror[out_node] && out_diff > -2) || in_list[pick] == out_node) {
            pick = first_in(pick + 1);
            if (pick >= NI) MTG_DIE("No further in_nodes left (implementation/mod.rs:553)");
        }
        const uint32_t in_node = in_list[pick];
        const uint32_t mirror_out_node = g.mirror[in_node];  // :569
        const uint32_t mirror_in_node = g.mirror[out_node];  // :570
        breaking(out_node, in_node);
        need[out_node] += 1;  // :582 (entry disappears at 0, :590-594)
        need[in_node] -= 1;   // :583 (:600-602)
        if (need[mirror_out_node] < 0) need[mirror_out_node] += 1;  // :609-627, only if still present
        if (need[mirror_in_node] > 0) need[mirror_in_node] -= 1;    // :628-644
    }
    if (first_in(in_cur) < NI) MTG_DIE("in_node_differences not empty after Eulerisation (implementation/mod.rs:648)");
    tm.lap("make_eulerian: pairing");
    const std::vector<uint64_t> brk_w(brk_out.size(), k);
    g.add_biedges_bulk(brk_out.data(), brk_in.data(), brk_w.data(), dummy_edge_id, brk_out.size());  // ids continue, :573
    tm.lap("make_eulerian: bulk add");
    return dummy_edge_id + brk_out.size();
}

// ---------------------------------------------------------------------------------------------
// Euler bicycle decomposition (SURVEY App. A.2 policy), O(E):
//  * an edge and its mirror edge (e ^ 1) are consumed together;
//  * cycles start at the lowest unused edge id; the walk always takes the first unused out-edge in
//    adjacency order (newest first) and can only get stuck at its start node;
//  * the reference then scans the cycle vector from index 0 for the first edge whose from-node still
//    has an unused out-edge (policy P5 of mtg_policy.h; under its other setting: from the last index
//    backwards for the last such edge), rotate_left()s the vector to that index and appends the next closed
//    sub-walk. Here the cycle is a circular singly linked list of entries and the scan is a FIFO of
//    not-yet-exhausted entries: "rotate to x and append W" == "insert W just before x, x is the new
//    head"; inserting before x without a prev pointer is done by moving x's edge into a fresh entry y
//    placed after W (x's old entry receives W's first edge, which leaves the same node).
// ---------------------------------------------------------------------------------------------
// The latency-optimised formulation lives in euler_fast.cpp (euler_cycles); this is the simple O(E) one it falls
// back to for graphs whose node degrees do not fit its records.
Walks euler_cycles_generic(const HostGraph &g) {
    g.ensure_linked();
    const uint64_t E = g.edge_count();
    const uint64_t V = g.node_count();
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    std::vector<uint8_t> used(E / 2 + 1, 0);           // per biedge
    std::vector<uint32_t> cursor(g.head_out);          // first possibly-unused out-edge per node
    std::vector<uint32_t> ent_edge, ent_next;          // circular lists of the cycle under construction
    std::vector<uint32_t> fifo;                        // candidate entries in cycle order
    Walks out;
    out.edges.reserve(E / 2);
    (void)V;

    auto next_unused = [&](uint32_t node) -> uint32_t {
        uint32_t e = cursor[node];
        while (e != NONE && used[e >> 1]) e = g.e_next_out[e];
        cursor[node] = e;
        return e;
    };

    for (uint64_t e0 = 0; e0 < E; e0++) {
        if (used[e0 >> 1]) continue;
        ent_edge.clear(); ent_next.clear(); fifo.clear();
        size_t fifo_head = 0;
        uint32_t head = NONE;  // entry index of the cycle's first edge
        uint32_t start_edge = (uint32_t)e0;
        uint32_t splice_at = NONE;  // entry x before which the next closed walk is inserted (NONE: first walk)

        while (start_edge != NONE) {
            // ---- one closed walk W starting with start_edge ----
            const uint32_t start_node = g.e_from[start_edge];
            const size_t w_begin = ent_edge.size();
            uint32_t e = start_edge;
            for (;;) {
                used[e >> 1] = 1;
                ent_edge.push_back(e);
                ent_next.push_back((uint32_t)ent_edge.size());  // provisional: chain to the following entry
                const uint32_t node = g.e_to[e];
                e = next_unused(node);
                if (e == NONE) {
                    if (node != start_node)
                        MTG_DIE("Euler walk stuck at node %u != start node %u: graph is not Eulerian", node, start_node);
                    break;
                }
            }
            const size_t w_end = ent_edge.size();  // entries [w_begin, w_end)
            if (splice_at == NONE) {
                head = (uint32_t)w_begin;
                ent_next[w_end - 1] = head;
                for (size_t i = w_begin; i < w_end; i++) fifo.push_back((uint32_t)i);
            } else {
                // insert W before x = splice_at; x becomes y (a fresh entry) and stays the head.
                const uint32_t x = splice_at;
                const uint32_t y = (uint32_t)ent_edge.size();
                ent_edge.push_back(ent_edge[x]);
                ent_next.push_back(ent_next[x]);  // if x was the only entry this is x itself: y -> x(W1)
                ent_edge[x] = ent_edge[w_begin];  // x now carries W's first edge (same from-node)
                if (w_end - w_begin == 1) {
                    ent_next[x] = y;
                } else {
                    ent_next[x] = (uint32_t)(w_begin + 1);
                    ent_next[w_end - 1] = y;
                }
                // entry w_begin itself is now unused (its edge lives in x)
                head = y;
                if (!mtg_policy_euler_splice_last()) fifo[fifo_head] = y;  // x was at the front of the FIFO; y takes its place
                // (policy P5's other setting: x was popped off the stack, and y needs no entry -- it leaves the same node as x's new edge)
                fifo.push_back(x);
                for (size_t i = w_begin + 1; i < w_end; i++) fifo.push_back((uint32_t)i);
            }
            // ---- find the next start edge: policy P5 (mtg_policy.h) -- the FIRST entry in cycle order whose from-node has an unused
            // out-edge (the candidates are a queue), or the LAST one (they are a stack) ----
            start_edge = NONE;
            if (mtg_policy_euler_splice_last()) {
                while (!fifo.empty()) {
                    const uint32_t ent = fifo.back();
                    fifo.pop_back();
                    const uint32_t cand = next_unused(g.e_from[ent_edge[ent]]);
                    if (cand != NONE) { start_edge = cand; splice_at = ent; break; }
                }
            } else
            while (fifo_head < fifo.size()) {
                const uint32_t ent = fifo[fifo_head];
                const uint32_t cand = next_unused(g.e_from[ent_edge[ent]]);
                if (cand != NONE) { start_edge = cand; splice_at = ent; break; }
                fifo_head++;
            }
        }
        // ---- emit the cycle starting at head ----
        uint32_t ent = head;
        do {
            out.edges.push_back(ent_edge[ent]);
            ent = ent_next[ent];
        } while (ent != head);
        out.limits.push_back(out.edges.size());
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// rotate + cut, greedytigs/mod.rs:726-789 (== eulertigs/mod.rs:123-186), without moving data.
// ---------------------------------------------------------------------------------------------
// Generic form: follows the reference line by line (weights looked up per dummy edge).
static Walks cut_cycles_generic(const HostGraph &g, const Walks &cycles, uint64_t k) {
    Walks tigs;
    tigs.edges.reserve(cycles.edges.size());
    const uint64_t n_orig = g.n_original_edges;  // dummy edges are exactly the edges appended after the original ones
    uint64_t begin = 0;
    for (size_t c = 0; c < cycles.limits.size(); c++) {
        const uint32_t *cyc = cycles.edges.data() + begin;
        const uint64_t len = cycles.limits[c] - begin;
        begin = cycles.limits[c];
        uint64_t longest_w = 0, rot = 0;                         // :737-745
        for (uint64_t i = 0; i < len; i++) {
            const uint32_t e = cyc[i];
            if (e >= n_orig && g.weight(e) > longest_w) { longest_w = g.weight(e); rot = i; }
        }
        if (longest_w == 0) rot = 0;                             // :746-748
        auto at = [&](uint64_t i) -> uint32_t { uint64_t j = i + rot; return cyc[j >= len ? j - len : j]; };
        auto emit = [&](uint64_t from, uint64_t to) {            // rotated indices [from, to)
            for (uint64_t i = from; i < to; i++) tigs.edges.push_back(at(i));
            tigs.limits.push_back(tigs.edges.size());
        };
        uint64_t offset = 0;                                     // :750
        for (uint64_t i = 0; i < len; i++) {                     // :752
            const uint32_t e = at(i);
            const bool dummy = e >= n_orig;
            if (dummy && (g.weight(e) >= k || i == 0)) {       // :767-769
                if (offset < i) emit(offset, i);                 // :770-771
                offset = i + 1;                                  // :775
            }
        }
        if (offset < len) {                                      // :779-788
            if (at(len - 1) < n_orig) emit(offset, len);
            else if (offset < len - 1) emit(offset, len - 1);
        }
    }
    return tigs;
}

// Streaming form for graphs produced by insert_pair_edges + make_eulerian (the only producers on the path): matched
// dummies (weight <= k-1) occupy edge ids [n_original, first_breaking), breaking dummies (weight == k) the ids from
// first_breaking on, so "dummy" and "weight >= k" are id comparisons and no per-edge lookup is needed. A cycle's
// rotation point (first strictly-longest dummy, :737-748) is its first breaking edge if it has one.
Walks cut_cycles(const HostGraph &g, const Walks &cycles, uint64_t k) {
    if (!g.dummies_canonical || g.first_breaking_edge == UINT64_MAX || g.breaking_weight != k)
        return cut_cycles_generic(g, cycles, k);
    Walks tigs;
    tigs.edges.resize(cycles.edges.size());
    uint32_t *out = tigs.edges.data();
    uint64_t n_out = 0;
    const uint64_t n_orig = g.n_original_edges, first_brk = g.first_breaking_edge;
    uint64_t begin = 0;
    for (size_t c = 0; c < cycles.limits.size(); c++) {
        const uint32_t *cyc = cycles.edges.data() + begin;
        const uint64_t len = cycles.limits[c] - begin;
        begin = cycles.limits[c];
        uint64_t rot = len;
        for (uint64_t i = 0; i < len; i++)
            if (cyc[i] >= first_brk) { rot = i; break; }
        if (rot == len) {  // no breaking edge: rotate to the first strictly-longest matched dummy, if any
            uint64_t longest_w = 0;
            rot = 0;
            for (uint64_t i = 0; i < len; i++)
                if (cyc[i] >= n_orig && g.weight(cyc[i]) > longest_w) { longest_w = g.weight(cyc[i]); rot = i; }
        }
        if (len >= (1u << 20)) {
            // long cycle: the same cut, by host threads over chunks of the rotated index j (edge = cyc[(rot + j) % len]):
            // cut(j) = breaking edge, or any dummy at j == 0 (:768); a tig ends at a kept j whose successor is cut or absent.
            auto edge_at = [&](uint64_t j) { return cyc[j + rot < len ? j + rot : j + rot - len]; };
            auto is_cut = [&](uint64_t j, uint32_t e) { return e >= first_brk || (j == 0 && e >= n_orig); };
            constexpr unsigned MAXP = 64;
            uint64_t part_lo[MAXP], part_kept[MAXP], part_ends[MAXP];
            std::atomic<unsigned> n_parts{0};
            parallel_ranges(len, [&](uint64_t lo, uint64_t hi) {
                const unsigned id = n_parts++;
                uint64_t kept = 0, ends = 0;
                uint32_t e = edge_at(lo);
                for (uint64_t j = lo; j < hi; j++) {
                    const uint32_t nx = j + 1 < len ? edge_at(j + 1) : 0;
                    if (!is_cut(j, e)) {
                        kept++;
                        if (j + 1 == len || is_cut(j + 1, nx)) ends++;
                    }
                    e = nx;
                }
                part_lo[id] = lo; part_kept[id] = kept; part_ends[id] = ends;
            });
            const unsigned P = n_parts;
            unsigned order[MAXP];
            for (unsigned i = 0; i < P; i++) order[i] = i;
            std::sort(order, order + P, [&](unsigned a, unsigned b) { return part_lo[a] < part_lo[b]; });
            uint64_t kept_off[MAXP], ends_off[MAXP], kept_total = 0, ends_total = 0;
            for (unsigned i = 0; i < P; i++) {
                kept_off[order[i]] = kept_total; ends_off[order[i]] = ends_total;
                kept_total += part_kept[order[i]]; ends_total += part_ends[order[i]];
            }
            const uint64_t lim_base = tigs.limits.size();
            tigs.limits.resize(lim_base + ends_total);
            uint64_t *lim = tigs.limits.data() + lim_base;
            const uint64_t out_base = n_out;
            std::atomic<unsigned> n_parts2{0};
            parallel_ranges(len, [&](uint64_t lo, uint64_t hi) {
                n_parts2++;
                unsigned id = 0;
                while (part_lo[id] != lo) id++;
                uint64_t w = out_base + kept_off[id], l = ends_off[id];
                uint32_t e = edge_at(lo);
                for (uint64_t j = lo; j < hi; j++) {
                    const uint32_t nx = j + 1 < len ? edge_at(j + 1) : 0;
                    if (!is_cut(j, e)) {
                        out[w++] = e;
                        if (j + 1 == len || is_cut(j + 1, nx)) lim[l++] = w;
                    }
                    e = nx;
                }
            });
            n_out = out_base + kept_total;
            // tail (:779-788): a trailing (matched) dummy is dropped
            if (kept_total && !is_cut(len - 1, edge_at(len - 1)) && out[n_out - 1] >= n_orig) {
                n_out--;
                const uint64_t prev = tigs.limits.size() >= 2 && ends_total >= 2 ? tigs.limits[tigs.limits.size() - 2] : out_base;
                if (n_out > prev) tigs.limits.back() = n_out;
                else tigs.limits.pop_back();
            }
            continue;
        }
        uint64_t tig_begin = n_out;  // start of the tig being assembled
        auto close_tig = [&]() {
            if (n_out > tig_begin) tigs.limits.push_back(n_out);
            tig_begin = n_out;
        };
        bool first = true;  // rotated index 0: a dummy there is a cut point whatever its weight (:768)
        auto segment = [&](const uint32_t *p, uint64_t n) {
            for (uint64_t i = 0; i < n; i++) {
                const uint32_t e = p[i];
                if (e >= first_brk || (first && e >= n_orig)) close_tig();  // cut: drop the edge
                else out[n_out++] = e;
                first = false;
            }
        };
        segment(cyc + rot, len - rot);
        segment(cyc, rot);
        // tail (:779-788): a trailing (matched) dummy is dropped
        if (n_out > tig_begin && out[n_out - 1] >= n_orig) n_out--;
        close_tig();
    }
    tigs.edges.resize(n_out);
    return tigs;
}

// clib.rs:393-407
uint64_t flatten_clib(const HostGraph &g, const Walks &tigs, int64_t *edge_out, uint64_t *insert_out,
                      uint64_t *limits) {
    // clib.rs:393-407, per walk edge; position j of the flat arrays only depends on edge j (host threads over ranges of j)
    const uint64_t n_edges = tigs.limits.empty() ? 0 : tigs.limits.back();
    parallel_ranges(n_edges, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t j = lo; j < hi; j++) {
            const uint32_t e = tigs.edges[j];
            edge_out[j] = (int64_t)g.unitig(e) * (g.forwards(e) ? 1 : -1);
            insert_out[j] = g.is_dummy(e) ? g.weight(e) : 0;
        }
    });
    parallel_ranges(tigs.limits.size(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) limits[i] = tigs.limits[i];
    });
    return tigs.limits.size();
}

}  // namespace mtg
