// euler_splice_last.hpp -- the reference-order Euler walk under the OTHER setting of policy P5 (include/mtg_policy.h:
// MTG_POLICY_EULER_SPLICE_LAST = 1): once the walk is stuck at its start node it resumes at the LAST position of the cycle whose
// from-node still has an unused out-edge (the default scans for the first). Same common rules as euler_fast.cpp / euler_lean.cpp --
// an edge and its mirror (e ^ 1) are used together, cycles start at the lowest unused edge id, the first unused out-edge in
// adjacency order is taken -- over the same node records (any record type with LeanNode's header).
//
// "Rotate to x and append W" is "insert W before x, x is the new head" on a circular list of entries, exactly as in the default
// form; only the order in which candidates x are examined differs: last to first = a STACK of entries in cycle order. After a
// splice at x (popped) the closed walk's entries W_0 .. W_m are pushed; x's old edge needs no entry on the stack, it leaves the same
// node as W_0, which is examined after all of W_1 .. W_m and before anything older.
//
// A plain O(E) formulation: the latency-optimised machinery of the default setting (copied adjacency levels, parallel scans of the
// candidate queue) is tuned to the default's access pattern; this setting exists so that the parity suite can hold product ==
// oracle == restatement under both settings of P5 (tests/test_fuzz_small.py), not to be fast.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "host_graph.hpp"

namespace mtg {

template <typename Rec>
Walks euler_walk_splice_last(Rec *nodes, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from, const uint32_t *e_to, uint64_t E) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    std::vector<uint64_t> used((E / 2 + 63) / 64 + 1, 0);
    auto is_used = [&](uint32_t e) -> bool { return (used[(e >> 1) >> 6] >> ((e >> 1) & 63)) & 1ull; };
    auto set_used = [&](uint32_t e) { used[(e >> 1) >> 6] |= 1ull << ((e >> 1) & 63); };
    auto next_unused = [&](uint32_t node, uint32_t &to_out) -> uint32_t {  // first unused out-edge in iteration order (P3: the records' order)
        Rec &r = nodes[node];
        while (r.pos < r.deg) {
            const uint32_t e = r.pos < 3 ? r.eid[r.pos] : ext_eid[r.ext_begin + r.pos - 3];
            if (!is_used(e)) {
                to_out = r.pos < 3 ? r.to[r.pos] : ext_to[r.ext_begin + r.pos - 3];
                return e;
            }
            r.pos++;
        }
        return NONE;
    };
    std::vector<uint32_t> ent_edge, ent_next, ent_node, stack;
    Walks out;
    out.edges.reserve(E / 2);
    for (uint64_t e0 = 0; e0 < E; e0++) {
        if (is_used((uint32_t)e0)) continue;
        ent_edge.clear(); ent_next.clear(); ent_node.clear(); stack.clear();
        uint32_t head = NONE, splice_at = NONE;
        uint32_t start_edge = (uint32_t)e0, start_to = e_to[e0], start_node = e_from[e0];
        while (start_edge != NONE) {
            const size_t w_begin = ent_edge.size();
            uint32_t e = start_edge, from = start_node, to = start_to;
            for (;;) {  // one closed walk W
                set_used(e);
                ent_edge.push_back(e);
                ent_node.push_back(from);
                ent_next.push_back((uint32_t)ent_edge.size());
                from = to;
                e = next_unused(from, to);
                if (e == NONE) {
                    if (from != start_node) MTG_DIE("Euler walk stuck at node %u != start node %u: graph is not Eulerian", from, start_node);
                    break;
                }
            }
            const size_t w_end = ent_edge.size();
            if (splice_at == NONE) {
                head = (uint32_t)w_begin;
                ent_next[w_end - 1] = head;
                for (size_t i = w_begin; i < w_end; i++) stack.push_back((uint32_t)i);
            } else {  // insert W before x: x's edge moves to a fresh entry y behind W, x receives W's first edge
                const uint32_t x = splice_at, y = (uint32_t)ent_edge.size();
                ent_edge.push_back(ent_edge[x]);
                ent_node.push_back(ent_node[x]);
                ent_next.push_back(ent_next[x]);
                ent_edge[x] = ent_edge[w_begin];
                if (w_end - w_begin == 1) ent_next[x] = y;
                else { ent_next[x] = (uint32_t)(w_begin + 1); ent_next[w_end - 1] = y; }
                head = y;
                stack.push_back(x);
                for (size_t i = w_begin + 1; i < w_end; i++) stack.push_back((uint32_t)i);
            }
            start_edge = NONE;
            while (!stack.empty()) {  // the LAST entry in cycle order whose from-node still has an unused out-edge
                const uint32_t ent = stack.back();
                stack.pop_back();
                uint32_t to2 = NONE;
                const uint32_t cand = next_unused(ent_node[ent], to2);
                if (cand != NONE) { start_edge = cand; start_to = to2; start_node = ent_node[ent]; splice_at = ent; break; }
            }
        }
        uint32_t ent = head;
        do {
            out.edges.push_back(ent_edge[ent]);
            ent = ent_next[ent];
        } while (ent != head);
        out.limits.push_back(out.edges.size());
    }
    return out;
}

}  // namespace mtg
