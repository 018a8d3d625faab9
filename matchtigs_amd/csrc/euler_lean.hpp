// euler_lean.hpp -- the reference-order Euler walk over 32-byte node records (euler_lean.cpp); records built on the GPU by
// finish_device.hip. No HIP types here: the walk is host C++.
#pragma once

#include <atomic>
#include <cstdint>

#include "host_graph.hpp"

namespace mtg {

struct LeanNode {
    uint32_t eid[3];  // own adjacency positions 0..2, newest edge first (NONE beyond the degree)
    uint32_t to[3];
    uint16_t deg;     // min(out-degree, 65535); the walk refuses larger degrees before it starts
    uint16_t pos;     // positions < pos are known to be used
    uint32_t ext_begin;  // spill entries for own positions 3..deg-1
};
static_assert(sizeof(LeanNode) == 32, "LeanNode must be 32 bytes");
// Closed walks in the reference's order (same sequences as euler_cycles / euler_cycles_generic). nodes[V] is consumed (the
// cursors move); from / to are the edge arrays of the Eulerised graph (E entries).
Walks euler_cycles_lean(LeanNode *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                        const uint32_t *e_to, uint64_t E, HugeArena *arena);

// Copied adjacency is in the owner's iteration order (newest edge first). A copy holds the first `cnt` positions of its
// node and a `more` bit (the node has further positions): if none of the copied edges is unused and `more` is clear the
// node is exhausted, with `more` set the walk falls back to the node's own record.
struct alignas(256) EulerNode3 {
    static constexpr int LEVELS = 3;
    uint32_t eid[3];             // own adjacency positions 0..2
    uint32_t to[3];
    uint16_t deg;
    uint16_t pos;                // positions < pos are known to be used
    uint16_t sub_info;           // 3 bits per inline edge j: cnt (0..3) | more << 2
    uint16_t pad;
    uint32_t sub2_info;          // 3 bits per (j, q): cnt (0..2) | more << 2
    uint32_t ext_begin;          // spill entries for own positions 3..deg-1
    uint32_t sub_eid[3][3];      // adjacency of to[j]
    uint32_t sub_to[3][3];
    uint32_t sub2_eid[3][3][2];  // adjacency of sub_to[j][q]
    uint32_t sub2_to[3][3][2];
    uint32_t sub_cnt(uint32_t j) const { return (sub_info >> (3 * j)) & 3u; }
    bool sub_more(uint32_t j) const { return (sub_info >> (3 * j + 2)) & 1u; }
    uint32_t sub2_cnt(uint32_t j, uint32_t q) const { return (sub2_info >> (3 * (3 * j + q))) & 3u; }
    bool sub2_more(uint32_t j, uint32_t q) const { return (sub2_info >> (3 * (3 * j + q) + 2)) & 1u; }
};
static_assert(sizeof(EulerNode3) == 256, "EulerNode3 must be 256 bytes");

// The same record without the third level: 128 bytes, one record read per ~1.9 steps (the 256-byte one: 2.6; the 32-byte one: 1).
// For graphs whose 256-byte records would not fit the host comfortably (BASELINE configs[3] at its nominal size: 716 M nodes =
// 92 GB here against 183 GB).
struct alignas(128) EulerNode2 {
    static constexpr int LEVELS = 2;
    uint32_t eid[3];             // own adjacency positions 0..2
    uint32_t to[3];
    uint16_t deg;
    uint16_t pos;                // positions < pos are known to be used
    uint16_t sub_info;           // 3 bits per inline edge j: cnt (0..3) | more << 2
    uint16_t pad;
    uint32_t sub2_info;          // (unused: keeps the two formats' headers alike)
    uint32_t ext_begin;          // spill entries for own positions 3..deg-1
    uint32_t sub_eid[3][3];      // adjacency of to[j]
    uint32_t sub_to[3][3];
    uint32_t pad2[4];
    uint32_t sub_cnt(uint32_t j) const { return (sub_info >> (3 * j)) & 3u; }
    bool sub_more(uint32_t j) const { return (sub_info >> (3 * j + 2)) & 1u; }
    uint32_t sub2_cnt(uint32_t, uint32_t) const { return 0; }
    bool sub2_more(uint32_t, uint32_t) const { return false; }
};
static_assert(sizeof(EulerNode2) == 128, "EulerNode2 must be 128 bytes");

// The latency-optimised walk of euler_fast.cpp (256-byte records with two levels of copied adjacency) seeded from the same
// GPU-built records: faster than euler_cycles_lean while 256 bytes per node fit the host.
// (nodes[V]: the caller's buffer for the 256-byte records, filled here by host threads)
Walks euler_cycles_from_lean(const LeanNode *lean, EulerNode3 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                             const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena);

// The same walk over complete 256-byte records (all three levels filled, e.g. by the GPU: finish_device.hip); nodes[V] is consumed.
Walks euler_cycles_from_wide(EulerNode3 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                             const uint32_t *e_to, uint64_t E, HugeArena *arena);

// The same walk started while the 256-byte records still arrive from the GPU, in node order: the records of nodes >= *arrived are
// not there yet (the caller's transfer raises *arrived with release order after every completed slice, up to V) -- a step that
// needs one takes the node's 32-byte record in `lean` instead (own adjacency, one step per record). Same sequences; lean[V] is
// consumed as well (its cursors move).
Walks euler_cycles_from_wide_arriving(EulerNode3 *nodes, LeanNode *lean, const std::atomic<uint64_t> *arrived, uint64_t V, const uint32_t *ext_eid,
                                      const uint32_t *ext_to, const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena);

// (the same for the 128-byte records)
Walks euler_cycles_from_mid_arriving(EulerNode2 *nodes, LeanNode *lean, const std::atomic<uint64_t> *arrived, uint64_t V, const uint32_t *ext_eid,
                                     const uint32_t *ext_to, const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena);

// The walk over 128-byte records: seeded from the 32-byte ones (level two filled by host threads) / complete (e.g. from the GPU).
Walks euler_cycles_from_lean_mid(const LeanNode *lean, EulerNode2 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                                 const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena);
Walks euler_cycles_from_mid(EulerNode2 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                            const uint32_t *e_to, uint64_t E, HugeArena *arena);

}  // namespace mtg
