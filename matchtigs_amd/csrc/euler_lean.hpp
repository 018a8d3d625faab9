// euler_lean.hpp -- the reference-order Euler walk over 32-byte node records (euler_lean.cpp); records built on the GPU by
// finish_device.hip. No HIP types here: the walk is host C++.
#pragma once

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

// The latency-optimised walk of euler_fast.cpp (256-byte records with two levels of copied adjacency) seeded from the same
// GPU-built records: faster than euler_cycles_lean while 256 bytes per node fit the host.
Walks euler_cycles_from_lean(const LeanNode *lean, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                             const uint32_t *e_to, uint64_t E, HugeArena *arena);

}  // namespace mtg
