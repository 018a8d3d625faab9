// finish_device.hpp -- internal interface between the device finishing stages (finish_device.hip, euler_device.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "euler_lean.hpp"
#include "hip_util.hpp"
#include "host_graph.hpp"

namespace mtg {

// euler_device.hip
// adj[row[v] + i] = i-th out-dart of v in ascending dart id (row: u32[V + 1]); pos[e] = slot of e in its bucket (may be null)
void device_build_buckets(hipStream_t st, const uint32_t *d_from, uint64_t E, uint64_t V, uint32_t *d_row, uint32_t *d_adj, uint32_t *d_pos,
                          uint32_t *d_scratch = nullptr);
// Euler bicycles of the Eulerian bigraph given by from[E] (mirror of dart e is e ^ 1) and mirror[V], all on the device:
// closed walks back to back in b_out (u32[E / 2]), their lengths / start offsets in b_clen / b_cbase (u32[*n_cycles]).
// (d_row0 / d_adj0 / E0: the kept buckets of the original darts [0, E0), or null: see device_build_buckets_merged)
void device_euler_decompose(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E, uint64_t V, hu::Buf &b_out,
                            hu::Buf &b_clen, hu::Buf &b_cbase, uint32_t *n_cycles, double *kernel_ms_out, const uint32_t *d_row0 = nullptr,
                            const uint32_t *d_adj0 = nullptr, uint64_t E0 = 0);
// the buckets of darts [0, E) = kept buckets of the original darts [0, E0) + fresh buckets of the dummy darts [E0, E)
void device_build_buckets_merged(hipStream_t st, const uint32_t *d_from, uint64_t E0, uint64_t E, uint64_t V, const uint32_t *d_row0,
                                 const uint32_t *d_adj0, uint32_t *d_row, uint32_t *d_adj);

}  // namespace mtg
