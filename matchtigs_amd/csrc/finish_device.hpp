// finish_device.hpp -- internal interface between the device finishing stages (finish_device.hip, euler_device.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "euler_lean.hpp"
#include "hip_util.hpp"
#include "host_graph.hpp"

namespace mtg {

// The Euleriser's regular steps, as the bucket build needs them (finish_device.hip -> euler_device.hip): step s < n_steps added the
// breaking darts zip_first + 2 s (leaves a_node[s]) and zip_first + 2 s + 1 (leaves mirror(b_node[s + delta])), and both unit
// orders are runs per node (a_node: out-nodes descending, b_node: in-nodes ascending), so the breaking out-darts of a node are two
// arithmetic id ranges given the counters and their prefixes -- no atomics, no sort for five sixths of the dummy darts.
struct ZipBuckets {
    uint64_t zip_first;   // dart of step 0
    uint32_t n_steps;     // regular steps (the parallel prefix of the Euleriser)
    uint32_t delta, n_units;
    const uint32_t *cin, *cout, *p_in, *p_out;  // [V] missing in- / out-edges per node and their exclusive prefixes
};

// euler_device.hip
// adj[row[v] + i] = i-th out-dart of v in ascending dart id (row: u32[V + 1]); pos[e] = slot of e in its bucket (may be null)
void device_build_buckets(hipStream_t st, const uint32_t *d_from, uint64_t E, uint64_t V, uint32_t *d_row, uint32_t *d_adj, uint32_t *d_pos,
                          uint32_t *d_scratch = nullptr);
// Euler bicycles of the Eulerian bigraph given by from[E] (mirror of dart e is e ^ 1) and mirror[V], all on the device:
// closed walks back to back in b_out (u32[E / 2]), their lengths / start offsets in b_clen / b_cbase (u32[*n_cycles]).
// (d_row0 / d_adj0 / E0: the kept buckets of the original darts [0, E0), or null: see device_build_buckets_merged)
void device_euler_decompose(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E, uint64_t V, hu::Buf &b_out,
                            hu::Buf &b_clen, hu::Buf &b_cbase, uint32_t *n_cycles, double *kernel_ms_out, const uint32_t *d_row0 = nullptr,
                            const uint32_t *d_adj0 = nullptr, uint64_t E0 = 0, const ZipBuckets *zip = nullptr);
// the buckets of darts [0, E) = kept buckets of the original darts [0, E0) + fresh buckets of the dummy darts [E0, E)
// (zip: the part of the dummy darts whose buckets are arithmetic, or null)
void device_build_buckets_merged(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E0, uint64_t E, uint64_t V,
                                 const uint32_t *d_row0, const uint32_t *d_adj0, uint32_t *d_row, uint32_t *d_adj, const ZipBuckets *zip = nullptr);

// the pairing of step 2 alone (succ[E]; *d_error bit 0 = a node is not balanced)
void device_pairing(hipStream_t st, const uint32_t *d_mirror, uint64_t V, const uint32_t *d_row, const uint32_t *d_adj, uint32_t *d_succ, uint32_t *d_error);

// cut_first_device.hip: the tigs straight from the pairing (no closed walks). false = not applicable, the caller decomposes instead.
struct CutFirstStats {
    uint64_t stretches = 0;            // breaking darts = walkers
    uint64_t breaking_free_darts = 0;  // darts on trails without a breaking dart (spliced in, or the reason for a `false`)
    uint64_t spliced_trails = 0, cyclic_tigs = 0;
};
bool device_cut_first(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E, uint64_t V, uint64_t E0, uint64_t first_brk,
                      const uint32_t *d_pw, const uint32_t *d_row0, const uint32_t *d_adj0, const ZipBuckets *zip, hu::Buf &b_te, hu::Buf &b_tl,
                      uint64_t *n_kept_out, uint64_t *n_tigs_out, CutFirstStats *stats);

}  // namespace mtg
