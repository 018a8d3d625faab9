// device.hpp -- interface of the HIP device stage (device_build.hip, device_classify.hip, device_sssp.hip, device_replay.hip, device_pairs.hip) towards the C-ABI layer.
#pragma once

#include <thread>
#include <cstdint>
#include <vector>

#include "../../include/mtg_engine.h"
#include "host_graph.hpp"

namespace mtg {

struct Device;

int device_count();
// device memory of a call, reserved ahead of it (device_build.hip)
size_t device_call_bytes_estimate(uint64_t V, uint64_t E, uint64_t k);
void device_reserve_async(uint64_t V, uint64_t E, int device_id = -1);  // helper thread: HIP runtime, code objects, one arena chunk (-1: on the default device)
void device_arena_stats(int device_id, uint64_t out[4]);  // bytes in chunks, live bytes, peak of live bytes, chunks taken from the driver so far
void device_arena_reset_peak(int device_id);
void device_reserve_step_work(Device *d);  // MTG_DEVICE_RESERVE_WORK
size_t device_step_work_bytes_estimate(uint64_t V, uint64_t E);
void device_set_default(int device_id);
void device_set_reserve_ahead(int on);  // 0: host-only graph constructors reserve nothing on any GPU
void device_drop_foreign_reservation(const int *used, int n);  // a constructor's provisional chunk on a device the call did not use goes back
int device_get_default();
// lower_bounds: the goal-directed lower bounds (k <= 255) are computed with the graph; without them the search explores full balls
// (same candidate lists) until device_build_lower_bounds adds them
Device *device_create(const HostGraph &g, uint64_t k, int device_id, bool lower_bounds = true);
void device_build_lower_bounds(Device *d, void *stream);
double device_lower_bounds_ms(const Device *d);   // GPU time (HIP events) of that precompute, 0 if the device graph has none
bool device_has_lower_bounds(const Device *d);
void device_free(Device *d);
void device_set_single_use(Device *d);  // the caller searches once: the search's arrays go back before the claim replay (device_pairs.hip)
uint64_t device_graph_bytes(const Device *d);
uint64_t device_classify(Device *d, void *stream);
void device_classify_download(Device *d, void *stream, uint32_t *out_nodes, int32_t *mult, uint8_t *live);
const uint32_t *device_d_out_nodes(const Device *d);
uint64_t device_n_sources(const Device *d);
int device_sssp(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, uint64_t *d_pool, uint64_t pool_cap,
                uint64_t *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed);
void device_sssp_count(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats);
void device_sssp_count_visited(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats);
bool device_prunes(const Device *d);
uint64_t device_last_active_sources(const Device *d);
double device_last_kernel_ms(const Device *d);
const char *device_last_level_name(const Device *d, int level);
int device_last_levels(const Device *d, double *ms, uint64_t *sources, int cap);
int device_set_plan(Device *d, int plan);
int device_last_replay_rounds(const Device *d);
void device_set_replay_tuning(Device *d, uint64_t windows, int block, int grid, int role_mod, int plain_barrier);
void device_last_replay_ms(const Device *d, double out[2]);
uint64_t device_last_replay_visits(const Device *d);
void device_performance_data(Device *d, void *stream, mtg_dijkstra_performance_data *out);
uint64_t device_replay(Device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start, const uint32_t *d_cand_count,
                       const uint64_t *d_pool, mtg_pair **pairs_out, int *rounds_out);
// (pairs_out == nullptr in device_replay / device_pairs / device_pairs_multi: the pairs are not downloaded, they stay in the HBM of the
// (first) device for a finish there)
uint64_t device_pairs(Device *d, void *stream, mtg_pair **pairs_out, int *rounds_out);
int device_id_of(const Device *d);
bool device_matches(const Device *d, const HostGraph &g, uint64_t k);
const mtg_pair *device_resident_pairs(const Device *d, uint64_t *n_out);
mtg_pair *device_take_pairs(Device *d, uint64_t *n_out);
void device_free_array(int device_id, void *p);
uint64_t device_download_pairs(Device *d, mtg_pair **pairs_out);
// SURVEY 8e inside the library: sources block-partitioned by work over the devices, candidate lists gathered on devs[0]
void device_last_pairs_wall_s(const Device *d, double out[3]);  // host wall clock of the last device_pairs[_multi] on d: {SSSP stage (+ gather), claim replay, pair download}
uint64_t device_pairs_multi(Device *const *devs, int n_dev, mtg_pair **pairs_out, int *rounds_out, double *gather_ms_out);
std::vector<uint64_t> device_partition_sources(Device *d, void *stream, int parts);
// euler_device.hip: Euler bicycles on the GPU (valid, but not in the reference's order)
Walks device_euler_cycles(const HostGraph &g, int device_id, double *kernel_ms_out);
void device_euler_force_bitmap(int on);
// finish_device.hip: insertion + Euleriser + Euler bicycles + cut on the GPU (see mtg_finish_device)
// (d_pairs_resident: the n_pairs pairs as they lie in the HBM of `device_id`, e.g. left there by the claim replay -- `pairs` may then be null:
// nothing is uploaded, and the host graph gets its dummy weights from a download that runs beside the GPU stages)
// (times_out: 12 values, see mtg_last_finish_device_times / mtg_last_finish_device_stage_ms)
// (sink: the tigs go straight into a caller's clib.rs output arrays instead of a Walks object -- the host threads that empty the
// download ring write the flattened form, clib.rs:393-407 -- and the returned Walks is empty)
struct TigSink {
    int64_t *edge_out = nullptr;    // [>= kept edges]  +/- unitig id, 0 for a dummy edge (clib.rs:397-398)
    uint64_t *insert_out = nullptr; // [>= kept edges]  0 for an original edge, else the dummy's weight (clib.rs:399-403)
    uint64_t *limits_out = nullptr; // [>= tigs]        exclusive end of tig i (clib.rs:405-406)
    uint64_t n_tigs = 0, n_edges = 0;  // filled by the finish
    // a caller's helper thread that touches the pages of the three arrays (it WRITES zero bytes into them): the finish joins it before
    // its first result write, so that no helper write can land after a result (and then does not touch the arrays itself)
    std::thread *pretoucher = nullptr;
};
// (resident_out: the tigs stay in the HBM of `device_id` -- the returned Walks is empty, *resident_out owns the cutter's output
// arrays; whoever wants them on the host calls download(): a caller that asks for counts, flattens through a sink or spells on the GPU
// never pays for the 0.37-GB copy into pageable memory that a step at 2^27 used to end with)
struct ResidentTigs {
    int device = 0;
    uint32_t *d_edges = nullptr;   // [n_edges] edge ids of the tigs, one after the other
    uint32_t *d_limits = nullptr;  // [n_tigs]  exclusive end of tig i
    uint64_t n_edges = 0, n_tigs = 0;
    ResidentTigs() = default;
    ResidentTigs(const ResidentTigs &) = delete;
    ResidentTigs &operator=(const ResidentTigs &) = delete;
    ~ResidentTigs();
    void download(Walks &w) const;  // through the pinned ring; limits widened to 64 bits on the way
};
Walks device_finish(HostGraph &g, const Pair *pairs, uint64_t n_pairs, uint64_t k, int device_id, int euler_mode, double times_out[12],
                    const mtg_pair *d_pairs_resident = nullptr, TigSink *sink = nullptr, ResidentTigs **resident_out = nullptr);
void device_set_finish_tuning(int records, int flags, long record_delay_us);
void device_release_memory(int device_id);
uint64_t device_memory_held(int device_id);
void device_release_graph_cache(const HostGraph &g);
// synth_device.hip: the G-csr generator on the GPU
HostGraph *device_synth_g_csr(uint64_t n_binodes, uint64_t n_self_mirrors, uint64_t n_unitigs, uint64_t seed, uint64_t k,
                              const uint64_t *thresholds, uint64_t n_thresholds, int max_degree, int device_id);
// spell_device.hip: tig spelling on the GPU (bin.rs:466-606 / 667-818), byte-identical to spell.cpp for ACGT input
uint64_t device_write_walks_text(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                                 const char *seqs, const uint64_t *seq_off, bool gfa, const char *gfa_header, int device_id,
                                 char **out_buf, double *kernel_ms_out, uint64_t *bytes_out, const struct ResidentTigs *resident = nullptr);
void device_candidates_to_host(Device *d, void *stream, std::vector<uint64_t> &cand_start,
                               std::vector<uint32_t> &cand_count, std::vector<uint64_t> &pool);

}  // namespace mtg
