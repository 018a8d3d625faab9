// device_internal.hpp -- what the translation units of the device stage share (device_build.hip, device_classify.hip, device_sssp.hip,
// device_replay.hip, device_pairs.hip): the family block, the class flags, the counters, the kernel argument block, the Device object.
// Not an interface: device.hpp is what the host stages see.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mtg_policy.h"
#include "device.hpp"
#include "hip_util.hpp"
#include "parallel.hpp"

namespace mtg {

#ifndef HIP_CHECK
#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MTG_DIE("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Node records
// ------------------------------------------------------------------------------------------------
enum : uint8_t { F_TARGET = 1, F_EXT = 2, F_SOURCE = 4, F_SELF_MIRROR = 8, F_REACH = 16 };  // (F_REACH: class bytes only -- a source that can reach an in-node within the bound)
constexpr uint32_t ODEG_REACH = 0x80000000u;  // bit 31 of odeg[n] (8:8 format): lb+(n) <= k - 1, set once with the device graph (build_lbx_kernel)

// One 64-byte "family" block per node: the node's own <= 4 out-edges (first 32 bytes: what the cooperative levels read) AND, for as
// many of its children as fit, the child's in-node flag and the child's out-edges (grandchildren of the node). A path enumeration
// therefore spends ONE 64-byte gather on a node and its embedded children instead of one gather each: 1.8 visited nodes per gather
// on the bench graph (35 fetched bytes per visited node instead of 64; a 32-byte record costs a 64-byte request anyway).
// Everything in it is a function of the graph alone, so it is built once with the device graph (build kernels below).
// The encoding is chosen so that the enumeration decodes a block without a single table lookup or slot -> parent search: an unused
// weight slot holds 0xFFFF (never within a bound < 0x8000), and a grandchild slot holds the weight of the whole two-edge path.
struct alignas(64) NodeBlock {
    uint32_t nbr[4];   // words 0-3: inline neighbours; if F_EXT: nbr[0]/nbr[1] = ext_begin lo/hi, nbr[2] = ext_count
    uint16_t w[4];     // words 4-5: weights clamped to min(w, k) (an edge with w >= k can never lie on a <= k-1 path); unused slots 0xFFFF.
                       //            k <= 255 ("8:8 format"): low byte = that weight, high byte = min(255, weight + lb+(child)), lb+(v) = the
                       //            distance from v to the nearest initial in-node BEYOND v: the goal-directed lower bound of what a
                       //            search needs v's own block for (whether v is an in-node itself is told by cmeta)
    uint8_t deg;       // word 6: inline degree 0..4 (0 if F_EXT)
    uint8_t flags;     //         F_TARGET (initial in-node, greedytigs/mod.rs:231-240) | F_EXT
    uint16_t cmeta;    //         bit j (0-3): child j is an in-node; bit 4+j: child j is NOT embedded below (it needs its own gather);
                       //         8:8 format: bit 8+t: the neighbour in gnbr[t] is an in-node
    uint32_t gnbr[6];  // words 7-12: out-neighbours of the embedded children, children in order, each child's edges in order
    uint16_t gw[6];    // words 13-15: weight of the path node -> child -> that neighbour, saturated at 0xFFFF; unused slots 0xFFFF
                       //            (8:8 format: low byte = path weight, high byte = path weight + lb+(that neighbour), both saturated at 255)
};
static_assert(sizeof(NodeBlock) == 64, "NodeBlock must be 64 bytes");
constexpr int GSLOTS = 6;

// compute_eulerian_superfluous_out_biedges (bigraph) and the classification rule of greedytigs/mod.rs:229-245
struct NodeClass { int32_t diff; uint8_t cls; };
__device__ __forceinline__ NodeClass classify_node(uint32_t out_deg, uint32_t out_deg_mirror, bool self_mirror) {
    NodeClass c;
    c.diff = self_mirror ? (int32_t)(out_deg & 1u) : (int32_t)out_deg - (int32_t)out_deg_mirror;  // in_degree(n) == out_degree(mirror(n))
    c.cls = self_mirror ? F_SELF_MIRROR : 0;
    if (self_mirror && c.diff != 0) c.cls |= F_TARGET | F_SOURCE;  // :231-236
    else if (c.diff > 0) c.cls |= F_TARGET;                        // :237-240
    else if (c.diff < 0) c.cls |= F_SOURCE;                        // :241-244
    return c;
}

constexpr int CLS_BLOCK = 256, CLS_PER = 8;  // nodes per workgroup = 2048 (the single-workgroup scan in between sees V / 2048 counts)
constexpr int CLS_NODES = CLS_BLOCK * CLS_PER;

constexpr uint32_t CAND_OVERFLOW = 0xFFFFFFFFu;
constexpr unsigned long long TBL_EMPTY = 0xFFFFFFFFFFFFFFFFull;
// table entry: [63:54] local source (10 bits) | [53:22] node (32) | [21:1] distance (21) | [0] 1 = not (yet) known to be a target
constexpr int ENT_SRC_SHIFT = 54, ENT_NODE_SHIFT = 22;
constexpr unsigned long long ENT_DIST_MASK = 0x1FFFFFull;

enum Counter : int {
    C_BATCH = 0,     // (unused since kernel v2: batches are strided statically)
    C_POOL = 1,      // pool cursor (keys)
    C_OVERFLOW = 2,  // number of overflowed sources
    C_SETTLED = 3,
    C_RELAXED = 4,
    C_EMITTED = 5,
    C_ATTEMPTS = 6,
    C_OVF_LIST = 7,  // cursor of the overflow source list
    C_FIX = 8,       // enumeration level: number of candidate lists its post-pass has to put in order
    C_PUSHES = 9,    // COUNT: frontier-log items of all finished batches
    C_MAX_LOG = 10,  // COUNT: longest frontier log of one batch
    C_MAX_ENT = 11,  // COUNT: most table entries of one batch
    C_DEMAND = 12,   // classification: sum of the positive multiplicities (bounds the number of pairs)
    C_FIX_CLASS0 = 13,  // enumeration level's post-pass: number of work-list entries per length class (13, 14, 15)
    C_FIX_CURSOR0 = 16, // ... and the cursors of its compaction (16, 17, 18)
    C_ACTIVE = 19,      // sources of the launch's range that can reach an in-node within the bound (the only ones searched)
    C_ACT_BEGIN = 20,   // ... and where they start in the classification's list of such sources (active_range_kernel)
    C_COUNT = 24
};

struct SsspArgs {
    const NodeBlock *recs;  // family blocks (the cooperative levels read the first 32 bytes of each)
    const uint32_t *ext_col;
    const uint16_t *ext_w;
    const uint32_t *sources;     // out_nodes (ascending)
    const uint32_t *src_index;   // optional list of absolute source indices to process (re-runs); null = contiguous range
    uint64_t n_items;            // number of sources in this launch
    uint64_t src_begin;          // first absolute source index (outputs are indexed by abs - src_begin)
    uint32_t K1;                 // bound k-1 (inclusive)
    unsigned long long *pool;
    uint64_t pool_cap;
    unsigned long long *cand_start;
    uint32_t *cand_count;
    unsigned long long *counters;
    unsigned long long *ws;      // global workspace (GLOBAL_WS levels)
    uint64_t ws_stride;          // 64-bit words per block
    uint32_t *ovf_list;          // out: absolute indices of the sources this launch could not finish (cursor: C_OVERFLOW)
    unsigned long long *fix_val; // out (enumeration level): beside every work-list entry its list's place, start << 8 | count -- the post-pass
                                 // reads its lists' places in the order of its work list, not from cand_start / cand_count at the sources' indices
    uint32_t *fix_list;          // out (enumeration level): post-pass work list in chunks of ENUM_FIX_CHUNK slots (cursor: C_FIX): slot 0 = the
                                 // chunk's length class, then indices (relative to src_begin) of lists of that class, FIX_NONE = unused
    uint32_t wmask;              // inline weight slots: 0xFF in the 8:8 format (k <= 255: weight | weight + lower bound << 8), else 0xFFFF
    uint32_t prune;              // cooperative levels: 1 = skip a successor whose lower bound puts every in-node behind it beyond the bound
    const uint32_t *act_index;   // enumeration level with pruning: the classification's list of sources that can reach an in-node (absolute
    const uint32_t *act_node;    // indices, ascending) and their nodes; the launch's part of it is counters[C_ACT_BEGIN] .. + counters[C_ACTIVE]
                                 // (neither number travels to the host before the launch)
};

struct Touch;  // replay_kernels.inc
struct Dense;
struct ReplayWork {
    uint64_t cap_v = 0, cap_s = 0, cap_spill = 0, cap_blocks = 0, cap_out = 0;
    unsigned long long *state = nullptr;      // [V] working node states {mirror, multiplicity, live}
    unsigned long long *resv[2] = {nullptr, nullptr};
    uint32_t tag_base = 0xFFFFFFFFu;          // reservation tags used so far (the arrays are never cleared between calls)
    Touch *touch = nullptr;
    uint32_t *src_mirror = nullptr;
    Dense *dense = nullptr;                   // the sources that have candidates (replay_kernels.inc)
    uint64_t cap_dense = 0;
    unsigned long long *claims = nullptr;
    uint32_t *pending[2] = {nullptr, nullptr}, *spill = nullptr;
    unsigned long long *final_off = nullptr, *block_sums = nullptr;
    unsigned long long *ctl = nullptr, *h_ctl = nullptr;  // control block (device / pinned host copy)
    mtg_pair *out = nullptr;
    mtg_pair *h_out = nullptr;                // pinned staging of the pair download (pageable D2H runs at a few GB/s)
    uint64_t cap_h_out = 0;
    unsigned grid = 0, grid_small = 0;        // co-resident workgroups of the cooperative launch (workgroups of 1024 / of 256)
};

struct Device {
    int dev = 0;
    uint64_t k = 0;
    uint32_t K1 = 0;
    uint64_t V = 0;
    uint64_t E0 = 0;  // original edges of the graph the device copy was built from
    double lower_bounds_ms = 0.0;  // GPU time of the lower-bound precompute (0 without)
    bool single_use = false;  // the caller searches once (mtg_compute_tigs_cfg): what only the search needs goes back before the claim replay takes its arrays
    NodeBlock *d_recs = nullptr;              // [V] family blocks
    uint32_t *d_odeg = nullptr;               // [V] out-degree (classification)
    uint8_t *d_cls = nullptr;                 // [V] class byte of the last classification (F_TARGET | F_SOURCE | F_SELF_MIRROR)
    uint32_t *d_ext_col = nullptr;
    uint16_t *d_ext_w = nullptr;
    uint64_t ext_n = 0;
    int32_t *d_mult = nullptr;
    uint32_t *d_out_nodes = nullptr;
    uint32_t *d_block_counts = nullptr;
    uint64_t n_cls_blocks = 0;
    uint64_t n_sources = 0;
    uint64_t total_demand = 0;  // sum of the positive multiplicities (classification)
    bool classified = false;
    unsigned long long *d_counters = nullptr;
    unsigned long long *h_counters = nullptr;  // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_r[4] = {nullptr, nullptr, nullptr, nullptr};  // claim replay: start, before / after the rounds kernel, end of the GPU work
    double last_replay_kernel_ms = 0.0, last_replay_gpu_ms = 0.0;
    double last_wall_s[3] = {0, 0, 0};  // host wall clock of the last device_pairs[_multi]: SSSP stage (+ gather), claim replay, pair download
    double last_kernel_ms = 0.0;  // sum of the SSSP level kernels' HIP-event durations of the last call
    uint32_t *d_mirror = nullptr;             // [V] mirror node (claim replay)
    uint32_t *d_ovf[2] = {nullptr, nullptr};  // ping-pong overflow source lists
    unsigned long long *d_fix_val = nullptr, *d_fix_dense_val = nullptr;  // ... and the lists' places beside the entries
    uint32_t *d_fix = nullptr, *d_fix_dense = nullptr;  // enumeration level: work list of its post-pass (chunked, as written / dense, by length class)
    uint64_t ovf_cap = 0;
    int last_n_levels = 0;           // per-level record of the last call: kernel ms and sources handed to the level
    double last_level_ms[8] = {0};
    uint64_t last_level_sources[8] = {0};
    std::string last_level_name[8];
    int plan = 0;  // 0 = table-free path enumeration per lane, then the cooperative cascade for the heaviest sources (the form of its gathers
                   // chosen by the size of the graph); 1 = cascade only; 2 / 3 = plan 0 with quad-cooperative / per-lane gathers regardless of size;
                   // + 4 = without the goal-directed pruning (full balls, every source searched: A/B runs and tests)
    bool w8 = false;                 // blocks in the 8:8 format with lower bounds (k <= 255)
    // the sources that can reach an in-node (8:8 format), in order, written by the classification (cap: act_cap sources); per-block
    // counts of them (d_act_blocks, n_cls_blocks words) and their number (d_act_total, on the device)
    uint32_t *d_act_index = nullptr, *d_act_node = nullptr, *d_act_blocks = nullptr;
    unsigned long long *d_act_total = nullptr;
    uint64_t act_cap = 0;
    uint64_t last_active_sources = 0;
    int n_cu = 256;
    uint64_t graph_bytes = 0;
    ReplayWork replay;
    // claim replay tuning (0 = the engine's choice; never changes a result: tests run the rounds under several settings)
    uint64_t tune_windows = 0;
    int tune_block = 0, tune_grid = 0, tune_role_mod = 0;
    bool tune_plain_barrier = false;  // every workgroup releases at the grid barrier (no per-XCD stage)
    int last_replay_rounds = 0;
    uint64_t last_n_pairs = 0;        // pairs of the last claim replay; they stay in replay.out until the next one (or device_take_pairs)
    uint64_t last_replay_visits = 0;  // sum over the rounds of the pending-list lengths (first RC_TRACE_ROUNDS rounds)
};

// ---- shared between the translation units ----
void read_counters(Device *d, hipStream_t st);                      // device_sssp.hip: counters -> d->h_counters (synchronises)
float elapsed_ms(Device *d);                                        // ev0 .. ev1
int run_levels(Device *d, hipStream_t st, int count_mode, uint64_t src_begin, uint64_t src_end, unsigned long long *d_pool, uint64_t pool_cap,
               unsigned long long *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed, mtg_sssp_stats *stats);
void scan_u32(Device *d, hipStream_t st, ReplayWork &w, const uint32_t *in, uint64_t n, unsigned long long *out, unsigned long long *d_total);  // device_replay.hip
void device_drop_search_arrays(Device *d);
// one kernel's attributes per translation unit of the stage: asking for them loads the unit's code object (3-10 ms otherwise, inside its first launch)
void device_warm_classify_unit(hipFuncAttributes *a);
void device_warm_sssp_unit(hipFuncAttributes *a);
void device_warm_replay_unit(hipFuncAttributes *a);
void device_warm_pairs_unit(hipFuncAttributes *a);                          // device_pairs.hip
constexpr uint32_t ENUM_POOL_CHUNK = 2048;  // keys per wave-local pool chunk of the enumeration level (one global atomic per chunk)

}  // namespace mtg
