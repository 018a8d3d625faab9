/*
 * mtg_engine.h -- engine-level C-ABI underneath matchtigs.h (NEW; the reference has no such layer).
 *
 * It is what both the clib.rs-compatible shim (matchtigs_compute_tigs) and a Rust
 * `impl TigAlgorithm for GreedytigAlgorithm` body (src/implementation/greedytigs/mod.rs:75-90,
 * see INTEGRATION.md) call: an edge-centric bigraph in, (out,in,distance) pairs and tig walks out.
 * Stages are exported separately so that they can be parity-checked tier by tier and so that
 * bench.py / a multi-GPU driver can shard the SSSP stage across ranks:
 *
 *   stage                          reference lines it replaces                      where it runs
 *   mtg_classify                   greedytigs/mod.rs:222-255 (eulertigs :64-97)     GPU (HIP)
 *   mtg_sssp_candidates            greedytigs/mod.rs:324-335 + traitgraph-algo      GPU (HIP)
 *                                  Dijkstra::shortest_path_lens (all sources at once)
 *   mtg_replay_claims_device       greedytigs/mod.rs:301-523 (1-thread order)       GPU (HIP)
 *   mtg_replay_claims              the same loop over host arrays (A/B, tests)      host (C++)
 *   mtg_finish_greedytigs          greedytigs/mod.rs:678-801 + implementation/      host (C++); Euler bicycles optionally
 *                                  mod.rs:392-649 + bigraph Euler decomposition     on the GPU (mtg_config.euler_mode)
 *   mtg_compute_eulertigs          eulertigs/mod.rs:48-198                          host (C++); idem
 *   mtg_write_walks_fasta / _gfa   bin.rs:466-606 / 667-818                         host (C++)
 *   mtg_read_bcalm2                bin.rs:902-912 (genome-graph bcalm2 reader)      host (C++)
 *
 * All pointers are plain host pointers unless the name starts with d_ (device pointer in the
 * HBM of the GPU the mtg_device was created on). `stream` is a hipStream_t passed as void*
 * (NULL = the default stream). No torch types anywhere.
 *
 * Errors follow the reference's convention (panic => abort): message on stderr + abort().
 * Functions that can fail recoverably return a status code instead, documented per function.
 */
#ifndef MTG_ENGINE_H
#define MTG_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtg_graph mtg_graph;   /* host bigraph; grows when dummy edges are inserted */
typedef struct mtg_device mtg_device; /* one GPU: device-resident graph + workspaces */
typedef struct mtg_walks mtg_walks;   /* a list of edge walks (tigs or Euler cycles) */

/* (out_node, in_node, distance): one matched pair, greedytigs/mod.rs:461. */
typedef struct {
    uint32_t out_node;
    uint32_t in_node;
    uint64_t distance;
} mtg_pair;

/* Counters of one SSSP stage (SURVEY.md 8d "unit of work"). */
typedef struct {
    uint64_t sources;          /* sources processed */
    uint64_t settled_nodes;    /* (source,node) pairs with distance <= k-1 (full ball) */
    uint64_t relaxed_edges;    /* sum of out-degrees of settled nodes = SSSP edges */
    uint64_t emitted;          /* candidates written */
    uint64_t relax_attempts;   /* edge relaxations the label-correcting kernel really performed */
    uint64_t overflow_sources; /* sources that needed a larger kernel level */
} mtg_sssp_stats;

/* Engine configuration = the fields of GreedytigAlgorithmConfiguration (greedytigs/mod.rs:40-73; the C-ABI fixes
 * them at clib.rs:378-389, the CLI fills them from bin.rs:1074-1082) plus what only this engine has.
 * threads / staged_parallelism_divisor / resource_limit_factor / node_weight_array_type / heap_type select CPU data
 * structures and scheduling in the reference and never change its 1-thread result; the engine validates and otherwise
 * ignores them (results always equal the reference's 1-thread order, its only deterministic one). */
enum { MTG_NODE_WEIGHT_EPOCH_ARRAY = 0, MTG_NODE_WEIGHT_HASHBROWN_HASH_MAP = 1 }; /* implementation/mod.rs:61-81 */
enum { MTG_HEAP_STD_BINARY_HEAP = 0 };                                            /* implementation/mod.rs:83-102 */
enum { MTG_PERFORMANCE_DATA_NONE = 0, MTG_PERFORMANCE_DATA_COMPLETE = 1 };        /* implementation/mod.rs:104-126 */
enum { MTG_EULER_HOST_REFERENCE_ORDER = 0, MTG_EULER_DEVICE = 1 };
/* Where dummy insertion, the Euleriser and the cutter run: AUTO = on the GPU (device_ids[0]) whenever one is visible and the
 * graph holds only its original edges, else the host stages; HOST / DEVICE force one (DEVICE aborts without a GPU). Both give
 * the same edges and tigs: the GPU Euleriser reproduces the reference's sequence of breaking edges. With
 * MTG_EULER_HOST_REFERENCE_ORDER the device stage builds the walk's node records and the host only walks. */
enum { MTG_FINISH_AUTO = 0, MTG_FINISH_HOST = 1, MTG_FINISH_DEVICE = 2 };
#define MTG_MAX_DEVICES 8
typedef struct {
    uint64_t struct_size;              /* sizeof(mtg_config) of the header the caller was built against: set by mtg_config_init and checked by
                                          every function that takes a configuration. ALWAYS fill a configuration through mtg_config_init.
                                          New fields are only ever APPENDED: a caller built against an older header (struct_size smaller,
                                          down to MTG_CONFIG_MIN_SIZE = the layout of round 4) keeps working -- the fields it does not
                                          know take the values mtg_config_init gives them --, a struct_size LARGER than this library's
                                          (a caller newer than the library) or below the minimum aborts with a message. */
    uint64_t threads;                  /* greedytigs/mod.rs:42 */
    uint64_t k;                        /* :44 */
    double staged_parallelism_divisor; /* :47, 0 = None */
    uint64_t resource_limit_factor;    /* :49 */
    int32_t node_weight_array_type;    /* :51 */
    int32_t heap_type;                 /* :53 */
    int32_t performance_data_type;     /* :55; COMPLETE additionally runs the counting kernels (mtg_last_performance_data) */
    int32_t euler_mode;                /* MTG_EULER_HOST_REFERENCE_ORDER (bit-exact tigs) or MTG_EULER_DEVICE (valid walks, same
                                          #tigs / cumulative length, different order) */
    int32_t n_devices;                 /* GPUs to shard the SSSP sources over; >= 1 */
    int32_t device_ids[MTG_MAX_DEVICES];
    /* MatchtigAlgorithmConfiguration (matchtigs/mod.rs:33-45), tig algorithm 4 only; may be NULL otherwise */
    const char *matching_file_prefix; /* the instance goes to <prefix>.minimalperfectmatching, the matcher writes <that>.solution */
    const char *matcher_path;         /* blossom5-compatible executable: `<matcher> -e <instance> -w <solution>` */
    int32_t finish_stage;              /* MTG_FINISH_AUTO / _HOST / _DEVICE (appended in round 3) */
    int32_t reserved0;
} mtg_config;
#define MTG_CONFIG_MIN_SIZE 120 /* sizeof(mtg_config) of round 4, the oldest layout with struct_size in front */
/* GreedytigAlgorithmConfiguration::new(threads, k) (greedytigs/mod.rs:62-72): staged None, factor 0, HashbrownHashMap,
 * StdBinaryHeap, performance data None; engine fields: host Euler walk, one device (id 0). */
void mtg_config_init(mtg_config *cfg, uint64_t threads, uint64_t k);

/* The reference's Dijkstra performance counters (logged at greedytigs/mod.rs:647-673; DijkstraPerformanceData of
 * traitgraph-algo) in the engine's terms, over all queries of the last greedy run with performance_data_type COMPLETE.
 * The engine searches the FULL (k-1)-ball of every source with a label-correcting wavefront, one source per workgroup in
 * this mode: iterations = expansions of a settled (source,node) pair; heap pushes = frontier-log items; unnecessary heap
 * elements = log items superseded by a shorter distance before they were expanded (the analogue of a stale heap entry);
 * max heap size of a query = its frontier-log length; max distance array size of a query = its table entries. */
typedef struct {
    uint64_t dijkstras;                 /* number of queries */
    uint64_t iterations;
    uint64_t heap_pushes;
    uint64_t unnecessary_heap_elements;
    uint64_t max_max_heap_size;
    uint64_t max_max_distance_array_size;
    uint64_t sum_max_heap_size;           /* average_max_heap_size = sum / dijkstras */
    uint64_t sum_max_distance_array_size; /* average_max_distance_array_size = sum / dijkstras */
} mtg_dijkstra_performance_data;

/* Library / device probes. */
const char *mtg_version(void);
int mtg_device_count(void); /* hipGetDeviceCount; 0 when no GPU (never initialises a context) */
/* Which setting of the four out-of-tree policies (include/mtg_policy.h: heap tie-break, inclusive bound, adjacency order, union-find
 * tie) this library was built with: bit i set = policy P(i+1) flipped. 0 for libmatchtigs.so; 15 for the libmatchtigs_flipped.so
 * that `make flipped` builds for the parity suite. */
unsigned mtg_policies(void);

/* ---- host graph ------------------------------------------------------------------------ */
/* Edges in id order: edge 2u = unitig u forwards, edge 2u+1 = its mirror (clib.rs:239-248).
 * weight = k-mers of the unitig (bin.rs:357-379). Validates node pairing and the edge mirror
 * property like clib.rs:251-252 (abort on violation). Arrays are copied. */
mtg_graph *mtg_graph_from_edges(uint64_t n_nodes, const uint32_t *mirror, uint64_t n_edges,
                                const uint32_t *edge_from, const uint32_t *edge_to,
                                const uint64_t *edge_weight);
/* The clib.rs builder as three calls (matchtigs.h wraps exactly these). */
mtg_graph *mtg_graph_builder_new(uint64_t unitig_amount);
void mtg_graph_builder_merge(mtg_graph *g, uint64_t unitig_a, int strand_a, uint64_t unitig_b, int strand_b);
/* n links at once: row i = (unitig_a, strand_a, unitig_b, strand_b) as int64; the same unions in the same order as n merges. */
void mtg_graph_builder_merge_links(mtg_graph *g, uint64_t n_links, const int64_t *links);
void mtg_graph_builder_build(mtg_graph *g, const uint64_t *unitig_weights);
void mtg_graph_free(mtg_graph *g);
/* Removes every dummy edge again (undoes what compute_tigs appended), restoring the adjacency order of the
 * original graph exactly. The reference instead clones the graph before each algorithm (bin.rs:1069). */
void mtg_graph_reset(mtg_graph *g);
uint64_t mtg_graph_node_count(const mtg_graph *g);
uint64_t mtg_graph_edge_count(const mtg_graph *g); /* includes dummy edges added so far */
/* Copy out the current graph (any pointer may be NULL). forwards: 1/0. dummy_id 0 = original. */
void mtg_graph_export(const mtg_graph *g, uint32_t *mirror, uint32_t *edge_from, uint32_t *edge_to,
                      uint64_t *edge_weight, uint64_t *edge_dummy_id, uint64_t *edge_unitig,
                      uint8_t *edge_forwards);

/* The same for the edges [first_edge, first_edge + n_edges) only (large graphs: checks that stream over the edge arrays). */
void mtg_graph_export_range(const mtg_graph *g, uint64_t first_edge, uint64_t n_edges, uint32_t *edge_from, uint32_t *edge_to,
                            uint64_t *edge_weight, uint64_t *edge_dummy_id, uint64_t *edge_unitig, uint8_t *edge_forwards);
uint64_t mtg_graph_original_edge_count(const mtg_graph *g); /* edges before any dummy edge (2 x unitigs) */

/* ---- device stage ---------------------------------------------------------------------- */
/* Uploads the original edges to GPU `device_id` and builds the 64-byte family blocks for bound k-1 there.
 * Aborts if no GPU is present (there is no CPU path). Weights must be >= 1 for algorithm 5
 * (checked here; the reference's (distance, node) pop order needs it). */
mtg_device *mtg_device_create(const mtg_graph *g, uint64_t k, int device_id);
/* The same with options. MTG_DEVICE_NO_LOWER_BOUNDS: the goal-directed lower bounds (k <= 255, see mtg_sssp_count_visited) are NOT
 * computed with the graph -- they cost k - 1 passes over the nodes and a rewrite of the blocks, several times what they save ONE search;
 * the search then explores full balls (identical candidate lists). mtg_compute_tigs_cfg, whose device graph is searched once (the
 * reference's calling convention, clib.rs:291), builds it this way; a caller that holds an mtg_device and iterates keeps the default,
 * or adds the bounds later with mtg_device_build_lower_bounds (which repeats the classification if there was one).
 * mtg_device_lower_bounds_ms: GPU time (HIP events) of that precompute, 0 while the device graph has none.
 * MTG_DEVICE_RESERVE_WORK: for a caller that will STEP through the stages with this device graph (mtg_classify / mtg_sssp_candidates /
 * mtg_replay_claims_* / a finish on the same GPU), once or again and again: the device memory those stages take beside the graph
 * (about 60 bytes per node + 62 per edge at their peak) is taken from the driver now, in one piece, unless the library's arena holds it
 * already -- the first step then makes no driver allocation (it makes five otherwise, and single driver calls sporadically stall for
 * a second on shared hosts). Nothing is reserved if that would leave less than a sixth of the device to the process's other
 * allocators. Never changes a result. mtg_compute_tigs_cfg does not need it: its whole call is reserved when the graph is constructed. */
enum { MTG_DEVICE_DEFAULT = 0, MTG_DEVICE_NO_LOWER_BOUNDS = 1, MTG_DEVICE_RESERVE_WORK = 2 };
mtg_device *mtg_device_create_opts(const mtg_graph *g, uint64_t k, int device_id, int flags);
void mtg_device_build_lower_bounds(mtg_device *d, void *stream);
double mtg_device_lower_bounds_ms(const mtg_device *d);
void mtg_device_free(mtg_device *d);
/* Bytes of HBM held by the device graph. */
uint64_t mtg_device_graph_bytes(const mtg_device *d);

/* Node classification on the GPU. Returns the number of sources (out-nodes, ascending). */
uint64_t mtg_classify(mtg_device *d, void *stream);
/* Host copies of the classification (sizes: n_sources, n_nodes, n_nodes; any may be NULL). */
void mtg_classify_download(mtg_device *d, void *stream, uint32_t *out_nodes, int32_t *multiplicity,
                           uint8_t *is_in_node);
/* Device pointer to the ascending out-node list (uint32[n_sources]); valid until the next classify. */
const uint32_t *mtg_classify_d_out_nodes(const mtg_device *d);

/* Bounded many-to-many SSSP: for every source index i in [src_begin, src_end) writes its
 * candidate list L(s_i) = all in-nodes within k-1 of s_i (s_i excluded) as keys
 * (distance << 32 | node), ascending, into d_pool[d_cand_start[i-src_begin] ..+ d_cand_count[..]]
 * (d_cand_start of a source with d_cand_count 0 is left as it was: the start of an empty list means nothing).
 * Returns 0 on success; 1 if pool_capacity was too small (*pool_needed tells the size to retry
 * with; outputs are then invalid). Sources whose ball does not fit the fast kernel's LDS tables
 * are re-run by larger kernel levels internally, down to a dense level without any limit on the ball
 * (no legal input aborts: the reference's search has no size limit either, greedytigs/mod.rs:548-551).
 * k may exceed 65535 as long as every unitig weight stays below 65535 (weights are 16-bit in HBM).
 * Synchronises `stream` before returning. */
int mtg_sssp_candidates(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end,
                        uint64_t *d_pool, uint64_t pool_capacity, uint64_t *d_cand_start,
                        uint32_t *d_cand_count, uint64_t *pool_needed);
/* Sum of the HIP-event durations (ms) of the SSSP level kernels of the last mtg_sssp_candidates call. */
double mtg_last_sssp_kernel_ms(const mtg_device *d);
/* Per level of that call: kernel duration (ms) and number of sources handed to the level. Level 0 is the
 * lane-per-source kernel (or the first cooperative level with preset 4); later levels re-run overflowed sources.
 * Returns the number of levels written (<= capacity). */
int mtg_last_sssp_levels(const mtg_device *d, double *ms_out, uint64_t *sources_out, int capacity);
/* Kernel instantiation that ran as level `level` of the last call ("" beyond the last level); valid until the next call. */
const char *mtg_last_sssp_level_name(const mtg_device *d, int level);
/* Runs the counting variant of the kernel (untimed instrumentation) over the same sources. */
void mtg_sssp_count(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats);
/* Goal-directed pruning (k <= 255): with the device graph the engine computes lb(v) = distance from v to the nearest
 * initial in-node and lb+(v) = distance to the nearest one BEYOND v, and keeps weight + lb+(head) beside every edge weight and the
 * in-node flags of a node's children and grandchildren in its block. A search records an in-node it reaches at distance d <= k-1
 * from the block of its parent, and expands a node v -- reads its block, relaxes its out-edges -- only when d + lb+(v) <= k-1:
 * no in-node behind v can be within the bound otherwise, so the candidate lists are unchanged (the reference truncates its
 * searches as well, by target_amount, greedytigs/mod.rs:323-335); sources that cannot reach any in-node within the bound
 * are not searched at all. mtg_sssp_count keeps counting FULL balls (the unit of work of SURVEY 8d, equal to a full-ball
 * Dijkstra's counters); mtg_sssp_count_visited counts what the pruned search does: stats->sources = sources searched,
 * settled_nodes = distinct (source, node) pairs whose distance it determines, relaxed_edges = the out-edges of the pairs it
 * expands, emitted = the same candidates. Aborts when the device graph / plan does not prune (mtg_sssp_prunes() == 0). */
void mtg_sssp_count_visited(mtg_device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats);
int mtg_sssp_prunes(const mtg_device *d);
/* Sources the enumeration level of the last mtg_sssp_candidates call searched (0 when it does not prune). */
uint64_t mtg_last_sssp_searched_sources(const mtg_device *d);
/* Which SSSP level plan runs: 0 = default (table-free path enumeration per lane + sorting post-pass, then the cooperative
 * cascade for the sources it hands on; the level gathers its 64-byte node blocks per lane, or four lanes per block once the
 * blocks exceed 3 GB), 1 = cooperative cascade only (exact for any ball; the fallback plan and the one the counting kernels
 * use), 2 / 3 = plan 0 with the four-lanes-per-block / per-lane form of the gathers regardless of the graph's size (same
 * results; lets small graphs exercise both forms); 4 / 6 / 7 = plans 0 / 2 / 3 WITHOUT the goal-directed pruning (full balls, every
 * source searched; same results: A/B runs and tests); + 8 = the enumeration level on one workgroup (same results, slow: lets a test graph
 * of a few thousand sources take a wave through many chunks of sources). Returns the plan in force. */
int mtg_set_sssp_plan(mtg_device *d, int plan);

/* The claim loop (greedytigs/mod.rs:301-523, 1-thread order) on the GPU, over the candidate lists of ALL classified
 * sources (device arrays indexed by absolute source index, e.g. straight from mtg_sssp_candidates or an all-gather):
 * deterministic reservations by source index reproduce the sequential result exactly. Returns the number of pairs;
 * *pairs_out (HOST memory, malloc'd, free with mtg_free) holds them in the reference's push order. */
uint64_t mtg_replay_claims_device(mtg_device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start,
                                  const uint32_t *d_cand_count, const uint64_t *d_pool, mtg_pair **pairs_out);
/* The same claim loop, but the pairs STAY in the HBM of d's GPU (no download): for a finish on the same GPU, which takes them from
 * there (mtg_finish_greedytigs_resident) -- mtg_compute_tigs_cfg works this way. Returns the number of pairs. They remain valid until
 * the next claim replay on d or mtg_device_free(d). */
uint64_t mtg_replay_claims_resident(mtg_device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start,
                                    const uint32_t *d_cand_count, const uint64_t *d_pool);
/* Device pointer to those pairs (NULL when there are none) and their number. */
const mtg_pair *mtg_resident_pairs(const mtg_device *d, uint64_t *n_pairs_out);
/* A host copy of them (malloc'd, free with mtg_free), in the reference's push order. */
uint64_t mtg_download_resident_pairs(mtg_device *d, mtg_pair **pairs_out);
/* Geometry of the claim replay's cooperative launch, for measurements and tests (0 = the engine's choice for that parameter): number
 * of index-ordered admission windows (bits 0-15; bits 16-23: the sixteenth of them from which they grow, bits 24-31: by which factor
 * -- the engine's own choice is 36 windows, the second half twice as large), threads per workgroup (1024 or 256), workgroups, one admitting workgroup in `role_mod`, and
 * plain_barrier != 0 = every workgroup releases at the grid barrier (without the per-XCD stage that leans on gfx942 / gfx950
 * hardware). The pair list never depends on any of them. (The library reads no environment variable for this.) */
void mtg_set_replay_tuning(mtg_device *d, uint64_t windows, int block, int grid, int role_mod, int plain_barrier);
/* GPU time (HIP events, ms) of the last claim replay on d: [0] the rounds kernel (replay_rounds_kernel), [1] all of its GPU work
 * (state copy, dense list, rounds, pair-count scan, compaction). */
void mtg_last_replay_ms(const mtg_device *d, double out[2]);
/* Reservation rounds the last mtg_replay_claims_device needed. */
int mtg_last_replay_rounds(const mtg_device *d);
/* Source visits of those rounds (sum of the pending-list lengths): the unit of the replay's cost model. */
uint64_t mtg_last_replay_visits(const mtg_device *d);

/* ---- multi-GPU ------------------------------------------------------------- */
/* Two forms, by design:
 *  - INSIDE the library (this function, mtg_config.device_ids): ONE process drives n GPUs from n host threads, and the candidate
 *    lists travel to the first GPU as peer copies (hipMemcpyPeerAsync over xGMI: three copies per device, counts / starts / keys).
 *    The library does not link RCCL: a collective library adds nothing to a gather whose pieces already sit in one address space
 *    and whose replay order IS the device order, and it would make libmatchtigs.so depend on a communicator the reference's
 *    callers (clib.rs: one thread, no runtime) do not have.
 *  - ONE PROCESS PER GPU (matchtigs_amd/distributed.py, what bench.py runs under torch.distributed.run): the same exchange as
 *    grouped broadcasts at exact sizes over torch.distributed's "nccl" backend = RCCL over xGMI. This is the form north_star words
 *    ("RCCL all-gather of matched pairs"), and the only one in which ranks do not share an address space.
 * Both give the pair list of one GPU bit for bit (tests: gloo world 2 / 3, several resident copies on one GPU). Neither has run on
 * two distinct GPUs yet: the development pool hands out one GPU per job. */
/* SSSP + gather + claim replay over ALL classified sources of n_devices resident copies of the SAME graph (each created
 * with mtg_device_create on its GPU and classified): the ascending source list is cut into contiguous blocks of equal
 * estimated work (1 + out-degree of the source), device i searches block i from its own host thread, the candidate lists
 * are gathered on devices[0] with peer copies over xGMI (concatenation in device order is the replay order) and the claim
 * loop runs there. The pair list is identical for every n_devices. *pairs_out: host memory, free with mtg_free.
 * mtg_compute_tigs_cfg does exactly this for mtg_config.device_ids. */
uint64_t mtg_compute_pairs(mtg_device *const *devices, int n_devices, mtg_pair **pairs_out);
double mtg_last_gather_ms(void); /* wall-clock of the peer-copy gather of the last mtg_compute_pairs on this thread */
/* The block boundaries that split uses: cuts_out[0..parts], cuts_out[0] = 0, cuts_out[parts] = number of sources. */
void mtg_partition_sources(mtg_device *d, int parts, uint64_t *cuts_out);

/* ---- optimal matchtigs around the external matcher ---------------------------- */
/* matchtigs/mod.rs:150-940 minus the matcher itself. The reference runs one bounded Dijkstra per out-node with
 * target_amount = #in-nodes (:235-246) -- exactly the candidate lists of mtg_sssp_candidates -- and folds them into a
 * minimum-perfect-matching instance: collapsed matching nodes (GraphMatchingNodeMap, implementation/mod.rs:188-250), the
 * (min, max) -> (weight, out, target) edge map, two copies of that graph joined by weight-(k-1) edges, four extra nodes per
 * weakly connected component. The lists come from the GPU (cfg->device_ids[0]); the instance equals the reference's
 * threads == 1 result (with more threads the reference's node numbering depends on thread timing). */
typedef struct mtg_matching mtg_matching;
typedef struct {
    uint64_t transformed_node_count; /* matching nodes of one copy (:531) */
    uint64_t edge_count;             /* edges.len() (:535) */
    uint64_t wcc_amount;             /* WCCs that contain a matching node (:565) */
    uint64_t matching_node_count;    /* first number of the instance file (:598) */
    uint64_t matching_edge_count;    /* second number (:600) */
    uint64_t mirror_biedges, mirror_expanded_biedges; /* the two counts logged at :537-540 */
} mtg_matching_stats;
mtg_matching *mtg_matching_instance(const mtg_graph *g, const mtg_config *cfg);
/* The host stage alone, over candidate lists the caller holds (layout of mtg_replay_claims; multiplicity as downloaded by
 * mtg_classify_download: negative for out-nodes). */
mtg_matching *mtg_matching_instance_from_lists(const mtg_graph *g, uint64_t k, uint64_t n_sources, const uint32_t *out_nodes,
                                               const int32_t *multiplicity, const uint64_t *cand_start,
                                               const uint32_t *cand_count, const uint64_t *pool);
void mtg_matching_get_stats(const mtg_matching *m, mtg_matching_stats *out);
/* Writes the instance text (:591-719) to `path`; returns the bytes written. */
uint64_t mtg_matching_write(const mtg_matching *m, const char *path);
/* Reads a matcher solution (:746-812): the matched pairs (original out-node, in-node, weight) in file order, ready for
 * mtg_finish_matchtigs_cfg. *pairs_out: malloc'd, free with mtg_free. Aborts on an edge that is not in the instance (:794). */
uint64_t mtg_matching_read_solution(const mtg_matching *m, const char *solution_path, mtg_pair **pairs_out);
void mtg_matching_free(mtg_matching *m);
/* Inserts the matched edges, Eulerises, walks and cuts (:797-935): mtg_finish_greedytigs_cfg plus the reference's assertion
 * that every bicycle with a dummy edge holds a breaking edge (:883). */
mtg_walks *mtg_finish_matchtigs_cfg(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg);
/* The whole algorithm (tig algorithm 4 of mtg_compute_tigs_cfg): instance -> cfg->matcher_path as a child process ->
 * solution -> matchtigs. */
mtg_walks *mtg_compute_matchtigs_cfg(mtg_graph *g, const mtg_config *cfg);

/* ---- host stages ----------------------------------------------------------------------- */
/* Replays the reference's claim loop over the candidate lists in ascending source order.
 * cand_start/cand_count index `pool`. multiplicity / is_in_node are the classification
 * (copied, not modified). Returns the number of pairs; *pairs_out is malloc'd (mtg_free). */
uint64_t mtg_replay_claims(const mtg_graph *g, uint64_t n_sources, const uint32_t *out_nodes,
                           const int32_t *multiplicity, const uint8_t *is_in_node,
                           const uint64_t *cand_start, const uint32_t *cand_count,
                           const uint64_t *pool, mtg_pair **pairs_out);
void mtg_free(void *p);

/* Dummy insertion + Eulerisation + Euler decomposition + cut. Mutates g like the reference. */
mtg_walks *mtg_finish_greedytigs(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, uint64_t k);
mtg_walks *mtg_compute_eulertigs(mtg_graph *g, uint64_t k);
/* Sub-stages, exported for parity tests. */
uint64_t mtg_insert_pair_edges(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs); /* returns last dummy id */
uint64_t mtg_make_eulerian(mtg_graph *g, uint64_t dummy_edge_id, uint64_t k);          /* returns last dummy id */
mtg_walks *mtg_euler_cycles(const mtg_graph *g);
/* The same walk over the record formats the device finish feeds it, with the records built on the host from the
 * graph's adjacency lists: 1 = 32-byte records (euler_lean.cpp), 2 = 256-byte records seeded from 32-byte ones. Same sequences as
 * mtg_euler_cycles (format 0); exported for the CPU test suite and the sanitizer builds. */
mtg_walks *mtg_euler_cycles_records(const mtg_graph *g, int record_format);
mtg_walks *mtg_cut_cycles(const mtg_graph *g, const mtg_walks *cycles, uint64_t k);
/* the Euler bicycles computed on the GPU by a parallel algorithm (trail pairing, union-find merge,
 * list ranking). Same guarantees as the reference's decomposition (greedytigs/mod.rs:722: one closed biwalk per
 * connected component, every biedge exactly once in one orientation) but NOT the reference's walk order, so tig
 * boundaries differ while #tigs and cumulative length stay the same (SURVEY 8a invariance note). Opt-in:
 * mode 0 (default) = host walk in the reference's order, mode 1 = this. Aborts without a GPU. */
mtg_walks *mtg_euler_cycles_device(const mtg_graph *g, int device_id);
double mtg_last_euler_kernel_ms(void); /* of the last device decomposition on this thread */
/* The decomposition's segment walks recognise a splitter by a mark in bit 31 of its predecessor's successor word while dart ids
 * leave that bit free (fewer than 2^31 darts), else by a bitmap lookup per step. Bit 0 of `flags` forces the bitmap form, bit 1 ranks
 * the reduced list by pointer jumping over ALL splitters instead of two levels, bit 2 writes the closed walks by a second walk through
 * the successor array instead of from the sequence the measuring walk recorded, bit 3 shrinks the per-wave chunk tables of that
 * record to one entry, so that large graphs take the fallback to the second walk (process-wide; tests hold the forms to the same walks). */
void mtg_set_euler_device_tuning(int flags);
/* The whole finish on the GPU for a graph that holds only its original edges (finish_device.hip): matched-pair darts, the
 * Euleriser in the reference's sequence (implementation/mod.rs:392-649), Euler bicycles per cfg->euler_mode (device
 * decomposition, or the reference-order host walk over GPU-built records), rotate + cut (greedytigs/mod.rs:726-789). Appends the
 * dummy edges to g like the host stages do (same ids, weights and dummy ids). n_pairs == 0 is the Eulertig finish. */
mtg_walks *mtg_finish_device(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg);
/* mtg_finish_greedytigs_cfg with the matched pairs the last claim replay on `d` left in HBM (mtg_replay_claims_resident): when the
 * finish runs on d's GPU (cfg->device_ids[0], finish_stage AUTO / DEVICE, a graph without dummy edges) nothing is uploaded and
 * only the dummy weights come down, beside the GPU stages; otherwise the pairs take the way through the host. Same tigs either way. */
mtg_walks *mtg_finish_greedytigs_resident(mtg_graph *g, mtg_device *d, const mtg_config *cfg);
/* Device memory the library keeps BETWEEN calls, and how to get it back:
 *  - the finishing stages keep the work arrays of the last call on a GPU for the next one (a caller that finishes graph after graph
 *    pays no allocation); a call frees what it did not touch itself, so at most one call's arrays are held, and calls that worked
 *    on more than a third of the device's memory keep nothing. mtg_device_memory_held tells how much that is right now, mtg_release_device_memory gives
 *    all of it back to the driver (e.g. before another allocator of the process needs the HBM);
 *  - a graph keeps the device copy of its original edges, its mirror array and the buckets of its original darts on the GPU of
 *    its first device stage (<= 8 B per edge + 8 B per node) until mtg_graph_free / mtg_graph_release_device_cache.
 * Neither changes a result; the next call simply allocates / uploads again. */
void mtg_release_device_memory(int device_id);
uint64_t mtg_device_memory_held(int device_id);
/* The arena of that GPU in numbers: out[0] bytes in chunks taken from the driver, [1] bytes of live arrays, [2] their peak since the last
 * reset, [3] driver allocations made so far. reset_peak != 0: the peak restarts at what is live now (measurements: what a stage needs). */
void mtg_device_arena_stats(int device_id, uint64_t out[4], int reset_peak);
void mtg_graph_release_device_cache(mtg_graph *g);
/* The GPU whose memory a graph's construction reserves ahead of the call that will follow (default 0): building a graph of more
 * than a few million edges (mtg_graph_from_edges, matchtigs_initialise_graph, mtg_synth_g_csr on its own device) starts a helper
 * thread that brings up the HIP runtime, loads the kernels and takes ONE chunk of device memory sized for the whole call from
 * (nodes, edges) -- every device array of the call is then a range of it (hip_util.hpp: DeviceArena), so no stage waits for the
 * driver. A process that runs one rank per GPU names its GPU here before it builds graphs. Never changes a result; without a GPU
 * nothing happens. */
void mtg_set_default_device(int device_id);
/* on = 0: host-only graph constructors (mtg_graph_from_edges, matchtigs_initialise_graph) start no helper thread and reserve nothing
 * on any GPU -- for callers that want them free of GPU side effects; the first computing call then reserves for itself. Default 1.
 * Either way a reservation is provisional: a call that computes on other GPUs than the one it was made on gives it back when it
 * ends, and the helper threads are joined (each new one joins those that are through, the library's teardown joins the rest). */
void mtg_set_reserve_ahead(int on);
/* Tuning of the finishing stages for measurements and tests (process-wide, read at the start of a call; the library reads no
 * environment variable for any of it, and none changes a result). records: walk-record format of the reference-order mode, 0 = the
 * engine's choice by size and host memory, 1 = 32-byte, 2 = 128-byte, 3 = 256-byte records. flags: bit 0 = the walk
 * waits until all of its records have arrived (instead of starting on the 32-byte ones), bit 1 = never page-lock the record arena,
 * bit 2 = keep nothing of a graph on the device between calls (edges, mirror, buckets), bit 3 = a trivial kernel every 2 ms while the
 * host walks in the reference's order (measurement: what the GPU's idle state costs the first kernels of the next step), bit 4 = the device
 * Euler mode builds the closed walks of every component even where the tigs can be cut straight from the pairing (cut_first_device.hip; tests and
 * A/B measurements -- the tigs of that mode then come in another order, which the mode leaves unspecified, with the same count and cumulative
 * length). record_delay_us: slows the arrival of the
 * records by that much per slice (tests: small graphs then take the 32-byte path for most of their steps). */
void mtg_set_finish_tuning(int records, int flags, int64_t record_delay_us);
/* Seconds of the last mtg_finish_device on this thread: [0] upload + insertion + Euleriser, [1] dummy edges into the host graph,
 * [2] Euler bicycles, [3] rotate + cut + tig download; [4] kernel ms of the device decomposition; [5] breaking biedges added. */
void mtg_last_finish_device_times(double out[6]);
/* GPU time (HIP events on the finish stream, ms) of the stages of the last mtg_finish_device on this thread, for the per-stage
 * rooflines of bench.py: [0] matched-pair darts + Euleriser kernels, [1] buckets + walk records (reference-order mode),
 * [2] Euler decomposition (device mode), [3] rotate + cut kernels (without the tig download); [4] darts after the finish,
 * [5] Euleriser units (missing in-edges). */
void mtg_last_finish_device_stage_ms(double out[6]);
/* The two finishing paths with an explicit configuration (euler_mode, finish_stage, device_ids[0]). */
mtg_walks *mtg_finish_greedytigs_cfg(mtg_graph *g, const mtg_pair *pairs, uint64_t n_pairs, const mtg_config *cfg);
mtg_walks *mtg_compute_eulertigs_cfg(mtg_graph *g, const mtg_config *cfg);

/* A walk set. The tigs of a finish on the GPU (finish_stage auto / device) stay in that GPU's memory until a call needs them on the
 * host: the two counts below never copy; mtg_walks_export, mtg_walks_data, mtg_flatten_clib, the duplication bit vectors and the host
 * writers bring them over once (0.37 GB at the human-like bench size, through the pinned transfer ring) and release the device
 * arrays; mtg_write_tigs_text_file_device spells them in place when asked for the same GPU; mtg_walks_free of an unread handle copies
 * nothing. One thread per handle. */
uint64_t mtg_walks_count(const mtg_walks *w);
uint64_t mtg_walks_total_edges(const mtg_walks *w);
/* limits[i] = exclusive end of walk i in edges[] (edge ids into the mutated graph). */
void mtg_walks_export(const mtg_walks *w, uint64_t *limits, uint32_t *edges);
/* The walks' own arrays (no copy): valid until mtg_walks_free. */
void mtg_walks_data(const mtg_walks *w, const uint64_t **limits, const uint32_t **edges);
void mtg_walks_free(mtg_walks *w);
/* Walks from caller arrays (copied), e.g. externally computed Euler cycles for mtg_cut_cycles. */
mtg_walks *mtg_walks_from_arrays(uint64_t n_walks, const uint64_t *limits, const uint32_t *edges);

/* clib.rs:393-407 flattening into caller arrays sized as clib.rs:332-348. Returns #tigs. */
uint64_t mtg_flatten_clib(const mtg_graph *g, const mtg_walks *tigs, int64_t *tigs_edge_out,
                          uint64_t *tigs_insert_out, uint64_t *tigs_out_limits);

/* Tig spelling (replaces write_walks_fasta, bin.rs:466-606): walks (edge ids into the mutated graph,
 * limits[i] = exclusive end of walk i) -> FASTA text. unitig_seqs = concatenated ASCII sequences, unitig u occupies
 * [seq_offsets[u], seq_offsets[u+1]). Header ">{i+1}"; overlaps k-1 after a unitig edge, k-1-weight after a dummy
 * edge; backwards edges are reverse-complemented. Returns the byte count; *fasta_out is malloc'd (mtg_free). */
uint64_t mtg_write_walks_fasta(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges,
                               uint64_t k, const char *unitig_seqs, const uint64_t *seq_offsets, char **fasta_out);

/* BCALM2 / GGCAT unitig FASTA input (the `--bcalm-in X -k K` route, bin.rs:902-912): every
 * `L:<strand>:<id>:<strand>` annotation becomes one merge of unitig ends exactly as matchtigs_merge_nodes does
 * (clib.rs:135-170), weights are len + 1 - k (bin.rs:369). `.gz` files are inflated. Returns the built graph and the
 * sequence store (free with mtg_unitigs_free). Aborts on malformed input, non-ACGT characters, sequences shorter than k. */
typedef struct mtg_unitigs mtg_unitigs;
mtg_graph *mtg_read_bcalm2(const char *path, uint64_t k, mtg_unitigs **unitigs_out);
uint64_t mtg_unitigs_count(const mtg_unitigs *u);
const char *mtg_unitigs_data(const mtg_unitigs *u);        /* concatenated ASCII sequences */
const uint64_t *mtg_unitigs_offsets(const mtg_unitigs *u); /* count + 1 offsets into data */
void mtg_unitigs_free(mtg_unitigs *u);
/* Spells `tigs` (mtg_write_walks_fasta) into a file; a ".gz" suffix selects gzip at compression_level 0-9
 * (bin.rs:203, :442-446; pass 6 for the reference's default). Returns the uncompressed byte count. */
uint64_t mtg_write_tigs_fasta_file(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs,
                                   const char *path, int compression_level);
/* The same tigs as GFA1 text (bin.rs:667-818): header line (`header`, e.g. the one read from a GFA input, or NULL for
 * "H\tKL:Z:{k}"), then one "S\t{i+1}\t{sequence}" record per tig. */
uint64_t mtg_write_walks_gfa(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                             const char *unitig_seqs, const uint64_t *seq_offsets, const char *header, char **gfa_out);
uint64_t mtg_write_tigs_gfa_file(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs,
                                 const char *header, const char *path, int compression_level);
/* Both file writers with the GPU to spell on named explicitly (mtg_config.device_ids[0] of the run that made the tigs): gfa == 0
 * writes FASTA, else GFA with `gfa_header` or "H\tKL:Z:{k}". The two functions above spell on GPU 0. */
uint64_t mtg_write_tigs_text_file_device(const mtg_graph *g, const mtg_walks *tigs, uint64_t k, const mtg_unitigs *unitigs, int gfa,
                                         const char *gfa_header, const char *path, int compression_level, int device_id);
/* The same text (FASTA when gfa == 0, else GFA with `gfa_header` or "H\tKL:Z:{k}") spelled ON THE GPU `device_id`: the unitig
 * store is packed to 2 bits per base on the device and one kernel writes the characters at prefix-summed offsets. Byte-identical to
 * mtg_write_walks_fasta / _gfa for upper-case ACGT input (lower case is accepted and written upper case; any other character
 * aborts, like the reference's DnaAlphabet store). mtg_write_tigs_fasta_file / _gfa_file use this path whenever a GPU is present. */
uint64_t mtg_write_walks_text_device(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                                     const char *unitig_seqs, const uint64_t *seq_offsets, int gfa, const char *gfa_header, int device_id,
                                     char **text_out);
double mtg_last_spell_kernel_ms(void); /* HIP-event time of the last spelling kernel on this thread */
uint64_t mtg_last_spell_bytes(void);   /* HBM bytes it moved (text written + packed bases + per-position metadata) */
/* Duplication bitvectors (implementation/mod.rs:668-702): one line per tig with `weight` characters per edge, '1' for an
 * original edge and '0' for a dummy edge (k-mers that repeat ones spelled elsewhere). */
uint64_t mtg_write_duplication_bitvector(const mtg_graph *g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges,
                                         char **text_out);
uint64_t mtg_write_tigs_duplication_bitvector_file(const mtg_graph *g, const mtg_walks *tigs, const char *path);

/* ---- synthetic input on the GPU (bench / test infrastructure; numpy twin: matchtigs_amd/synth.py g_csr) ---------------- */
/* G-csr(n_binodes mirror pairs + n_self_mirrors self-mirror nodes; n_unitigs candidate unitigs with uniform endpoints from the
 * counter-based splitmix64 streams of `seed`; a unitig whose directed edges would exceed max_degree at their from-nodes is
 * dropped; weight = 1 + #{j : x <= weight_thresholds[j]} with x the 53-bit uniform of stream 3 and the thresholds descending,
 * i.e. a clipped geometric distribution in integer form). Returns the built graph. Aborts without a GPU. */
mtg_graph *mtg_synth_g_csr(uint64_t n_binodes, uint64_t n_self_mirrors, uint64_t n_unitigs, uint64_t seed, uint64_t k,
                           const uint64_t *weight_thresholds, uint64_t n_thresholds, int max_degree, int device_id);

/* Whole path: algorithm 1, 3 or 5 (clib.rs ids). Mutates g. matchtigs_compute_tigs builds the configuration from
 * clib.rs:378-389's constants and calls this. */
mtg_walks *mtg_compute_tigs_cfg(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg);
/* The whole path into a caller's clib.rs output arrays (clib.rs:332-348: 2 E, 2 E and E entries for E original edges; layout as
 * mtg_flatten_clib / clib.rs:393-407): mtg_compute_tigs_cfg + mtg_flatten_clib in one call -- what matchtigs_compute_tigs runs.
 * With a finish on the GPU (the default) the tigs never exist as walks on the host: the threads that empty the download ring write
 * the flattened form straight into the caller's arrays. Mutates g like mtg_compute_tigs_cfg. Returns the number of tigs. */
uint64_t mtg_compute_tigs_clib(mtg_graph *g, uint64_t tig_algorithm, const mtg_config *cfg, int64_t *tigs_edge_out,
                               uint64_t *tigs_insert_out, uint64_t *tigs_out_limits);
/* The same with mtg_config_init(threads = 1, k) on GPU `device_id`. */
mtg_walks *mtg_compute_tigs(mtg_graph *g, uint64_t tig_algorithm, uint64_t k, int device_id);
/* Counters of the last mtg_compute_tigs_cfg(5) on this thread run with MTG_PERFORMANCE_DATA_COMPLETE (zeros otherwise). */
void mtg_last_performance_data(mtg_dijkstra_performance_data *out);

/* Phase timings (seconds) of the last mtg_compute_tigs on this thread:
 * [0] device build+upload, [1] classify, [2] sssp (all levels; with several GPUs + the gather), [3] download of the matched pairs
 * (0 when they stay in HBM for a finish on the same GPU), [4] claim replay,
 * [5] dummy insertion + Euleriser, [6] Euler decomposition, [7] cut. */
void mtg_last_phase_seconds(double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* MTG_ENGINE_H */
