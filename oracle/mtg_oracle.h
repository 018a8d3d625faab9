/*
 * mtg_oracle.h -- CPU restatement of the reference's greedy-matchtigs / eulertigs path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under matchtigs_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (algbio/matchtigs 2.1.9, Rust)
 * cannot be built here (no cargo/rustc, crates not vendored) and its only test
 * (src/implementation/mod.rs:762-785) asserts nothing.  The arithmetic of the path
 * lives in third-party crates that are absent from /root/reference:
 *   traitgraph-algo 8.1.2 (Dijkstra), bigraph 5.0.1 (Euler / imbalance / mirror
 *   edges), traitgraph 8.1.2 + petgraph 0.7.1 (container, adjacency order),
 *   disjoint-sets 0.4.2 (union-find).  Their published algorithms are restated
 *   below, each behind one named policy function, and anchored on the reference's
 *   own call sites (cited per function).
 *
 * All citations are file:line into /root/reference/.
 */
#ifndef MTG_ORACLE_H
#define MTG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_NONE 0xFFFFFFFFu

typedef struct og_graph og_graph;

/* ---- graph container (policy: petgraph 0.7.1 `Graph` + bigraph NodeBigraphWrapper) ---- */
og_graph *og_graph_new(uint32_t n_nodes);
void og_graph_free(og_graph *g);
uint32_t og_add_node(og_graph *g);
/* bigraph set_mirror_nodes(a, b): mirror[a] = b, mirror[b] = a (clib.rs:236-237). */
void og_set_mirror_nodes(og_graph *g, uint32_t a, uint32_t b);
/* add_edge: edge ids are dense in insertion order (clib.rs:355 relies on it). */
uint32_t og_add_edge(og_graph *g, uint32_t from, uint32_t to, uint64_t weight,
                     uint64_t dummy_id, uint64_t handle, int forwards);
/* Bulk constructor for benchmarks: original edges in id order (edge 2u forwards of unitig u, 2u+1 its mirror). */
og_graph *og_graph_from_arrays(uint32_t n_nodes, const uint32_t *mirror, uint32_t n_edges, const uint32_t *from,
                               const uint32_t *to, const uint64_t *weight);
unsigned og_policies(void); /* include/mtg_policy.h: bit i = policy P(i+1) flipped in this build */
uint32_t og_node_count(const og_graph *g);
uint32_t og_edge_count(const og_graph *g);
uint32_t og_mirror_node(const og_graph *g, uint32_t n);
void og_edge_get(const og_graph *g, uint32_t e, uint32_t *from, uint32_t *to,
                 uint64_t *weight, uint64_t *dummy_id, uint64_t *handle, int *forwards);
/* out_neighbors(n) in petgraph iteration order = newest edge first. Returns count, writes edge ids. */
uint32_t og_out_edges(const og_graph *g, uint32_t n, uint32_t *edges_out, uint32_t cap);
/* bigraph mirror_edge_edge_centric: the edge mirror(to)->mirror(from) whose data == data.mirror(). */
uint32_t og_mirror_edge(const og_graph *g, uint32_t e);
int og_verify_node_pairing(const og_graph *g);
int og_verify_edge_mirror_property(const og_graph *g);

/* ---- clib.rs graph builder (clib.rs:97-259) ---- */
typedef struct og_builder og_builder;
og_builder *og_builder_new(uint64_t unitig_amount);                         /* clib.rs:97-102 */
void og_builder_merge_nodes(og_builder *b, uint64_t unitig_a, int strand_a,
                            uint64_t unitig_b, int strand_b);
void og_builder_merge_nodes_many(og_builder *b, uint64_t n, const int64_t *links);               /* clib.rs:135-170 */
og_graph *og_builder_build(og_builder *b, const uint64_t *unitig_weights);  /* clib.rs:180-259; frees b */

/* ---- bigraph::algo::eulerian restatements ---- */
int64_t og_superfluous_out_biedges(const og_graph *g, uint32_t n);
/* returns count; fills nodes[]/diffs[] (caller sizes them to node_count). */
uint32_t og_find_non_eulerian(const og_graph *g, uint32_t *nodes, int64_t *diffs);

/* ---- Dijkstra (traitgraph-algo 8.1.2 shortest_path_lens), call site greedytigs/mod.rs:324-335 ---- */
typedef struct {
    uint64_t iterations;        /* heap pops */
    uint64_t unnecessary;       /* stale pops */
    uint64_t settled_nodes;     /* nodes whose out-edges were relaxed */
    uint64_t relaxed_edges;     /* out-edges examined ("SSSP edges") */
    uint64_t queries;           /* shortest_path_lens calls */
} og_sssp_stats;

/* ---- greedy pair computation (greedytigs/mod.rs:222-526, 1-thread order) ---- */
typedef struct {
    uint32_t out_node, in_node;
    uint64_t distance;
} og_pair;

/* Classification greedytigs/mod.rs:229-245. out_nodes sized node_count; live sized node_count bytes; mult sized node_count. Returns #out_nodes. */
uint32_t og_classify(const og_graph *g, uint32_t *out_nodes, uint8_t *live, int64_t *mult,
                     uint32_t *in_node_count, uint32_t *self_mirror_count);

/* Runs the claim loop over all out-nodes in ascending order. Returns #pairs; *pairs malloc'd (caller frees with og_free). */
uint64_t og_greedy_pairs(const og_graph *g, uint64_t k, og_pair **pairs, og_sssp_stats *stats);
/* Same but only the first `max_sources` out-nodes (bench cpu_baseline sample). */
uint64_t og_greedy_pairs_prefix(const og_graph *g, uint64_t k, uint64_t max_sources,
                                og_pair **pairs, og_sssp_stats *stats);

/* The same claim loop over a classification the caller supplies (copied) instead of og_classify's: g may then be the subgraph that the
 * searches of a prefix of a LARGER graph's sources can reach, with that graph's classification restricted to g's nodes -- how the tests
 * pin the GPU pair list's prefix at sizes the oracle cannot hold (tests/gpu_props.py). */
uint64_t og_greedy_pairs_given(const og_graph *g, uint64_t k, const uint32_t *out_nodes, uint64_t n_out, const uint8_t *live,
                               const int64_t *mult, og_pair **pairs, og_sssp_stats *stats);

/* The same claim loop run by `threads` workers with the reference's locking scheme (greedytigs/mod.rs:528-644).
 * Thread-timing dependent like the reference with threads > 1: for the bench's multi-core timing only. */
uint64_t og_greedy_pairs_mt(const og_graph *g, uint64_t k, uint32_t threads, og_pair **pairs, og_sssp_stats *stats);
uint64_t og_greedy_pairs_mt_prefix(const og_graph *g, uint64_t k, uint32_t threads, uint64_t max_sources, og_pair **pairs,
                                   og_sssp_stats *stats);

/* Full candidate lists L(s): every initial in-node within k-1 of s (s excluded), in (dist,node)
 * order; CSR-style output: offsets[n_out+1] (malloc'd), keys = dist<<32|node (malloc'd). */
uint32_t og_candidate_lists(const og_graph *g, uint64_t k, uint32_t **out_nodes,
                            uint64_t **offsets, uint64_t **keys, og_sssp_stats *stats);

/* Same, but only sources with index in [src_lo, src_hi) are searched (the others get empty lists). */
uint32_t og_candidate_lists_range(const og_graph *g, uint64_t k, uint32_t src_lo, uint32_t src_hi, uint32_t **out_nodes,
                                  uint64_t **offsets, uint64_t **keys, og_sssp_stats *stats);

/* Full lists of the given sources under a live map the caller supplies (see og_greedy_pairs_given): offsets[n_sources + 1]. */
void og_candidate_lists_given(const og_graph *g, uint64_t k, const uint32_t *sources, uint64_t n_sources, const uint8_t *live,
                              uint64_t **offsets, uint64_t **keys, og_sssp_stats *stats);

void og_free(void *p);

/* ---- later stages ---- */
/* greedytigs/mod.rs:678-689: returns the final dummy_edge_id (= n_pairs). */
uint64_t og_insert_pair_edges(og_graph *g, const og_pair *pairs, uint64_t n_pairs);
/* implementation/mod.rs:392-649. */
void og_make_eulerian_with_breaking_edges(og_graph *g, uint64_t *dummy_edge_id, uint64_t k);
/* bigraph decomposes_into_eulerian_bicycles. */
int og_decomposes_into_eulerian_bicycles(const og_graph *g);
/* debug_assert_graph_has_no_consecutive_dummy_edges (implementation/mod.rs:319-390): 1 = invariant holds. */
int og_no_consecutive_dummy_edges(const og_graph *g, uint64_t k);

/* walks in flat form: limits[i] = exclusive end of walk i in edges[]. */
typedef struct {
    uint64_t n_walks;
    uint64_t n_edges;
    uint64_t *limits;
    uint32_t *edges;
} og_walks;
void og_walks_free(og_walks *w);

/* bigraph compute_minimum_bidirected_eulerian_cycle_decomposition (call greedytigs/mod.rs:722). */
og_walks *og_euler_cycles(const og_graph *g);
/* rotate + cut, greedytigs/mod.rs:726-789 == eulertigs/mod.rs:123-186. */
og_walks *og_cut_cycles(const og_graph *g, og_walks *cycles, uint64_t k, uint64_t *removed_edges);

/* Whole algorithms (mutate g by adding dummy edges, as the reference does). */
og_walks *og_compute_greedytigs(og_graph *g, uint64_t k, og_sssp_stats *stats);   /* greedytigs/mod.rs:201-801 */
og_walks *og_compute_eulertigs(og_graph *g, uint64_t k);                          /* eulertigs/mod.rs:48-198 */

/* ---- optimal matchtigs (matchtigs/mod.rs:150-940), everything but the external matcher itself ---- */
typedef struct og_matching og_matching;
/* :150-600, threads == 1: all (out-node, in-node) pairs within k-1, collapsed matching node ids, the edge map, WCC extra nodes */
og_matching *og_matching_instance(const og_graph *g, uint64_t k);
/* transformed_node_count, edges.len(), relevant WCCs, matching_node_count, matching_edge_count, mirror biedges, expanded mirror biedges */
void og_matching_counts(const og_matching *m, uint64_t out[7]);
/* :591-719: the `<prefix>.minimalperfectmatching` text */
void og_matching_write(const og_matching *m, const char *path);
/* :746-940: applies the matcher's `.solution` file to g (mutated), returns the matchtigs */
og_walks *og_matching_apply(const og_matching *m, og_graph *g, const char *solution_path, uint64_t k);
void og_matching_free(og_matching *m);

/* clib.rs:393-407 flattening. Arrays caller-allocated as clib.rs:332-348 says. Returns #tigs. */
uint64_t og_flatten_clib(const og_graph *g, const og_walks *tigs, int64_t *tigs_edge_out,
                         uint64_t *tigs_insert_out, uint64_t *tigs_out_limits);

/* clib.rs:280-410 in one call on a built graph: algorithm ids 1 (unitigs), 3 (eulertigs), 5 (greedy). */
uint64_t og_clib_compute_tigs(og_graph *g, uint64_t tig_algorithm, uint64_t k, int64_t *tigs_edge_out,
                              uint64_t *tigs_insert_out, uint64_t *tigs_out_limits);

/* ---- tig spelling (bin.rs:466-606), sequences as ASCII ACGT per unitig handle ---- */
/* seqs: concatenated unitig sequences; seq_off[handle]..seq_off[handle+1]. Returns malloc'd FASTA text, *len set. */
char *og_write_walks_fasta(const og_graph *g, const og_walks *tigs, const char *seqs,
                           const uint64_t *seq_off, uint64_t k, uint64_t *len);

#ifdef __cplusplus
}
#endif
#endif
