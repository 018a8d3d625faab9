/*
 * mtg_policy.h -- the five behaviours of the reference's path that are decided OUTSIDE its tree, named in one place.
 *
 * algbio/matchtigs gets its Dijkstra, its Euler decomposition, its graph container and its union-find from crates whose source is
 * not under /root/reference (traitgraph-algo 8.1.2, bigraph 5.0.1, traitgraph 8.1.2 over petgraph 0.7.1, disjoint-sets 0.4.2:
 * Cargo.lock:1260, :104, :1248 / :842, :412). Five of their choices decide BYTES of the result -- which pair is claimed, which walk
 * order comes out, which node gets which number -- and were restated from the crates' published behaviour, not
 * read from source: parity with the real binary is unpinned exactly there. Each of them is therefore ONE named
 * switch, consulted by every party through the functions below -- the product's kernels and host stages (matchtigs_amd/csrc), the C
 * oracle (oracle/mtg_oracle.c) and the Python restatement (tests/pyref.py reads the same switches through mtg_policies()) -- so
 * that, should a recollection prove wrong, one definition changes; and so that the parity suite can run under the OTHER setting of
 * every switch too (`make flipped` builds libmatchtigs_flipped.so / libmtg_oracle_flipped.so with all five flipped; the tiny-graph
 * fuzz holds oracle == restatement == product under both settings, on the CPU and through the HIP path:
 * tests/test_fuzz_small.py).
 *
 * Plain C (the oracle is C), usable from HIP device code. Default = 0 everywhere = the behaviour SURVEY App. A describes.
 */
#ifndef MTG_POLICY_H
#define MTG_POLICY_H

#include <stdint.h>

/* P1 -- Dijkstra's pop order among equal distances (traitgraph-algo: BinaryHeap<Reverse<(weight, node_index)>>).
 *   0: (distance, node index) ascending -- found targets come out in that order, and the claim loop takes the first `demand + 1`.
 *   1: (distance, node index DEscending). */
#ifndef MTG_POLICY_HEAP_TIE_DESCENDING
#define MTG_POLICY_HEAP_TIE_DESCENDING 0
#endif
/* P2 -- is the search bound inclusive? (traitgraph-algo: `if w > max_weight break`, called with max_weight = k - 1,
 * greedytigs/mod.rs:324-335)
 *   0: a target at distance exactly k - 1 is found.   1: it is not (largest accepted distance k - 2). */
#ifndef MTG_POLICY_BOUND_EXCLUSIVE
#define MTG_POLICY_BOUND_EXCLUSIVE 0
#endif
/* P3 -- iteration order of a node's out-edges (petgraph Graph: per-node singly linked edge lists, most recently added edge first,
 * SURVEY App. A.3); it decides which edge Hierholzer's walk (bigraph, App. A.2) takes first, hence the walk, hence the tigs.
 *   0: newest edge first (dummy edges, added last, are tried before original ones).   1: oldest edge first. */
#ifndef MTG_POLICY_ADJACENCY_OLDEST_FIRST
#define MTG_POLICY_ADJACENCY_OLDEST_FIRST 0
#endif
/* P4 -- union by rank with EQUAL ranks (disjoint-sets 0.4.2); it decides the representative of a merged set, hence
 * the node numbering clib.rs:193-234 derives from the sorted representatives, hence the greedy order.
 *   0: the first argument's root goes below the second's.   1: the second's below the first's. */
#ifndef MTG_POLICY_UNION_TIE_SECOND_UNDER_FIRST
#define MTG_POLICY_UNION_TIE_SECOND_UNDER_FIRST 0
#endif

/* P5 -- where Hierholzer's walk resumes once it is stuck at its start node (bigraph 5.0.1
 * compute_minimum_bidirected_eulerian_cycle_decomposition, calls greedytigs/mod.rs:722, eulertigs/mod.rs:119; SURVEY App. A.2 --
 * the one detail that appendix marks "not recalled with confidence"). Common to both settings: a biedge and its mirror are used
 * together; a bicycle starts at the lowest unused edge index; the walk always takes the first unused out-edge in adjacency order (P3);
 * the cycle built so far is rotate_left()-ed to the chosen edge and the next closed sub-walk is appended to it.
 *   0: the cycle is scanned from its FIRST edge forwards for the first edge whose from-node still has an unused out-edge
 *      (queue-like: sub-walks are inserted in the order the cycle visits their nodes).
 *   1: it is scanned from its LAST edge backwards for the last such edge (stack-like, the textbook backtracking Hierholzer:
 *      the walk resumes at the most recently visited node that still has an unused out-edge).
 * Either way the result is one closed walk per connected bi-component over the same biedges: tig COUNT and cumulative length do not
 * depend on it (SURVEY 8a's invariance note), the order of edges inside the walks -- hence the tig sequences -- does. */
#ifndef MTG_POLICY_EULER_SPLICE_LAST
#define MTG_POLICY_EULER_SPLICE_LAST 0
#endif

#if defined(__HIPCC__)
#define MTG_POLICY_FN __host__ __device__ static inline
#else
#define MTG_POLICY_FN static inline
#endif

/* bit i = switch Pi+1 is flipped (mtg_policies() / og_policies() report it for the library they live in) */
#define MTG_POLICY_MASK                                                                                        \
    ((MTG_POLICY_HEAP_TIE_DESCENDING ? 1u : 0u) | (MTG_POLICY_BOUND_EXCLUSIVE ? 2u : 0u) |                     \
     (MTG_POLICY_ADJACENCY_OLDEST_FIRST ? 4u : 0u) | (MTG_POLICY_UNION_TIE_SECOND_UNDER_FIRST ? 8u : 0u) |       \
     (MTG_POLICY_EULER_SPLICE_LAST ? 16u : 0u))

/* P1: candidate keys are (distance << 32 | node); a list is in pop order when mtg_policy_pop_rank(key) ascends */
MTG_POLICY_FN uint64_t mtg_policy_pop_rank(uint64_t key) { return MTG_POLICY_HEAP_TIE_DESCENDING ? key ^ 0xFFFFFFFFull : key; }
MTG_POLICY_FN int mtg_policy_pops_before(uint64_t key_a, uint64_t key_b) { return mtg_policy_pop_rank(key_a) < mtg_policy_pop_rank(key_b); }
/* P2: the largest distance at which a target is still found, for k-mer size k >= 2 (the call passes max_weight = k - 1) */
MTG_POLICY_FN uint64_t mtg_policy_search_bound(uint64_t k) { return MTG_POLICY_BOUND_EXCLUSIVE ? k - 2 : k - 1; }
/* P3: of a node's out-edges sorted by ascending edge id (= insertion order), position p of n in iteration order is index ... */
MTG_POLICY_FN uint32_t mtg_policy_adjacency_index(uint32_t p, uint32_t n) { return MTG_POLICY_ADJACENCY_OLDEST_FIRST ? p : n - 1u - p; }
/* P4: roots a (of the first argument) and b (of the second) have equal rank: the one that goes BELOW the other */
MTG_POLICY_FN int mtg_policy_union_tie_first_goes_below(void) { return MTG_POLICY_UNION_TIE_SECOND_UNDER_FIRST ? 0 : 1; }

/* P5: candidate positions of a cycle are examined first-to-last (0) or last-to-first (1) */
MTG_POLICY_FN int mtg_policy_euler_splice_last(void) { return MTG_POLICY_EULER_SPLICE_LAST ? 1 : 0; }

#endif /* MTG_POLICY_H */
