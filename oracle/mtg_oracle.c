/*
 * mtg_oracle.c -- plain-C CPU restatement of the reference's greedy-matchtigs / eulertigs path.
 *
 * TEST INFRASTRUCTURE ONLY (see mtg_oracle.h).  PARITY STATUS: parity unpinned (ibid.).
 *
 * It deliberately mirrors the reference's *sequential* structure (binary-heap Dijkstra that is
 * truncated at `target_amount`, re-run against a live target bitmap; literal Hierholzer with
 * rescans) so that the product (which is structured completely differently: full-ball candidate
 * lists on the GPU + replay, linked-list Hierholzer) is checked against an independent shape.
 *
 * Citations: file:line into /root/reference/.
 */
#include "../include/mtg_policy.h"
#include "mtg_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DIE(...)                                   \
    do {                                           \
        fprintf(stderr, "mtg_oracle: " __VA_ARGS__); \
        fprintf(stderr, "\n");                     \
        abort();                                   \
    } while (0)

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) DIE("out of memory (%zu bytes)", n);
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) DIE("out of memory");
    return p;
}
static void *xrealloc(void *p, size_t n) {
    p = realloc(p, n ? n : 1);
    if (!p) DIE("out of memory");
    return p;
}

/* ===================================================================================== */
/* Graph container.  Policy: petgraph 0.7.1 `Graph` keeps, per node, a  */
/* singly linked list of outgoing (and incoming) edges with the most recently added edge  */
/* first; traitgraph's PetGraph::out_neighbors iterates that list.  Node and edge indices */
/* are dense in insertion order.                                                          */
/* ===================================================================================== */
struct og_graph {
    uint32_t n_nodes, cap_nodes;
    uint32_t n_edges, cap_edges;
    uint32_t *mirror;   /* OG_NONE until set */
    uint32_t *head_out; /* newest outgoing edge of node, OG_NONE if none */
    uint32_t *head_in;
    uint32_t *out_deg;
    uint32_t *in_deg;
    /* per edge */
    uint32_t *from, *to;
    uint32_t *next_out; /* next (older) outgoing edge of `from` */
    uint32_t *next_in;
    uint64_t *weight;
    uint64_t *dummy_id; /* 0 = original (implementation/mod.rs:291-293) */
    uint64_t *handle;   /* unitig id / sequence handle */
    uint8_t *forwards;
};

static void grow_nodes(og_graph *g, uint32_t need) {
    if (need <= g->cap_nodes) return;
    uint32_t cap = g->cap_nodes ? g->cap_nodes : 16;
    while (cap < need) cap = cap < 0x7FFFFFFFu ? cap * 2 : 0xFFFFFFFEu;
    g->mirror = xrealloc(g->mirror, (size_t)cap * 4);
    g->head_out = xrealloc(g->head_out, (size_t)cap * 4);
    g->head_in = xrealloc(g->head_in, (size_t)cap * 4);
    g->out_deg = xrealloc(g->out_deg, (size_t)cap * 4);
    g->in_deg = xrealloc(g->in_deg, (size_t)cap * 4);
    g->cap_nodes = cap;
}
static void grow_edges(og_graph *g, uint32_t need) {
    if (need <= g->cap_edges) return;
    uint32_t cap = g->cap_edges ? g->cap_edges : 16;
    while (cap < need) cap = cap < 0x7FFFFFFFu ? cap * 2 : 0xFFFFFFFEu;
    g->from = xrealloc(g->from, (size_t)cap * 4);
    g->to = xrealloc(g->to, (size_t)cap * 4);
    g->next_out = xrealloc(g->next_out, (size_t)cap * 4);
    g->next_in = xrealloc(g->next_in, (size_t)cap * 4);
    g->weight = xrealloc(g->weight, (size_t)cap * 8);
    g->dummy_id = xrealloc(g->dummy_id, (size_t)cap * 8);
    g->handle = xrealloc(g->handle, (size_t)cap * 8);
    g->forwards = xrealloc(g->forwards, (size_t)cap);
    g->cap_edges = cap;
}

og_graph *og_graph_new(uint32_t n_nodes) {
    og_graph *g = xcalloc(1, sizeof *g);
    grow_nodes(g, n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        g->mirror[i] = OG_NONE;
        g->head_out[i] = g->head_in[i] = OG_NONE;
        g->out_deg[i] = g->in_deg[i] = 0;
    }
    g->n_nodes = n_nodes;
    return g;
}
void og_graph_free(og_graph *g) {
    if (!g) return;
    free(g->mirror); free(g->head_out); free(g->head_in); free(g->out_deg); free(g->in_deg);
    free(g->from); free(g->to); free(g->next_out); free(g->next_in);
    free(g->weight); free(g->dummy_id); free(g->handle); free(g->forwards);
    free(g);
}
uint32_t og_add_node(og_graph *g) {
    grow_nodes(g, g->n_nodes + 1);
    uint32_t n = g->n_nodes++;
    g->mirror[n] = OG_NONE;
    g->head_out[n] = g->head_in[n] = OG_NONE;
    g->out_deg[n] = g->in_deg[n] = 0;
    return n;
}
void og_set_mirror_nodes(og_graph *g, uint32_t a, uint32_t b) {
    if (a >= g->n_nodes || b >= g->n_nodes) DIE("set_mirror_nodes: node out of range");
    g->mirror[a] = b;
    g->mirror[b] = a;
}
uint32_t og_add_edge(og_graph *g, uint32_t from, uint32_t to, uint64_t weight, uint64_t dummy_id,
                     uint64_t handle, int forwards) {
    if (from >= g->n_nodes || to >= g->n_nodes) DIE("add_edge: node out of range");
    grow_edges(g, g->n_edges + 1);
    uint32_t e = g->n_edges++;
    g->from[e] = from; g->to[e] = to;
    g->weight[e] = weight; g->dummy_id[e] = dummy_id; g->handle[e] = handle;
    g->forwards[e] = forwards ? 1 : 0;
    if (!MTG_POLICY_ADJACENCY_OLDEST_FIRST) { /* policy P3 (include/mtg_policy.h): newest first */
        g->next_out[e] = g->head_out[from]; g->head_out[from] = e;
        g->next_in[e] = g->head_in[to]; g->head_in[to] = e;
    } else { /* the other setting: the new edge goes to the end of both lists (degrees are tiny where this build is used) */
        g->next_out[e] = OG_NONE;
        if (g->head_out[from] == OG_NONE) g->head_out[from] = e;
        else { uint32_t t = g->head_out[from]; while (g->next_out[t] != OG_NONE) t = g->next_out[t]; g->next_out[t] = e; }
        g->next_in[e] = OG_NONE;
        if (g->head_in[to] == OG_NONE) g->head_in[to] = e;
        else { uint32_t t = g->head_in[to]; while (g->next_in[t] != OG_NONE) t = g->next_in[t]; g->next_in[t] = e; }
    }
    g->out_deg[from]++; g->in_deg[to]++;
    return e;
}
og_graph *og_graph_from_arrays(uint32_t n_nodes, const uint32_t *mirror, uint32_t n_edges, const uint32_t *from,
                               const uint32_t *to, const uint64_t *weight) {
    og_graph *g = og_graph_new(n_nodes);
    for (uint32_t n = 0; n < n_nodes; n++) g->mirror[n] = mirror[n];
    grow_edges(g, n_edges);
    for (uint32_t e = 0; e < n_edges; e++) og_add_edge(g, from[e], to[e], weight[e], 0, e / 2, e % 2 == 0);
    return g;
}
unsigned og_policies(void) { return MTG_POLICY_MASK; } /* which setting of the four out-of-tree policies this build follows */
uint32_t og_node_count(const og_graph *g) { return g->n_nodes; }
uint32_t og_edge_count(const og_graph *g) { return g->n_edges; }
uint32_t og_mirror_node(const og_graph *g, uint32_t n) { return g->mirror[n]; }
void og_edge_get(const og_graph *g, uint32_t e, uint32_t *from, uint32_t *to, uint64_t *weight,
                 uint64_t *dummy_id, uint64_t *handle, int *forwards) {
    if (from) *from = g->from[e];
    if (to) *to = g->to[e];
    if (weight) *weight = g->weight[e];
    if (dummy_id) *dummy_id = g->dummy_id[e];
    if (handle) *handle = g->handle[e];
    if (forwards) *forwards = g->forwards[e];
}
uint32_t og_out_edges(const og_graph *g, uint32_t n, uint32_t *edges_out, uint32_t cap) {
    uint32_t c = 0;
    for (uint32_t e = g->head_out[n]; e != OG_NONE; e = g->next_out[e]) {
        if (c < cap) edges_out[c] = e;
        c++;
    }
    return c;
}

static inline int is_dummy(const og_graph *g, uint32_t e) { return g->dummy_id[e] != 0; }
static inline int is_self_mirror(const og_graph *g, uint32_t n) { return g->mirror[n] == n; }

/* Policy (App. A.3): bigraph NodeBigraphWrapper::mirror_edge_edge_centric(e) = among the edges
 * mirror(to) -> mirror(from), the one whose data equals data(e).mirror(); `mirror()` flips only
 * `forwards` (bin.rs:241-247, clib.rs:73-79), `Eq` compares all fields (clib.rs:45). Returns
 * OG_NONE if there is none; if several match, the first in adjacency order. */
uint32_t og_mirror_edge(const og_graph *g, uint32_t e) {
    uint32_t rf = g->mirror[g->to[e]], rt = g->mirror[g->from[e]];
    if (rf == OG_NONE || rt == OG_NONE) return OG_NONE;
    for (uint32_t m = g->head_out[rf]; m != OG_NONE; m = g->next_out[m]) {
        if (g->to[m] != rt) continue;
        if (g->weight[m] == g->weight[e] && g->dummy_id[m] == g->dummy_id[e] &&
            g->handle[m] == g->handle[e] && g->forwards[m] != g->forwards[e])
            return m;
    }
    return OG_NONE;
}
int og_verify_node_pairing(const og_graph *g) {
    for (uint32_t n = 0; n < g->n_nodes; n++) {
        uint32_t m = g->mirror[n];
        if (m == OG_NONE || m >= g->n_nodes || g->mirror[m] != n) return 0;
    }
    return 1;
}
int og_verify_edge_mirror_property(const og_graph *g) {
    for (uint32_t e = 0; e < g->n_edges; e++)
        if (og_mirror_edge(g, e) == OG_NONE) return 0;
    return 1;
}

/* ===================================================================================== */
/* Union-find.  Policy (App. A.4): disjoint-sets 0.4.2 `UnionFind`: union by rank; on equal */
/* ranks the first argument's root is attached under the second's, whose rank increments;   */
/* `find` with path halving (does not change which element is the root).                    */
/* ===================================================================================== */
struct og_builder {
    uint64_t n; /* 4 * unitig_amount */
    uint64_t *parent;
    uint8_t *rank;
};
static uint64_t uf_find(og_builder *b, uint64_t x) {
    uint64_t p = b->parent[x];
    while (x != p) {
        uint64_t gp = b->parent[p];
        b->parent[x] = gp;
        x = p;
        p = gp;
    }
    return x;
}
static void uf_union(og_builder *b, uint64_t x, uint64_t y) {
    uint64_t a = uf_find(b, x), c = uf_find(b, y);
    if (a == c) return;
    uint8_t ra = b->rank[a], rc = b->rank[c];
    if (ra > rc) b->parent[c] = a;
    else if (rc > ra) b->parent[a] = c;
    else if (mtg_policy_union_tie_first_goes_below()) { b->parent[a] = c; b->rank[c]++; } /* policy P4 (include/mtg_policy.h) */
    else { b->parent[c] = a; b->rank[a]++; }
}
og_builder *og_builder_new(uint64_t unitig_amount) { /* clib.rs:97-102 */
    og_builder *b = xmalloc(sizeof *b);
    b->n = unitig_amount * 4;
    b->parent = xmalloc(b->n * 8);
    b->rank = xcalloc(b->n, 1);
    for (uint64_t i = 0; i < b->n; i++) b->parent[i] = i;
    return b;
}
/* slot map clib.rs:104-122 */
static inline uint64_t fwd_in(uint64_t u) { return u * 4; }
static inline uint64_t fwd_out(uint64_t u) { return u * 4 + 2; }
static inline uint64_t bwd_in(uint64_t u) { return u * 4 + 3; }
static inline uint64_t bwd_out(uint64_t u) { return u * 4 + 1; }

void og_builder_merge_nodes(og_builder *b, uint64_t ua, int sa, uint64_t ub, int sb) { /* clib.rs:135-170 */
    uint64_t out_a = sa ? fwd_out(ua) : bwd_out(ua);
    uint64_t in_b = sb ? fwd_in(ub) : bwd_in(ub);
    uint64_t mirror_in_a = sa ? bwd_in(ua) : fwd_in(ua);
    uint64_t mirror_out_b = sb ? bwd_out(ub) : fwd_out(ub);
    if (out_a >= b->n || in_b >= b->n) DIE("merge_nodes: unitig id out of range");
    uf_union(b, out_a, in_b);
    uf_union(b, mirror_in_a, mirror_out_b);
}
/* the same, once per row (unitig_a, strand_a, unitig_b, strand_b) of an int64 table: test convenience only */
void og_builder_merge_nodes_many(og_builder *b, uint64_t n, const int64_t *links) {
    for (uint64_t i = 0; i < n; i++)
        og_builder_merge_nodes(b, (uint64_t)links[4 * i], links[4 * i + 1] != 0, (uint64_t)links[4 * i + 2], links[4 * i + 3] != 0);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static uint64_t bsearch_u64(const uint64_t *a, uint64_t n, uint64_t key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo >= n || a[lo] != key) DIE("representative not found");
    return lo;
}
og_graph *og_builder_build(og_builder *b, const uint64_t *unitig_weights) { /* clib.rs:180-259 */
    if (!unitig_weights) DIE("unitig_weights is null"); /* clib.rs:188 */
    uint64_t units = b->n / 4;
    uint64_t *reps = xmalloc(b->n * 8);
    for (uint64_t i = 0; i < b->n; i++) reps[i] = uf_find(b, i); /* force + to_vec, :193-194 */
    qsort(reps, b->n, 8, cmp_u64);                               /* :195 */
    uint64_t nr = 0;
    for (uint64_t i = 0; i < b->n; i++)
        if (i == 0 || reps[i] != reps[i - 1]) reps[nr++] = reps[i]; /* dedup :196 */
    if (nr >= OG_NONE) DIE("too many nodes for u32 ids");
    og_graph *g = og_graph_new((uint32_t)nr);                      /* :198-200 */
    for (uint64_t u = 0; u < units; u++) {                         /* :202-249 */
        uint32_t n1 = (uint32_t)bsearch_u64(reps, nr, uf_find(b, fwd_in(u)));
        uint32_t n2 = (uint32_t)bsearch_u64(reps, nr, uf_find(b, fwd_out(u)));
        uint32_t mn2 = (uint32_t)bsearch_u64(reps, nr, uf_find(b, bwd_in(u)));
        uint32_t mn1 = (uint32_t)bsearch_u64(reps, nr, uf_find(b, bwd_out(u)));
        og_set_mirror_nodes(g, n1, mn1);
        og_set_mirror_nodes(g, n2, mn2);
        og_add_edge(g, n1, n2, unitig_weights[u], 0, u, 1);
        og_add_edge(g, mn2, mn1, unitig_weights[u], 0, u, 0);
    }
    if (!og_verify_node_pairing(g)) DIE("assertion failed: graph.verify_node_pairing() (clib.rs:251)");
    if (!og_verify_edge_mirror_property(g)) DIE("assertion failed: graph.verify_edge_mirror_property() (clib.rs:252)");
    free(reps); free(b->parent); free(b->rank); free(b);
    return g;
}

/* ===================================================================================== */
/* bigraph 5.0.1 algo::eulerian (App. A.2)                                                */
/* ===================================================================================== */
/* compute_eulerian_superfluous_out_biedges: self-mirror -> out_degree % 2; else out - in.
 * Evidence: ranges asserted greedytigs/mod.rs:401-410; "diff > 0 => misses incoming" :237-244. */
int64_t og_superfluous_out_biedges(const og_graph *g, uint32_t n) {
    if (is_self_mirror(g, n)) return (int64_t)(g->out_deg[n] % 2);
    return (int64_t)g->out_deg[n] - (int64_t)g->in_deg[n];
}
/* find_non_eulerian_binodes_with_differences: ascending node order; self-mirror with odd
 * out-degree -> (n, 0); other with diff != 0 -> (n, diff).  Evidence: implementation/mod.rs:424-427
 * treats difference-0 entries as the self-mirrors. */
uint32_t og_find_non_eulerian(const og_graph *g, uint32_t *nodes, int64_t *diffs) {
    uint32_t c = 0;
    for (uint32_t n = 0; n < g->n_nodes; n++) {
        if (is_self_mirror(g, n)) {
            if (g->out_deg[n] % 2 != 0) { nodes[c] = n; diffs[c] = 0; c++; }
        } else {
            int64_t d = (int64_t)g->out_deg[n] - (int64_t)g->in_deg[n];
            if (d != 0) { nodes[c] = n; diffs[c] = d; c++; }
        }
    }
    return c;
}
int og_decomposes_into_eulerian_bicycles(const og_graph *g) {
    for (uint32_t n = 0; n < g->n_nodes; n++) {
        if (is_self_mirror(g, n)) { if (g->out_deg[n] % 2 != 0) return 0; }
        else if (g->out_deg[n] != g->in_deg[n]) return 0;
    }
    return 1;
}

/* ===================================================================================== */
/* Dijkstra.  Policy (App. A.1): traitgraph-algo 8.1.2 Dijkstra::shortest_path_lens with   */
/* BinaryHeap<Reverse<(weight, node)>> (pops in (distance, node index) order) and an epoch  */
/* array for node weights (clib.rs:386).  Lazy deletion; bound `weight > max_weight` breaks */
/* at pop time (inclusive bound); found targets are still expanded.                         */
/* ===================================================================================== */
typedef struct { uint64_t w; uint32_t n; } hitem;
typedef struct {
    hitem *heap; size_t hn, hcap;
    uint64_t *dist; uint32_t *epoch; uint32_t cur_epoch; uint32_t n_nodes;
    /* hash-map node weights (the CLI default NodeWeightArrayType, bin.rs:156: hashbrown) for the worker-thread variant,
     * where a V-sized array per thread would be page-fault bound: open addressing, cleared through the touched list */
    int hashed; uint32_t *mkey; uint64_t *mval; uint32_t mcap, mn; uint32_t *mtouched;
} dijkstra;

/* pop order among equal distances: policy P1 (include/mtg_policy.h) */
static inline int hless(hitem a, hitem b) { return a.w < b.w || (a.w == b.w && (MTG_POLICY_HEAP_TIE_DESCENDING ? a.n > b.n : a.n < b.n)); }
static void hpush(dijkstra *d, uint64_t w, uint32_t n) {
    if (d->hn == d->hcap) { d->hcap = d->hcap ? d->hcap * 2 : 64; d->heap = xrealloc(d->heap, d->hcap * sizeof(hitem)); }
    size_t i = d->hn++;
    hitem it = {w, n};
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!hless(it, d->heap[p])) break;
        d->heap[i] = d->heap[p]; i = p;
    }
    d->heap[i] = it;
}
static hitem hpop(dijkstra *d) {
    hitem top = d->heap[0];
    hitem last = d->heap[--d->hn];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, c;
        if (l >= d->hn) break;
        c = (r < d->hn && hless(d->heap[r], d->heap[l])) ? r : l;
        if (!hless(d->heap[c], last)) break;
        d->heap[i] = d->heap[c]; i = c;
    }
    if (d->hn) d->heap[i] = last;
    return top;
}
static dijkstra *dijkstra_new(uint32_t n_nodes) {
    dijkstra *d = xcalloc(1, sizeof *d);
    d->dist = xmalloc((size_t)n_nodes * 8);
    d->epoch = xcalloc(n_nodes, 4);
    d->cur_epoch = 1;
    d->n_nodes = n_nodes;
    return d;
}
static dijkstra *dijkstra_new_hashed(uint32_t n_nodes) {
    dijkstra *d = xcalloc(1, sizeof *d);
    d->hashed = 1; d->n_nodes = n_nodes; d->cur_epoch = 1;
    d->mcap = 1024; d->mkey = xmalloc((size_t)d->mcap * 4); d->mval = xmalloc((size_t)d->mcap * 8);
    d->mtouched = xmalloc((size_t)d->mcap * 4);
    memset(d->mkey, 0xFF, (size_t)d->mcap * 4);
    return d;
}
static void dijkstra_free(dijkstra *d) { free(d->heap); free(d->dist); free(d->epoch); free(d->mkey); free(d->mval); free(d->mtouched); free(d); }
static inline uint32_t mslot(const dijkstra *d, uint32_t n) {
    uint32_t h = (n * 0x9E3779B1u) & (d->mcap - 1);
    while (d->mkey[h] != OG_NONE && d->mkey[h] != n) h = (h + 1) & (d->mcap - 1);
    return h;
}
static void mgrow(dijkstra *d) {
    uint32_t ocap = d->mcap, on = d->mn; uint32_t *ok = d->mkey, *ot = d->mtouched; uint64_t *ov = d->mval;
    d->mcap = ocap * 2; d->mn = 0;
    d->mkey = xmalloc((size_t)d->mcap * 4); d->mval = xmalloc((size_t)d->mcap * 8); d->mtouched = xmalloc((size_t)d->mcap * 4);
    memset(d->mkey, 0xFF, (size_t)d->mcap * 4);
    for (uint32_t i = 0; i < on; i++) {
        uint32_t os = ot[i], h = mslot(d, ok[os]);
        d->mkey[h] = ok[os]; d->mval[h] = ov[os]; d->mtouched[d->mn++] = h;
    }
    free(ok); free(ov); free(ot);
}
static inline uint64_t dget(dijkstra *d, uint32_t n) {
    if (d->hashed) { uint32_t h = mslot(d, n); return d->mkey[h] == n ? d->mval[h] : UINT64_MAX; }
    return d->epoch[n] == d->cur_epoch ? d->dist[n] : UINT64_MAX;
}
static inline void dset(dijkstra *d, uint32_t n, uint64_t w) {
    if (d->hashed) {
        uint32_t h = mslot(d, n);
        if (d->mkey[h] != n) {
            if ((d->mn + 1) * 2 > d->mcap) { mgrow(d); h = mslot(d, n); }
            d->mkey[h] = n; d->mtouched[d->mn++] = h;
        }
        d->mval[h] = w;
        return;
    }
    d->epoch[n] = d->cur_epoch; d->dist[n] = w;
}

typedef struct { uint32_t node; uint64_t dist; } dist_entry;
typedef struct { dist_entry *v; size_t n, cap; } dist_vec;
static void dv_push(dist_vec *v, uint32_t node, uint64_t dist) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->v = xrealloc(v->v, v->cap * sizeof(dist_entry)); }
    v->v[v->n].node = node; v->v[v->n].dist = dist; v->n++;
}

/* shortest_path_lens(graph, source, targets, target_amount, max_weight, forbid_source_target,
 *                    distances, max_node_weight_data_size = max_heap_data_size = usize::MAX, perf)
 * Call site greedytigs/mod.rs:324-335 (limits are usize::MAX for threads == 1, :548-551, so the
 * result is always `Complete`). */
static void shortest_path_lens(const og_graph *g, dijkstra *d, uint32_t source, const uint8_t *targets,
                               uint64_t target_amount, uint64_t max_weight, int forbid_source_target,
                               dist_vec *distances, og_sssp_stats *st) {
    hpush(d, 0, source);
    dset(d, source, 0);
    distances->n = 0;
    if (st) st->queries++;
    while (d->hn) {
        hitem it = hpop(d);
        if (st) st->iterations++;
        uint64_t actual = dget(d, it.n);
        if (actual < it.w) { if (st) st->unnecessary++; continue; }
        if (MTG_POLICY_BOUND_EXCLUSIVE ? it.w >= max_weight : it.w > max_weight) break; /* inclusive bound: policy P2 (include/mtg_policy.h) */
        if (targets[it.n] && !(forbid_source_target && it.n == source)) {
            dv_push(distances, it.n, it.w);
            if (distances->n == target_amount) break;
        }
        if (st) st->settled_nodes++;
        for (uint32_t e = g->head_out[it.n]; e != OG_NONE; e = g->next_out[e]) {
            if (st) st->relaxed_edges++;
            uint64_t nw = it.w + g->weight[e];
            uint32_t nb = g->to[e];
            if (nw < dget(d, nb)) { dset(d, nb, nw); hpush(d, nw, nb); }
        }
    }
    d->hn = 0;       /* heap.clear() */
    if (d->hashed) {  /* node_weights.clear() */
        for (uint32_t i = 0; i < d->mn; i++) d->mkey[d->mtouched[i]] = OG_NONE;
        d->mn = 0;
        return;
    }
    d->cur_epoch++;  /* node_weights.clear() */
    if (d->cur_epoch == 0) { memset(d->epoch, 0, (size_t)d->n_nodes * 4); d->cur_epoch = 1; }
}

/* ===================================================================================== */
/* Classification, greedytigs/mod.rs:229-245 (== eulertigs/mod.rs:71-87)                  */
/* ===================================================================================== */
uint32_t og_classify(const og_graph *g, uint32_t *out_nodes, uint8_t *live, int64_t *mult,
                     uint32_t *in_node_count, uint32_t *self_mirror_count) {
    uint32_t no = 0, ni = 0, ns = 0;
    for (uint32_t n = 0; n < g->n_nodes; n++) {
        live[n] = 0; mult[n] = 0;
        int64_t diff = og_superfluous_out_biedges(g, n);
        if (is_self_mirror(g, n) && diff != 0) {
            ni++; live[n] = 1; mult[n] = diff; out_nodes[no++] = n; ns++;
        } else if (diff > 0) {
            ni++; live[n] = 1; mult[n] = diff;
        } else if (diff < 0) {
            out_nodes[no++] = n; mult[n] = diff;
        }
    }
    if (in_node_count) *in_node_count = ni;
    if (self_mirror_count) *self_mirror_count = ns;
    return no;
}

/* ===================================================================================== */
/* Claim loop, greedytigs/mod.rs:301-523 in its single-thread execution order.            */
/* `locks[0]` = mult[out], `locks[1]` = mult[mirror(out)], `locks[off]` = mult[in], ...     */
/* ===================================================================================== */
typedef struct { og_pair *v; uint64_t n, cap; } pair_vec;
static void pv_push(pair_vec *p, uint32_t o, uint32_t i, uint64_t d) {
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 64; p->v = xrealloc(p->v, p->cap * sizeof(og_pair)); }
    p->v[p->n].out_node = o; p->v[p->n].in_node = i; p->v[p->n].distance = d; p->n++;
}

/* the loop itself, over a classification the caller owns (out_nodes ascending; live / mult are updated in place) */
static uint64_t claim_loop(const og_graph *g, uint64_t k, const uint32_t *out_nodes, uint64_t limit, uint8_t *live, int64_t *mult,
                           og_pair **pairs, og_sssp_stats *stats) {
    uint32_t nn = g->n_nodes;
    if (stats) memset(stats, 0, sizeof *stats);
    dijkstra *dj = dijkstra_new(nn);
    dist_vec distances = {0};
    pair_vec res = {0};
    if (k == 0) DIE("k must be >= 1");

    for (uint64_t i = 0; i < limit; i++) {                         /* :301 */
        uint32_t out_node = out_nodes[i];
        int out_sm = is_self_mirror(g, out_node);                  /* :302 */
        uint32_t out_mirror = g->mirror[out_node];                 /* :303 */
        int64_t out_mult = mult[out_mirror];                       /* :306-311 */
        if (out_mult < 0 || out_mult > 4) DIE("out_node_multiplicity = %lld out of [0,4] (:313-316)", (long long)out_mult);
        if (out_mult == 0) continue;                               /* :318-320 */
        while (out_mult > 0) {                                     /* :322 */
            uint64_t target_amount = (uint64_t)(out_mult + 1);     /* :323 */
            shortest_path_lens(g, dj, out_node, live, target_amount, k - 1, 1, &distances, stats); /* :324-335 */
            if (distances.n == 0) break;                           /* :338-346 (status Complete) */
            int abort_after_this = distances.n < target_amount;    /* :348 */
            for (size_t c = 0; c < distances.n; c++) {             /* :350 */
                uint32_t in_node = distances.v[c].node;
                uint64_t dist = distances.v[c].dist;
                int self_edge = 0;
                if (in_node == out_mirror) {                       /* :352-358 */
                    if (out_mult < 2) continue;
                    self_edge = 1;
                }
                uint32_t in_mirror = g->mirror[in_node];           /* :362 */
                int in_sm = is_self_mirror(g, in_node);            /* :363 */
                int64_t red = self_edge ? 2 : 1;                   /* :399 */
                if (out_sm) {                                      /* :401-410 */
                    if (mult[out_node] < 0 || mult[out_node] > 1) DIE("self-mirror out multiplicity out of range");
                    out_mult = mult[out_node];
                } else {
                    if (mult[out_node] > 0 || mult[out_node] < -4 || mult[out_node] != -mult[out_mirror])
                        DIE("out multiplicity invariant violated (:406-408)");
                    out_mult = -mult[out_node];
                }
                if (out_mult == 0) break;                          /* :412-414 */
                if (!self_edge) {                                  /* :416-459 */
                    if (in_sm) { if (mult[in_node] < 0 || mult[in_node] > 1) DIE("self-mirror in multiplicity out of range"); }
                    else if (mult[in_node] < 0 || mult[in_node] > 4 || mult[in_node] != -mult[in_mirror])
                        DIE("in multiplicity invariant violated (:439-451)");
                    if (mult[in_node] == 0) { live[in_node] = 0; continue; } /* :454-458 */
                }
                pv_push(&res, out_node, in_node, dist);            /* :461 */
                if (out_sm) mult[out_node] -= 1;                   /* :463-466 */
                else { mult[out_node] += red; mult[out_mirror] -= red; } /* :467-473 */
                out_mult = -mult[out_node];                        /* :474 */
                if (!self_edge) {                                  /* :476-491 */
                    mult[in_node] -= 1;
                    if (!in_sm) mult[in_mirror] += 1;
                }
                if (out_mult == 0) live[out_mirror] = 0;           /* :493-495 */
                if (!self_edge && mult[in_node] == 0) live[in_node] = 0; /* :497-501 */
            }
            if (abort_after_this) break;                           /* :504-511 */
        }
    }
    dijkstra_free(dj);
    free(distances.v);
    *pairs = res.v;
    return res.n;
}
uint64_t og_greedy_pairs_prefix(const og_graph *g, uint64_t k, uint64_t max_sources, og_pair **pairs,
                                og_sssp_stats *stats) {
    uint32_t nn = g->n_nodes;
    uint32_t *out_nodes = xmalloc((size_t)nn * 4);
    uint8_t *live = xmalloc(nn);
    int64_t *mult = xmalloc((size_t)nn * 8);
    uint32_t n_out = og_classify(g, out_nodes, live, mult, NULL, NULL);
    uint64_t n = claim_loop(g, k, out_nodes, max_sources < n_out ? max_sources : n_out, live, mult, pairs, stats);
    free(out_nodes); free(live); free(mult);
    return n;
}
/* The same loop over a classification the CALLER supplies instead of og_classify's: for a graph too large for this oracle, g is the
 * subgraph the searches of a prefix of its sources can reach (every out-edge of every node within k - 1 of one of them, nodes
 * renumbered in order) and out_nodes / live / mult are the full graph's classification restricted to g's nodes -- computed from
 * the full graph's degrees, which the subgraph's boundary nodes do not have. The arrays are copied. */
uint64_t og_greedy_pairs_given(const og_graph *g, uint64_t k, const uint32_t *out_nodes, uint64_t n_out, const uint8_t *live_in,
                               const int64_t *mult_in, og_pair **pairs, og_sssp_stats *stats) {
    uint32_t nn = g->n_nodes;
    uint8_t *live = xmalloc(nn ? nn : 1);
    int64_t *mult = xmalloc(((size_t)nn + 1) * 8);
    memcpy(live, live_in, nn);
    memcpy(mult, mult_in, (size_t)nn * 8);
    uint64_t n = claim_loop(g, k, out_nodes, n_out, live, mult, pairs, stats);
    free(live); free(mult);
    return n;
}
uint64_t og_greedy_pairs(const og_graph *g, uint64_t k, og_pair **pairs, og_sssp_stats *stats) {
    return og_greedy_pairs_prefix(g, k, UINT64_MAX, pairs, stats);
}

/* ===================================================================================== */
/* The same claim loop run by `threads` workers, greedytigs/mod.rs:528-644 (no staging:    */
/* resource limits stay usize::MAX): workers take chunks of 1024 sources from a shared     */
/* cursor (:573-616 without the 5-second re-sizing), each with its own Dijkstra state; the  */
/* live bytes are relaxed atomics (implementation/mod.rs:128-186), every multiplicity cell  */
/* has its own lock and a claim takes {out, out', in, in'} in ascending order (:366-397).   */
/* Like the reference with threads > 1 the result depends on thread timing: this entry      */
/* exists for the bench's multi-core CPU timing and is checked by invariants only.          */
/* ===================================================================================== */
typedef struct {
    const og_graph *g; uint64_t k; uint32_t n_out; const uint32_t *out_nodes;
    uint8_t *live; int64_t *mult; uint8_t *locks; uint64_t *cursor;
    pair_vec res; og_sssp_stats st;
} mt_worker;

static void mt_lock(uint8_t *l) { while (__atomic_test_and_set(l, __ATOMIC_ACQUIRE)) { while (__atomic_load_n(l, __ATOMIC_RELAXED)) {} } }
static void mt_unlock(uint8_t *l) { __atomic_clear(l, __ATOMIC_RELEASE); }

static void *mt_worker_main(void *arg) {
    mt_worker *w = arg;
    const og_graph *g = w->g;
    dijkstra *dj = dijkstra_new_hashed(g->n_nodes);
    dist_vec distances = {0};
    for (;;) {
        uint64_t lo = __atomic_fetch_add(w->cursor, 1024, __ATOMIC_RELAXED);
        if (lo >= w->n_out) break;
        uint64_t hi = lo + 1024 < w->n_out ? lo + 1024 : w->n_out;
        for (uint64_t i = lo; i < hi; i++) {
            uint32_t out_node = w->out_nodes[i];
            int out_sm = is_self_mirror(g, out_node);
            uint32_t out_mirror = g->mirror[out_node];
            mt_lock(&w->locks[out_mirror]);
            int64_t out_mult = w->mult[out_mirror];                  /* :306-311 */
            mt_unlock(&w->locks[out_mirror]);
            while (out_mult > 0) {
                uint64_t target_amount = (uint64_t)(out_mult + 1);
                shortest_path_lens(g, dj, out_node, w->live, target_amount, w->k - 1, 1, &distances, &w->st);
                if (distances.n == 0) break;
                int abort_after_this = distances.n < target_amount;
                for (size_t c = 0; c < distances.n; c++) {
                    uint32_t in_node = distances.v[c].node;
                    uint64_t dist = distances.v[c].dist;
                    int self_edge = 0;
                    if (in_node == out_mirror) { if (out_mult < 2) continue; self_edge = 1; }
                    uint32_t in_mirror = g->mirror[in_node];
                    int in_sm = is_self_mirror(g, in_node);
                    uint32_t ls[4] = {out_node, out_mirror, in_node, in_mirror}; int nl = 4;   /* :366-397 */
                    for (int a = 1; a < nl; a++) for (int b = a; b > 0 && ls[b] < ls[b - 1]; b--) { uint32_t t = ls[b]; ls[b] = ls[b - 1]; ls[b - 1] = t; }
                    int u = 1; for (int a = 1; a < nl; a++) if (ls[a] != ls[u - 1]) ls[u++] = ls[a];
                    nl = u;
                    for (int a = 0; a < nl; a++) mt_lock(&w->locks[ls[a]]);
                    int64_t red = self_edge ? 2 : 1;
                    out_mult = out_sm ? w->mult[out_node] : -w->mult[out_node];
                    int claimed = 0, stop = 0;
                    if (out_mult == 0) stop = 1;                      /* :412-414 */
                    else if (!self_edge && w->mult[in_node] == 0) __atomic_store_n(&w->live[in_node], 0, __ATOMIC_RELAXED);  /* :454-458 */
                    else {
                        claimed = 1;
                        if (out_sm) w->mult[out_node] -= 1; else { w->mult[out_node] += red; w->mult[out_mirror] -= red; }
                        out_mult = -w->mult[out_node];
                        if (!self_edge) { w->mult[in_node] -= 1; if (!in_sm) w->mult[in_mirror] += 1; }
                        if (out_mult == 0) __atomic_store_n(&w->live[out_mirror], 0, __ATOMIC_RELAXED);
                        if (!self_edge && w->mult[in_node] == 0) __atomic_store_n(&w->live[in_node], 0, __ATOMIC_RELAXED);
                    }
                    for (int a = nl; a-- > 0;) mt_unlock(&w->locks[ls[a]]);
                    if (claimed) pv_push(&w->res, out_node, in_node, dist);
                    if (stop) break;
                }
                if (abort_after_this) break;
            }
        }
    }
    dijkstra_free(dj); free(distances.v);
    return NULL;
}

uint64_t og_greedy_pairs_mt(const og_graph *g, uint64_t k, uint32_t threads, og_pair **pairs, og_sssp_stats *stats) {
    return og_greedy_pairs_mt_prefix(g, k, threads, UINT64_MAX, pairs, stats);
}

/* The same over the first max_sources out-nodes only (bench.py: a bounded sample of a workload whose whole CPU run takes minutes). */
uint64_t og_greedy_pairs_mt_prefix(const og_graph *g, uint64_t k, uint32_t threads, uint64_t max_sources, og_pair **pairs,
                                   og_sssp_stats *stats) {
    uint32_t nn = g->n_nodes;
    if (threads < 1) threads = 1;
    uint32_t *out_nodes = xmalloc((size_t)nn * 4);
    uint8_t *live = xmalloc(nn);
    int64_t *mult = xmalloc((size_t)nn * 8);
    uint8_t *locks = xmalloc(nn ? nn : 1);
    memset(locks, 0, nn);
    uint32_t n_out = og_classify(g, out_nodes, live, mult, NULL, NULL);
    if ((uint64_t)n_out > max_sources) n_out = (uint32_t)max_sources;
    uint64_t cursor = 0;
    mt_worker *ws = xmalloc(sizeof(mt_worker) * threads);
    pthread_t *th = xmalloc(sizeof(pthread_t) * threads);
    for (uint32_t t = 0; t < threads; t++) {
        memset(&ws[t], 0, sizeof ws[t]);
        ws[t].g = g; ws[t].k = k; ws[t].n_out = n_out; ws[t].out_nodes = out_nodes;
        ws[t].live = live; ws[t].mult = mult; ws[t].locks = locks; ws[t].cursor = &cursor;
        if (pthread_create(&th[t], NULL, mt_worker_main, &ws[t])) DIE("pthread_create failed");
    }
    pair_vec res = {0};
    if (stats) memset(stats, 0, sizeof *stats);
    for (uint32_t t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        for (uint64_t i = 0; i < ws[t].res.n; i++) pv_push(&res, ws[t].res.v[i].out_node, ws[t].res.v[i].in_node, ws[t].res.v[i].distance);
        free(ws[t].res.v);
        if (stats) {
            stats->iterations += ws[t].st.iterations; stats->unnecessary += ws[t].st.unnecessary;
            stats->settled_nodes += ws[t].st.settled_nodes; stats->relaxed_edges += ws[t].st.relaxed_edges;
            stats->queries += ws[t].st.queries;
        }
    }
    free(ws); free(th); free(out_nodes); free(live); free(mult); free(locks);
    *pairs = res.v;
    return res.n;
}

/* Full candidate lists: one untruncated query per out-node against the *initial* bitmap. */
uint32_t og_candidate_lists(const og_graph *g, uint64_t k, uint32_t **out_nodes_p, uint64_t **offsets_p,
                            uint64_t **keys_p, og_sssp_stats *stats) {
    return og_candidate_lists_range(g, k, 0, UINT32_MAX, out_nodes_p, offsets_p, keys_p, stats);
}

uint32_t og_candidate_lists_range(const og_graph *g, uint64_t k, uint32_t src_lo, uint32_t src_hi, uint32_t **out_nodes_p,
                                  uint64_t **offsets_p, uint64_t **keys_p, og_sssp_stats *stats) {
    uint32_t nn = g->n_nodes;
    uint32_t *out_nodes = xmalloc((size_t)nn * 4);
    uint8_t *live = xmalloc(nn);
    int64_t *mult = xmalloc((size_t)nn * 8);
    uint32_t n_out = og_classify(g, out_nodes, live, mult, NULL, NULL);
    if (stats) memset(stats, 0, sizeof *stats);
    dijkstra *dj = dijkstra_new(nn);
    dist_vec distances = {0};
    uint64_t *offsets = xmalloc(((size_t)n_out + 1) * 8);
    uint64_t *keys = NULL; uint64_t nk = 0, capk = 0;
    for (uint32_t i = 0; i < n_out; i++) {
        offsets[i] = nk;
        if (i < src_lo || i >= src_hi) continue;  /* sources outside the range keep empty lists */
        shortest_path_lens(g, dj, out_nodes[i], live, UINT64_MAX, k - 1, 1, &distances, stats);
        for (size_t c = 0; c < distances.n; c++) {
            if (nk == capk) { capk = capk ? capk * 2 : 256; keys = xrealloc(keys, capk * 8); }
            keys[nk++] = (distances.v[c].dist << 32) | distances.v[c].node;
        }
    }
    offsets[n_out] = nk;
    dijkstra_free(dj); free(distances.v); free(live); free(mult);
    *out_nodes_p = out_nodes; *offsets_p = offsets; *keys_p = keys ? keys : xmalloc(8);
    return n_out;
}

/* Full lists L(s) of the given sources under a live map the caller supplies (see og_greedy_pairs_given): offsets[n_sources + 1]. */
void og_candidate_lists_given(const og_graph *g, uint64_t k, const uint32_t *sources, uint64_t n_sources, const uint8_t *live,
                              uint64_t **offsets_p, uint64_t **keys_p, og_sssp_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    dijkstra *dj = dijkstra_new(g->n_nodes);
    dist_vec distances = {0};
    uint64_t *offsets = xmalloc(((size_t)n_sources + 1) * 8);
    uint64_t *keys = NULL; uint64_t nk = 0, capk = 0;
    for (uint64_t i = 0; i < n_sources; i++) {
        offsets[i] = nk;
        shortest_path_lens(g, dj, sources[i], live, UINT64_MAX, k - 1, 1, &distances, stats);
        for (size_t c = 0; c < distances.n; c++) {
            if (nk == capk) { capk = capk ? capk * 2 : 256; keys = xrealloc(keys, capk * 8); }
            keys[nk++] = (distances.v[c].dist << 32) | distances.v[c].node;
        }
    }
    offsets[n_sources] = nk;
    dijkstra_free(dj); free(distances.v);
    *offsets_p = offsets; *keys_p = keys ? keys : xmalloc(8);
}

void og_free(void *p) { free(p); }

/* ===================================================================================== */
/* Dummy insertion, greedytigs/mod.rs:678-689                                             */
/* ===================================================================================== */
uint64_t og_insert_pair_edges(og_graph *g, const og_pair *pairs, uint64_t n_pairs) {
    uint64_t dummy_edge_id = 0;
    for (uint64_t i = 0; i < n_pairs; i++) {
        dummy_edge_id += 1;
        og_add_edge(g, pairs[i].out_node, pairs[i].in_node, pairs[i].distance, dummy_edge_id, 0, 1);
        og_add_edge(g, g->mirror[pairs[i].in_node], g->mirror[pairs[i].out_node], pairs[i].distance, dummy_edge_id, 0, 0);
    }
    return dummy_edge_id;
}

/* ===================================================================================== */
/* make_graph_eulerian_with_breaking_edges, implementation/mod.rs:392-649.                */
/* The two BTreeMaps only ever lose keys after construction, so they are sorted arrays    */
/* with tombstones here.                                                                  */
/* ===================================================================================== */
typedef struct { uint32_t *key; int64_t *val; uint8_t *alive; uint32_t n, first; int desc; } smap;
static int smap_find(const smap *m, uint32_t key) { /* index or -1 */
    uint32_t lo = 0, hi = m->n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        int before = m->desc ? (m->key[mid] > key) : (m->key[mid] < key);
        if (before) lo = mid + 1; else hi = mid;
    }
    if (lo < m->n && m->key[lo] == key && m->alive[lo]) return (int)lo;
    return -1;
}
static int smap_first(smap *m) {
    while (m->first < m->n && !m->alive[m->first]) m->first++;
    return m->first < m->n ? (int)m->first : -1;
}
static int smap_next(const smap *m, int i) {
    for (uint32_t j = (uint32_t)i + 1; j < m->n; j++) if (m->alive[j]) return (int)j;
    return -1;
}
static void smap_remove(smap *m, int i) { m->alive[i] = 0; }
static void smap_free(smap *m) { free(m->key); free(m->val); free(m->alive); }

static void add_breaking_biedge(og_graph *g, uint32_t out_node, uint32_t in_node, uint64_t *dummy_edge_id, uint64_t k) {
    uint32_t mirror_out_node = g->mirror[in_node];  /* naming as in :486-487 / :569-570 */
    uint32_t mirror_in_node = g->mirror[out_node];
    *dummy_edge_id += 1;
    og_add_edge(g, out_node, in_node, k, *dummy_edge_id, 0, 1);
    og_add_edge(g, mirror_out_node, mirror_in_node, k, *dummy_edge_id, 0, 0);
}

void og_make_eulerian_with_breaking_edges(og_graph *g, uint64_t *dummy_edge_id, uint64_t k) {
    uint32_t nn = g->n_nodes;
    uint32_t *nodes = xmalloc((size_t)nn * 4);
    int64_t *diffs = xmalloc((size_t)nn * 8);
    uint32_t cnt = og_find_non_eulerian(g, nodes, diffs);          /* :408 */
    smap outm = {0}, inm = {0};
    outm.desc = 1;
    outm.key = xmalloc((size_t)cnt * 4); outm.val = xmalloc((size_t)cnt * 8); outm.alive = xmalloc(cnt ? cnt : 1);
    inm.key = xmalloc((size_t)cnt * 4); inm.val = xmalloc((size_t)cnt * 8); inm.alive = xmalloc(cnt ? cnt : 1);
    uint32_t *sm = xmalloc((size_t)cnt * 4); uint32_t nsm = 0;
    for (uint32_t i = 0; i < cnt; i++) {                           /* :409-427 */
        if (diffs[i] > 0) { inm.key[inm.n] = nodes[i]; inm.val[inm.n] = diffs[i]; inm.alive[inm.n] = 1; inm.n++; }
        else if (diffs[i] == 0) sm[nsm++] = nodes[i];
    }
    for (uint32_t i = cnt; i-- > 0;)                                /* Reverse(node): largest first */
        if (diffs[i] < 0) { outm.key[outm.n] = nodes[i]; outm.val[outm.n] = diffs[i]; outm.alive[outm.n] = 1; outm.n++; }

    for (uint32_t p = 0; p < nsm; p += 2) {                        /* :481-524 */
        if (p + 1 < nsm) {
            add_breaking_biedge(g, sm[p], sm[p + 1], dummy_edge_id, k);   /* :484-493 */
        } else {
            int fi = smap_first(&inm);
            if (fi < 0) DIE("Have an uneven number of self-mirrors, but no other nodes with missing in edges. (:496-498)");
            uint32_t in_node = inm.key[fi];
            add_breaking_biedge(g, sm[p], in_node, dummy_edge_id, k);     /* :502-510 */
            inm.val[fi] -= 1;                                            /* :512 */
            int oi = smap_find(&outm, g->mirror[in_node]);
            if (oi < 0) DIE("Mirror of in_node not found (:517/:521)");
            if (inm.val[fi] == 0) { smap_remove(&inm, fi); smap_remove(&outm, oi); } /* :513-517 */
            else outm.val[oi] += 1;                                       /* :519-521 */
        }
    }

    for (;;) {                                                     /* :526 */
        int oi = smap_first(&outm);
        if (oi < 0) break;
        uint32_t out_node = outm.key[oi];
        int64_t out_diff = outm.val[oi];
        /* choose_in_node_from_iterator, :252-285 */
        int ii = smap_first(&inm);
        if (ii < 0) DIE("in_node_iterator.next().unwrap() on empty map (:262)");
        {
            uint32_t cand = inm.key[ii];
            if ((cand == g->mirror[out_node] && out_diff > -2) || cand == out_node) {
                ii = smap_next(&inm, ii);
                if (ii < 0) DIE("No further in_nodes left (:553)");
            }
        }
        uint32_t in_node = inm.key[ii];
        uint32_t mirror_out_node = g->mirror[in_node];             /* :569 */
        uint32_t mirror_in_node = g->mirror[out_node];             /* :570 */
        add_breaking_biedge(g, out_node, in_node, dummy_edge_id, k); /* :572-577 */
        outm.val[oi] += 1;                                         /* :582 */
        inm.val[ii] -= 1;                                          /* :583 */
        if (outm.val[oi] == 0) smap_remove(&outm, oi);             /* :590-594 */
        if (inm.val[ii] == 0) smap_remove(&inm, ii);               /* :600-602 */
        int mo = smap_find(&outm, mirror_out_node);                /* :609-627 */
        if (mo >= 0) { outm.val[mo] += 1; if (outm.val[mo] == 0) smap_remove(&outm, mo); }
        int mi = smap_find(&inm, mirror_in_node);                  /* :628-644 */
        if (mi >= 0) { inm.val[mi] -= 1; if (inm.val[mi] == 0) smap_remove(&inm, mi); }
    }
    if (smap_first(&inm) >= 0) DIE("in_node_differences not empty after Eulerisation (:648)");
    smap_free(&outm); smap_free(&inm); free(sm); free(nodes); free(diffs);
}

/* debug_assert_graph_has_no_consecutive_dummy_edges, implementation/mod.rs:319-390 */
int og_no_consecutive_dummy_edges(const og_graph *g, uint64_t k) {
    (void)k;
    for (uint32_t n = 0; n < g->n_nodes; n++) {
        uint32_t din = 0, dout = 0, ein = OG_NONE, eout = OG_NONE;
        for (uint32_t e = g->head_in[n]; e != OG_NONE; e = g->next_in[e]) if (is_dummy(g, e)) { din++; ein = e; }
        if (!din) continue;
        for (uint32_t e = g->head_out[n]; e != OG_NONE; e = g->next_out[e]) if (is_dummy(g, e)) { dout++; eout = e; }
        if (!dout) continue;
        if (din == 1 && dout == 1 && og_mirror_edge(g, ein) == eout && g->weight[ein] != 0) continue; /* :370-381 */
        return 0;
    }
    return 1;
}

/* ===================================================================================== */
/* Euler decomposition.  Policy (App. A.2): bigraph 5.0.1                                  */
/* compute_minimum_bidirected_eulerian_cycle_decomposition -- Hierholzer over biedges:     */
/*  * a bitset marks an edge AND its mirror edge used together;                            */
/*  * outer loop over edge indices ascending starts a cycle at the first unused edge;      */
/*  * walk: from the current node take the first unused out-edge in adjacency order         */
/*    (newest first) until none is left (then the walk is back at its start node);          */
/*  * then scan the cycle from index 0 for the first edge whose from-node still has an      */
/*    unused out-edge (policy P5 of mtg_policy.h; its other setting scans from the last     */
/*    index backwards for the last such edge); rotate_left the cycle to that index and      */
/*    continue the walk with that edge (the cycle then ends at that node, so the new closed */
/*    sub-walk is appended);                                                                */
/*  * push the finished cycle.                                                             */
/* Literal (rescanning) form; the product uses an O(E) linked-list formulation that must    */
/* produce the same sequences.                                                              */
/* ===================================================================================== */
static void rotate_left_u32(uint32_t *a, size_t n, size_t by) {
    if (n == 0 || by % n == 0) return;
    by %= n;
    uint32_t *tmp = xmalloc(by * 4);
    memcpy(tmp, a, by * 4);
    memmove(a, a + by, (n - by) * 4);
    memcpy(a + (n - by), tmp, by * 4);
    free(tmp);
}
void og_walks_free(og_walks *w) { if (!w) return; free(w->limits); free(w->edges); free(w); }

typedef struct { uint64_t *limits; uint64_t nw, capw; uint32_t *edges; uint64_t ne, cape; } walks_builder;
static void wb_push_walk(walks_builder *b, const uint32_t *edges, uint64_t n) {
    if (b->ne + n > b->cape) { while (b->ne + n > b->cape) b->cape = b->cape ? b->cape * 2 : 256; b->edges = xrealloc(b->edges, b->cape * 4); }
    memcpy(b->edges + b->ne, edges, n * 4);
    b->ne += n;
    if (b->nw == b->capw) { b->capw = b->capw ? b->capw * 2 : 64; b->limits = xrealloc(b->limits, b->capw * 8); }
    b->limits[b->nw++] = b->ne;
}
static og_walks *wb_finish(walks_builder *b) {
    og_walks *w = xmalloc(sizeof *w);
    w->n_walks = b->nw; w->n_edges = b->ne;
    w->limits = b->limits ? b->limits : xmalloc(8);
    w->edges = b->edges ? b->edges : xmalloc(4);
    return w;
}

og_walks *og_euler_cycles(const og_graph *g) {
    uint32_t ne = g->n_edges;
    uint8_t *used = xcalloc(ne ? ne : 1, 1);
    uint32_t *cycle = xmalloc((size_t)(ne ? ne : 1) * 4);
    walks_builder wb = {0};
    for (uint32_t e0 = 0; e0 < ne; e0++) {
        if (used[e0]) continue;
        size_t len = 0;
        uint32_t start_edge = e0;
        while (start_edge != OG_NONE) {
            uint32_t m = og_mirror_edge(g, start_edge);
            if (m == OG_NONE) DIE("edge %u has no mirror", start_edge);
            used[start_edge] = 1; used[m] = 1;
            uint32_t start_node = g->from[start_edge];
            cycle[len++] = start_edge;
            uint32_t current = g->to[start_edge];
            int has_neighbor = 1;
            while (has_neighbor) {
                has_neighbor = 0;
                for (uint32_t e = g->head_out[current]; e != OG_NONE; e = g->next_out[e]) {
                    if (!used[e]) {
                        uint32_t me = og_mirror_edge(g, e);
                        if (me == OG_NONE) DIE("edge %u has no mirror", e);
                        cycle[len++] = e;
                        used[e] = 1; used[me] = 1;
                        has_neighbor = 1;
                        current = g->to[e];
                        break;
                    }
                }
                if (!has_neighbor && current != start_node) DIE("Euler walk stuck at node %u != start %u: graph not Eulerian", current, start_node);
            }
            /* find new start edge: policy P5 (mtg_policy.h) -- the first such position of the cycle, or the last */
            start_edge = OG_NONE;
            for (size_t step = 0; step < len; step++) {
                size_t ci = mtg_policy_euler_splice_last() ? len - 1 - step : step;
                uint32_t fn = g->from[cycle[ci]];
                for (uint32_t e = g->head_out[fn]; e != OG_NONE; e = g->next_out[e])
                    if (!used[e]) { start_edge = e; break; }
                if (start_edge != OG_NONE) { rotate_left_u32(cycle, len, ci); break; }
            }
        }
        wb_push_walk(&wb, cycle, len);
    }
    free(used); free(cycle);
    return wb_finish(&wb);
}

/* rotate + cut, greedytigs/mod.rs:726-789 (== eulertigs/mod.rs:123-186). Rotates `cycles` in place like the reference. */
og_walks *og_cut_cycles(const og_graph *g, og_walks *cycles, uint64_t k, uint64_t *removed_edges) {
    walks_builder wb = {0};
    uint64_t removed = 0;
    uint64_t begin = 0;
    for (uint64_t c = 0; c < cycles->n_walks; c++) {
        uint32_t *cyc = cycles->edges + begin;
        uint64_t len = cycles->limits[c] - begin;
        begin = cycles->limits[c];
        uint64_t longest_w = 0, longest_i = 0;                     /* :737-745 */
        for (uint64_t i = 0; i < len; i++)
            if (is_dummy(g, cyc[i]) && g->weight[cyc[i]] > longest_w) { longest_w = g->weight[cyc[i]]; longest_i = i; }
        if (longest_w > 0) rotate_left_u32(cyc, len, longest_i);   /* :746-748 */
        uint64_t offset = 0;                                       /* :750 */
        for (uint64_t i = 0; i < len; i++) {                       /* :752 */
            uint32_t e = cyc[i];
            if ((g->weight[e] >= k && is_dummy(g, e)) || (is_dummy(g, e) && i == 0)) { /* :767-769 */
                if (offset < i) wb_push_walk(&wb, cyc + offset, i - offset);   /* :770-771 */
                offset = i + 1;                                    /* :775 */
                removed++;
            }
        }
        if (offset < len) {                                        /* :779-788 */
            if (!is_dummy(g, cyc[len - 1])) wb_push_walk(&wb, cyc + offset, len - offset);
            else if (offset < len - 1) wb_push_walk(&wb, cyc + offset, len - 1 - offset);
        }
    }
    if (removed_edges) *removed_edges = removed;
    return wb_finish(&wb);
}

og_walks *og_compute_greedytigs(og_graph *g, uint64_t k, og_sssp_stats *stats) { /* greedytigs/mod.rs:201-801 */
    og_pair *pairs = NULL;
    uint64_t np = og_greedy_pairs(g, k, &pairs, stats);
    uint64_t dummy_edge_id = og_insert_pair_edges(g, pairs, np);   /* :678-689 */
    free(pairs);
    og_make_eulerian_with_breaking_edges(g, &dummy_edge_id, k);    /* :705 */
    if (!og_decomposes_into_eulerian_bicycles(g)) DIE("Failed to make the graph Eulerian. (:714)");
    og_walks *cycles = og_euler_cycles(g);                         /* :722 */
    og_walks *tigs = og_cut_cycles(g, cycles, k, NULL);            /* :726-789 */
    og_walks_free(cycles);
    return tigs;
}
og_walks *og_compute_eulertigs(og_graph *g, uint64_t k) { /* eulertigs/mod.rs:48-198 */
    uint64_t dummy_edge_id = 0;                                    /* :101 */
    og_make_eulerian_with_breaking_edges(g, &dummy_edge_id, k);    /* :102 */
    if (!og_decomposes_into_eulerian_bicycles(g)) DIE("Failed to make the graph Eulerian. (eulertigs :111)");
    og_walks *cycles = og_euler_cycles(g);                         /* :119 */
    og_walks *tigs = og_cut_cycles(g, cycles, k, NULL);            /* :123-186 */
    og_walks_free(cycles);
    return tigs;
}

/* clib.rs:393-407 */
uint64_t og_flatten_clib(const og_graph *g, const og_walks *tigs, int64_t *tigs_edge_out,
                         uint64_t *tigs_insert_out, uint64_t *tigs_out_limits) {
    uint64_t begin = 0;
    for (uint64_t i = 0; i < tigs->n_walks; i++) {
        for (uint64_t j = begin; j < tigs->limits[i]; j++) {
            uint32_t e = tigs->edges[j];
            tigs_edge_out[j] = (int64_t)g->handle[e] * (g->forwards[e] ? 1 : -1);
            tigs_insert_out[j] = is_dummy(g, e) ? g->weight[e] : 0;
        }
        begin = tigs->limits[i];
        tigs_out_limits[i] = begin;
    }
    return tigs->n_walks;
}

uint64_t og_clib_compute_tigs(og_graph *g, uint64_t tig_algorithm, uint64_t k, int64_t *tigs_edge_out,
                              uint64_t *tigs_insert_out, uint64_t *tigs_out_limits) { /* clib.rs:350-409 */
    og_walks *tigs;
    if (tig_algorithm == 1) {                                      /* :351-361 */
        walks_builder wb = {0};
        for (uint32_t e = 0; e < g->n_edges; e += 2) wb_push_walk(&wb, &e, 1);
        tigs = wb_finish(&wb);
    } else if (tig_algorithm == 3) tigs = og_compute_eulertigs(g, k);
    else if (tig_algorithm == 5) tigs = og_compute_greedytigs(g, k, NULL);
    else DIE("Unknown tigs algorithm identifier %llu (oracle restates ids 1, 3, 5 only)", (unsigned long long)tig_algorithm);
    uint64_t n = og_flatten_clib(g, tigs, tigs_edge_out, tigs_insert_out, tigs_out_limits);
    og_walks_free(tigs);
    return n;
}

/* ===================================================================================== */
/* Optimal matchtigs: the minimum-perfect-matching instance handed to the external matcher, */
/* and the application of its solution.  matchtigs/mod.rs:150-940, the `threads == 1`       */
/* branch (:207-325): its order of (out_node, target) results is the deterministic one      */
/* (sources ascending, targets in Dijkstra pop order); with threads > 1 the reference        */
/* appends worker chunks in completion order (:330-458), which only permutes the matching    */
/* node numbering.                                                                           */
/* ===================================================================================== */
struct og_matching {
    uint64_t k;
    uint32_t n_nodes;
    /* GraphMatchingNodeMap (implementation/mod.rs:188-250): node_id_map[n] is the id range
     * first_id[n] .. first_id[n] + id_count[n]; id_count 0 = the empty Vec */
    uint32_t *first_id, *id_count;
    uint64_t current_node_id;
    /* edges: HashMap<(usize, usize), (weight, out_node, target_node)>, open addressing on n1 << 32 | n2 */
    uint64_t *ekey, *eweight; uint32_t *eout, *etarget; uint64_t ecap, en;
    uint64_t mirror_biedges, mirror_expanded_biedges;
    uint64_t wcc_amount;
    uint64_t *extra_offset; /* matching_node_extra_offset, [current_node_id] */
    uint64_t matching_node_count, matching_edge_count;
};
#define OM_EMPTY UINT64_MAX
static uint64_t om_slot(const og_matching *m, uint64_t key) {
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 17;
    h &= m->ecap - 1;
    while (m->ekey[h] != OM_EMPTY && m->ekey[h] != key) h = (h + 1) & (m->ecap - 1);
    return h;
}
static void om_grow(og_matching *m) {
    uint64_t ocap = m->ecap, *ok = m->ekey, *ow = m->eweight; uint32_t *oo = m->eout, *ot = m->etarget;
    m->ecap = ocap ? ocap * 2 : 1024;
    m->ekey = xmalloc(m->ecap * 8); m->eweight = xmalloc(m->ecap * 8); m->eout = xmalloc(m->ecap * 4); m->etarget = xmalloc(m->ecap * 4);
    memset(m->ekey, 0xFF, m->ecap * 8);
    for (uint64_t i = 0; i < ocap; i++)
        if (ok[i] != OM_EMPTY) { uint64_t h = om_slot(m, ok[i]); m->ekey[h] = ok[i]; m->eweight[h] = ow[i]; m->eout[h] = oo[i]; m->etarget[h] = ot[i]; }
    free(ok); free(ow); free(oo); free(ot);
}
/* HashMap::insert: returns 1 if there was a previous value (which is overwritten) */
static int om_insert(og_matching *m, uint64_t n1, uint64_t n2, uint64_t weight, uint32_t out, uint32_t target) {
    if ((m->en + 1) * 2 > m->ecap) om_grow(m);
    uint64_t key = n1 << 32 | n2, h = om_slot(m, key);
    int previous = m->ekey[h] == key;
    if (!previous) { m->ekey[h] = key; m->en++; }
    m->eweight[h] = weight; m->eout[h] = out; m->etarget[h] = target;
    return previous;
}
/* get_or_create_node_indexes, implementation/mod.rs:202-225 */
static void om_get_or_create(og_matching *m, const og_graph *g, uint32_t n) {
    if (m->id_count[n]) return;
    int64_t d = og_superfluous_out_biedges(g, n);
    if (d < 0) d = -d;
    if (m->current_node_id + (uint64_t)d > 0xFFFFFFFFull) DIE("matching node ids exceed 32 bits");
    m->first_id[n] = (uint32_t)m->current_node_id; m->id_count[n] = (uint32_t)d;
    m->current_node_id += (uint64_t)d;
    uint32_t mn = g->mirror[n];
    m->first_id[mn] = m->first_id[n]; m->id_count[mn] = m->id_count[n];   /* :218 */
}
static uint32_t wcc_find(uint32_t *p, uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
static int cmp_u64x3(const void *a, const void *b) {
    const uint64_t *x = a, *y = b;
    for (int i = 0; i < 3; i++) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

og_matching *og_matching_instance(const og_graph *g, uint64_t k) {
    og_matching *m = xcalloc(1, sizeof *m);
    uint32_t nn = g->n_nodes;
    m->k = k; m->n_nodes = nn;
    m->first_id = xcalloc(nn, 4); m->id_count = xcalloc(nn, 4);
    /* :167-199: out-nodes, in-nodes (the greedy path's classification; multiplicities are not used below) */
    uint32_t *out_nodes = xmalloc((size_t)nn * 4);
    uint8_t *in_node_map = xmalloc(nn);
    int64_t *mult = xmalloc((size_t)nn * 8);
    uint32_t in_count = 0;
    uint32_t n_out = og_classify(g, out_nodes, in_node_map, mult, &in_count, NULL);
    /* :207-300 */
    dijkstra *dj = dijkstra_new(nn);
    dist_vec distances = {0};
    for (uint32_t i = 0; i < n_out; i++) {
        uint32_t out_node = out_nodes[i];
        shortest_path_lens(g, dj, out_node, in_node_map, in_count, k - 1, 1, &distances, NULL);   /* :235-246 */
        for (size_t c = 0; c < distances.n; c++) {
            uint32_t target_node = distances.v[c].node;
            uint64_t weight = distances.v[c].dist;
            if (out_node == target_node) DIE("Found shortest path with same start and end (:251)");
            if (weight == 0) DIE("Found zero weight path from %u to %u (:257)", out_node, target_node);
            int is_mirror_biedge = out_node == g->mirror[target_node] && out_node != target_node;   /* :267 */
            if (is_mirror_biedge) m->mirror_biedges++;
            om_get_or_create(m, g, out_node);
            om_get_or_create(m, g, target_node);
            for (uint32_t a = 0; a < m->id_count[out_node]; a++)
                for (uint32_t b = 0; b < m->id_count[target_node]; b++) {
                    uint64_t c1 = m->first_id[out_node] + a, c2 = m->first_id[target_node] + b;
                    if (c1 == c2) { if (!is_mirror_biedge) DIE("Found self-loop not caused by a mirror biedge (:281)"); continue; }
                    int previous = om_insert(m, c1 < c2 ? c1 : c2, c1 < c2 ? c2 : c1, weight, out_node, target_node);   /* :292 */
                    if (!previous && is_mirror_biedge) m->mirror_expanded_biedges++;
                }
        }
    }
    dijkstra_free(dj); free(distances.v); free(out_nodes); free(in_node_map); free(mult);
    uint64_t T = m->current_node_id;   /* transformed_node_count, :531 */
    /* :545-565: WCCs of the graph (as a plain digraph: mirror nodes are not joined), those with a matching node numbered by first
     * appearance in node order */
    uint32_t *parent = xmalloc((size_t)nn * 4);
    for (uint32_t n = 0; n < nn; n++) parent[n] = n;
    for (uint32_t e = 0; e < g->n_edges; e++) {
        uint32_t a = wcc_find(parent, g->from[e]), b = wcc_find(parent, g->to[e]);
        if (a != b) parent[a > b ? a : b] = a > b ? b : a;
    }
    uint64_t *wcc_index = xmalloc((size_t)nn * 8);
    for (uint32_t n = 0; n < nn; n++) wcc_index[n] = UINT64_MAX;
    for (uint32_t n = 0; n < nn; n++)
        if (m->id_count[n]) { uint32_t r = wcc_find(parent, n); if (wcc_index[r] == UINT64_MAX) wcc_index[r] = m->wcc_amount++; }
    /* :569-587 */
    m->extra_offset = xmalloc(T * 8);
    for (uint64_t i = 0; i < T; i++) m->extra_offset[i] = UINT64_MAX;
    for (uint32_t n = 0; n < nn; n++)
        for (uint32_t a = 0; a < m->id_count[n]; a++) m->extra_offset[m->first_id[n] + a] = 2 * T + 4 * wcc_index[wcc_find(parent, n)];
    free(parent); free(wcc_index);
    m->matching_node_count = T * 2 + 4 * m->wcc_amount;   /* :598 */
    m->matching_edge_count = m->en * 2 + T + 4 * T;       /* :600 */
    return m;
}
void og_matching_counts(const og_matching *m, uint64_t out[7]) {
    out[0] = m->current_node_id; out[1] = m->en; out[2] = m->wcc_amount; out[3] = m->matching_node_count;
    out[4] = m->matching_edge_count; out[5] = m->mirror_biedges; out[6] = m->mirror_expanded_biedges;
}
void og_matching_free(og_matching *m) {
    if (!m) return;
    free(m->first_id); free(m->id_count); free(m->ekey); free(m->eweight); free(m->eout); free(m->etarget); free(m->extra_offset); free(m);
}

/* matchtigs/mod.rs:591-719 */
void og_matching_write(const og_matching *m, const char *path) {
    FILE *f = fopen(path, "w");
    if (!f) DIE("cannot create %s", path);
    uint64_t T = m->current_node_id, k = m->k;
    fprintf(f, "%llu %llu\n", (unsigned long long)m->matching_node_count, (unsigned long long)m->matching_edge_count);
    uint64_t (*sorted)[3] = xmalloc((m->en ? m->en : 1) * 24);   /* :603-607 */
    uint64_t ns = 0;
    for (uint64_t h = 0; h < m->ecap; h++)
        if (m->ekey[h] != OM_EMPTY) { sorted[ns][0] = m->ekey[h] >> 32; sorted[ns][1] = m->ekey[h] & 0xFFFFFFFFull; sorted[ns][2] = m->eweight[h]; ns++; }
    qsort(sorted, ns, 24, cmp_u64x3);
#define OM_X(i) ((unsigned long long)m->extra_offset[i])
    /* first copy, :610-655 */
    int have_last = 0; uint64_t last_n1 = 0;
    for (uint64_t i = 0; i < ns; i++) {
        uint64_t n1 = sorted[i][0], n2 = sorted[i][1], weight = sorted[i][2];
        if (have_last)
            while (last_n1 < n1) {
                fprintf(f, "%llu %llu %llu\n%llu %llu %d\n%llu %llu %d\n", (unsigned long long)last_n1, (unsigned long long)(last_n1 + T),
                        (unsigned long long)(k - 1), (unsigned long long)last_n1, OM_X(last_n1), 0, (unsigned long long)last_n1, OM_X(last_n1) + 1, 0);
                last_n1++;
            }
        fprintf(f, "%llu %llu %llu\n", (unsigned long long)n1, (unsigned long long)n2, (unsigned long long)weight);
        last_n1 = n1; have_last = 1;
    }
    if (!have_last) last_n1 = 0;   /* unwrap_or(0), :638 */
    while (last_n1 < T) {
        fprintf(f, "%llu %llu %llu\n%llu %llu %d\n%llu %llu %d\n", (unsigned long long)last_n1, (unsigned long long)(last_n1 + T),
                (unsigned long long)(k - 1), (unsigned long long)last_n1, OM_X(last_n1), 0, (unsigned long long)last_n1, OM_X(last_n1) + 1, 0);
        last_n1++;
    }
    /* second copy, :657-714 */
    have_last = 0; last_n1 = 0;
    for (uint64_t i = 0; i < ns; i++) {
        uint64_t n1 = sorted[i][0], n2 = sorted[i][1], weight = sorted[i][2];
        if (have_last)
            while (last_n1 < n1) {
                fprintf(f, "%llu %llu %d\n%llu %llu %d\n", (unsigned long long)(last_n1 + T), OM_X(last_n1) + 2, 0,
                        (unsigned long long)(last_n1 + T), OM_X(last_n1) + 3, 0);
                last_n1++;
            }
        last_n1 = n1; have_last = 1;
        if (n1 == n2) continue;
        fprintf(f, "%llu %llu %llu\n", (unsigned long long)(n1 + T), (unsigned long long)(n2 + T), (unsigned long long)weight);
    }
    if (!have_last) last_n1 = 0;
    while (last_n1 < T) {
        fprintf(f, "%llu %llu %d\n%llu %llu %d\n", (unsigned long long)(last_n1 + T), OM_X(last_n1) + 2, 0,
                (unsigned long long)(last_n1 + T), OM_X(last_n1) + 3, 0);
        last_n1++;
    }
#undef OM_X
    free(sorted);
    if (fclose(f) != 0) DIE("write error on %s", path);
}

/* matchtigs/mod.rs:746-940: reads the matcher's solution, inserts the matched edges, Eulerises, walks and cuts. */
og_walks *og_matching_apply(const og_matching *m, og_graph *g, const char *solution_path, uint64_t k) {
    FILE *f = fopen(solution_path, "r");
    if (!f) DIE("cannot open %s", solution_path);
    uint64_t T = m->current_node_id;
    char line[256];
    if (!fgets(line, sizeof line, f)) DIE("matcher output is empty (:756)");   /* header, only debug-asserted */
    uint64_t dummy_edge_id = 0;
    while (fgets(line, sizeof line, f)) {
        unsigned long long a, b;
        if (sscanf(line, "%llu %llu", &a, &b) != 2) DIE("malformed matcher output line: %s", line);
        uint64_t n1 = a, n2 = b;
        if ((n1 >= T && n2 >= T) || n1 >= 2 * T || n2 >= 2 * T) continue;   /* :769-776 */
        if (n1 >= T) n1 -= T;
        if (n2 >= T) n2 -= T;
        uint64_t key = n1 << 32 | n2;
        uint64_t h = m->ecap ? om_slot(m, key) : 0;
        if (!m->ecap || m->ekey[h] != key) {   /* :789-795 */
            if (n1 == n2) continue;
            DIE("Edge does not exist: (%llu, %llu)", (unsigned long long)n1, (unsigned long long)n2);
        }
        uint32_t o1 = m->eout[h], o2 = m->etarget[h];
        dummy_edge_id += 1;   /* :800-808 */
        og_add_edge(g, o1, o2, m->eweight[h], dummy_edge_id, 0, 1);
        og_add_edge(g, g->mirror[o2], g->mirror[o1], m->eweight[h], dummy_edge_id, 0, 0);
    }
    fclose(f);
    og_make_eulerian_with_breaking_edges(g, &dummy_edge_id, k);   /* :831 */
    if (!og_decomposes_into_eulerian_bicycles(g)) DIE("Failed to make the graph Eulerian. (matchtigs :849)");
    og_walks *cycles = og_euler_cycles(g);   /* :857 */
    uint64_t begin = 0;
    for (uint64_t c = 0; c < cycles->n_walks; c++) {   /* the one difference to the greedy cutter: :870-886 */
        uint64_t longest = 0;
        for (uint64_t i = begin; i < cycles->limits[c]; i++)
            if (is_dummy(g, cycles->edges[i]) && g->weight[cycles->edges[i]] > longest) longest = g->weight[cycles->edges[i]];
        if (longest > 0 && longest < k) DIE("Eulerian bicycle contains at least one dummy edge, but no breaking edge (:883)");
        begin = cycles->limits[c];
    }
    og_walks *tigs = og_cut_cycles(g, cycles, k, NULL);   /* :865-935 == greedytigs/mod.rs:726-789 */
    og_walks_free(cycles);
    return tigs;
}

/* ===================================================================================== */
/* Tig spelling, bin.rs:466-606.  Sequences are ASCII; reverse complement per character.   */
/* ===================================================================================== */
static char rc_char(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}
typedef struct { char *s; uint64_t n, cap; } sbuf;
static void sb_reserve(sbuf *b, uint64_t extra) {
    if (b->n + extra > b->cap) { while (b->n + extra > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->s = xrealloc(b->s, b->cap); }
}
static void sb_putc(sbuf *b, char c) { sb_reserve(b, 1); b->s[b->n++] = c; }
static void sb_edge(sbuf *b, const og_graph *g, uint32_t e, const char *seqs, const uint64_t *seq_off, uint64_t offset) {
    uint64_t h = g->handle[e];
    const char *s = seqs + seq_off[h];
    uint64_t len = seq_off[h + 1] - seq_off[h];
    if (offset > len) DIE("overlap offset %llu longer than sequence %llu", (unsigned long long)offset, (unsigned long long)len);
    sb_reserve(b, len - offset);
    if (g->forwards[e]) {                                          /* bin.rs:539-566: seq[offset..] */
        memcpy(b->s + b->n, s + offset, len - offset);
        b->n += len - offset;
    } else {                                                       /* bin.rs:567-596: revcomp(seq[0..len-offset]) */
        for (uint64_t i = len - offset; i-- > 0;) b->s[b->n++] = rc_char(s[i]);
    }
}
char *og_write_walks_fasta(const og_graph *g, const og_walks *tigs, const char *seqs, const uint64_t *seq_off,
                           uint64_t k, uint64_t *out_len) {
    sbuf b = {0};
    uint64_t begin = 0;
    for (uint64_t i = 0; i < tigs->n_walks; i++) {
        uint64_t end = tigs->limits[i];
        char hdr[32];
        int hl = snprintf(hdr, sizeof hdr, ">%llu\n", (unsigned long long)(i + 1)); /* bin.rs:492 */
        sb_reserve(&b, (uint64_t)hl); memcpy(b.s + b.n, hdr, (size_t)hl); b.n += (uint64_t)hl;
        uint32_t prev = tigs->edges[begin];
        sb_edge(&b, g, prev, seqs, seq_off, 0);                    /* bin.rs:497-501 */
        for (uint64_t j = begin + 1; j < end; j++) {               /* bin.rs:514 */
            uint32_t cur = tigs->edges[j];
            if (is_dummy(g, cur)) { prev = cur; continue; }        /* bin.rs:519-531 */
            uint64_t offset = !is_dummy(g, prev) ? k - 1 : k - 1 - g->weight[prev]; /* bin.rs:533-537 */
            sb_edge(&b, g, cur, seqs, seq_off, offset);
            prev = cur;
        }
        sb_putc(&b, '\n');                                         /* bin.rs:601 */
        begin = end;
    }
    sb_putc(&b, '\0');
    *out_len = b.n - 1;
    return b.s;
}
