// matching_instance.cpp -- optimal matchtigs around the external matcher: the minimum-perfect-matching instance built from the
// GPU's candidate lists, its text file, and the matcher's solution turned back into matched pairs.
//
//   build_matching_instance  matchtigs/mod.rs:150-600  (GraphMatchingNodeMap: implementation/mod.rs:188-250)
//   write_matching_instance  matchtigs/mod.rs:591-719
//   read_matching_solution   matchtigs/mod.rs:746-812
//
// All citations are into /root/reference/src/implementation/. The reference runs one bounded Dijkstra per out-node with
// target_amount = #in-nodes (:235-246), i.e. exactly the candidate lists L(s) the SSSP stage produces, and folds the
// (out, target, weight) results one after the other into a node map and a HashMap. Here the same maps come out of order-free
// bulk steps over the lists (first-touch positions by atomic min, a counting sort of the collapsed edges by their smaller
// endpoint, lock-free union-find for the WCCs), all from host threads; the results equal the reference's `threads == 1`
// branch (:207-325), whose result order -- sources ascending, targets in (distance, node) pop order -- is the deterministic
// one (with threads > 1 the reference appends worker chunks in completion order, :330-458, which permutes the numbering).
#include "host_graph.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <string>

#include "parallel.hpp"

namespace mtg {

namespace {

inline void atomic_min_u64(uint64_t *p, uint64_t v) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

// lock-free union-find: roots only ever get hooked under a smaller index
inline uint32_t uf_find(uint32_t *p, uint32_t x) {
    for (;;) {
        const uint32_t px = __atomic_load_n(&p[x], __ATOMIC_RELAXED);
        if (px == x) return x;
        const uint32_t ppx = __atomic_load_n(&p[px], __ATOMIC_RELAXED);
        if (ppx != px) __atomic_store_n(&p[x], ppx, __ATOMIC_RELAXED);  // path halving; any ancestor is a valid parent
        x = px;
    }
}
inline void uf_unite(uint32_t *p, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(p, a);
        b = uf_find(p, b);
        if (a == b) return;
        if (a < b) std::swap(a, b);  // hook the larger root a under b
        uint32_t expect = a;
        if (__atomic_compare_exchange_n(&p[a], &expect, b, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
    }
}

struct Elem {
    uint32_t n2;   // larger collapsed endpoint
    uint32_t src;  // source index
    uint64_t seq;  // position of (source, target) in the reference's result order
};

inline char *put_u64(char *p, uint64_t v) {
    char tmp[20];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char *put_line(char *p, uint64_t a, uint64_t b, uint64_t c) {
    p = put_u64(p, a); *p++ = ' ';
    p = put_u64(p, b); *p++ = ' ';
    p = put_u64(p, c); *p++ = '\n';
    return p;
}

}  // namespace

MatchingInstance *build_matching_instance(const HostGraph &g, uint64_t k, uint64_t S, const uint32_t *out_nodes,
                                          const int32_t *multiplicity, const uint64_t *cand_start, const uint32_t *cand_count,
                                          const uint64_t *pool) {
    const uint64_t V = g.node_count();
    auto *m = new MatchingInstance;
    m->k = k;
    auto ids_of = [&](uint32_t n) { return (uint32_t)std::abs(multiplicity[n]); };  // compute_eulerian_superfluous_out_biedges(..).abs()

    // position of every (source, target) result in the reference's order
    std::vector<uint64_t> base(S + 1, 0);
    for (uint64_t i = 0; i < S; i++) base[i + 1] = base[i] + cand_count[i];
    const uint64_t C = base[S];

    // ---- GraphMatchingNodeMap: ids in first-touch order (out_node before its first target, then targets in list order) ----
    PodVec<uint64_t> first_pos(V);
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) { std::fill(first_pos.begin() + lo, first_pos.begin() + hi, UINT64_MAX); });
    std::atomic<uint64_t> mirror_biedges{0};
    parallel_ranges(S, [&](uint64_t lo, uint64_t hi) {
        uint64_t mb = 0;
        for (uint64_t i = lo; i < hi; i++) {
            if (!cand_count[i]) continue;  // an out-node without results never enters the map (:270 is inside the loop)
            const uint32_t s = out_nodes[i];
            atomic_min_u64(&first_pos[std::min(s, g.mirror[s])], 2 * base[i]);
            const uint64_t *L = pool + cand_start[i];
            for (uint32_t j = 0; j < cand_count[i]; j++) {
                const uint32_t t = (uint32_t)L[j];
                if (t == s) MTG_DIE("Found shortest path with same start and end (matchtigs/mod.rs:251)");
                if (!(L[j] >> 32)) MTG_DIE("Found zero weight path from %u to %u (matchtigs/mod.rs:257)", s, t);
                if (s == g.mirror[t]) mb++;  // is_mirror_biedge, :267 (s != t holds)
                atomic_min_u64(&first_pos[std::min(t, g.mirror[t])], 2 * (base[i] + j) + 1);
            }
        }
        mirror_biedges += mb;
    });
    m->mirror_biedges = mirror_biedges;
    PodVec<uint32_t> at(2 * C);  // touch position -> binode that was created there
    parallel_ranges(2 * C, [&](uint64_t lo, uint64_t hi) { std::fill(at.begin() + lo, at.begin() + hi, NONE); });
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++)
            if (first_pos[n] != UINT64_MAX) at[first_pos[n]] = (uint32_t)n;
    });
    // prefix sums of the id counts in position order
    m->first_id.assign(V, 0);
    m->id_count.assign(V, 0);
    {
        const unsigned P = 64;
        const uint64_t chunk = (2 * C + P - 1) / P;
        std::vector<uint64_t> sums(P + 1, 0);
        parallel_tasks(P, [&](uint64_t c) {
            uint64_t s = 0;
            for (uint64_t p = c * chunk; p < std::min(2 * C, (c + 1) * chunk); p++)
                if (at[p] != NONE) s += ids_of(at[p]);
            sums[c + 1] = s;
        });
        for (unsigned c = 0; c < P; c++) sums[c + 1] += sums[c];
        if (sums[P] > 0xFFFFFFFFull) MTG_DIE("the matching instance has more than 2^32 nodes");
        m->transformed_node_count = sums[P];
        parallel_tasks(P, [&](uint64_t c) {
            uint64_t id = sums[c];
            for (uint64_t p = c * chunk; p < std::min(2 * C, (c + 1) * chunk); p++) {
                const uint32_t n = at[p];
                if (n == NONE) continue;
                const uint32_t cnt = ids_of(n), mn = g.mirror[n];
                m->first_id[n] = m->first_id[mn] = (uint32_t)id;  // implementation/mod.rs:218: the mirror node shares the ids
                m->id_count[n] = m->id_count[mn] = cnt;
                id += cnt;
            }
        });
    }
    { PodVec<uint32_t>().swap(at); PodVec<uint64_t>().swap(first_pos); }
    const uint64_t T = m->transformed_node_count;

    // ---- edges: HashMap<(min, max), (weight, out_node, target_node)>::insert over all id combinations, :276-312 ----
    // counting sort by the smaller endpoint, then per endpoint: order by (larger endpoint, result position); the LAST insert of a
    // key leaves its value, the FIRST decides the "expanded mirror biedge" statistic.
    std::vector<uint64_t> bucket(T + 2, 0);
    auto for_each_combination = [&](uint64_t lo, uint64_t hi, auto &&f) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint32_t s = out_nodes[i];
            const uint64_t *L = pool + cand_start[i];
            for (uint32_t j = 0; j < cand_count[i]; j++) {
                const uint32_t t = (uint32_t)L[j];
                for (uint32_t a = 0; a < m->id_count[s]; a++)
                    for (uint32_t b = 0; b < m->id_count[t]; b++) {
                        const uint32_t c1 = m->first_id[s] + a, c2 = m->first_id[t] + b;
                        if (c1 == c2) {
                            if (s != g.mirror[t]) MTG_DIE("Found self-loop not caused by a mirror biedge (matchtigs/mod.rs:281)");
                            continue;
                        }
                        f(std::min(c1, c2), std::max(c1, c2), i, base[i] + j);
                    }
            }
        }
    };
    parallel_ranges(S, [&](uint64_t lo, uint64_t hi) {
        for_each_combination(lo, hi, [&](uint32_t n1, uint32_t, uint64_t, uint64_t) { __atomic_fetch_add(&bucket[n1 + 2], 1, __ATOMIC_RELAXED); });
    });
    for (uint64_t x = 0; x < T; x++) bucket[x + 2] += bucket[x + 1];  // bucket[x + 1] = begin of x (shifted by one for the cursors)
    const uint64_t n_elems = bucket[T + 1];
    PodVec<Elem> elems(n_elems);
    parallel_ranges(S, [&](uint64_t lo, uint64_t hi) {
        for_each_combination(lo, hi, [&](uint32_t n1, uint32_t n2, uint64_t i, uint64_t seq) {
            const uint64_t at_ = __atomic_fetch_add(&bucket[n1 + 1], 1, __ATOMIC_RELAXED);
            elems[at_] = Elem{n2, (uint32_t)i, seq};
        });
    });
    // now bucket[x] = begin of x, bucket[x + 1] = end of x
    std::vector<uint64_t> uniq(T + 1, 0);
    std::atomic<uint64_t> mirror_expanded{0};
    auto target_of = [&](const Elem &e) { return (uint32_t)pool[cand_start[e.src] + (e.seq - base[e.src])]; };
    parallel_ranges(T, [&](uint64_t lo, uint64_t hi) {
        uint64_t me = 0;
        for (uint64_t x = lo; x < hi; x++) {
            Elem *b = elems.data() + bucket[x], *e = elems.data() + bucket[x + 1];
            std::sort(b, e, [](const Elem &p, const Elem &q) { return p.n2 != q.n2 ? p.n2 < q.n2 : p.seq < q.seq; });
            uint64_t u = 0;
            for (Elem *p = b; p < e; p++)
                if (p == b || p->n2 != p[-1].n2) {
                    u++;
                    if (out_nodes[p->src] == g.mirror[target_of(*p)]) me++;  // previous.is_none() && is_mirror_biedge, :299-303
                }
            uniq[x + 1] = u;
        }
        mirror_expanded += me;
    });
    m->mirror_expanded_biedges = mirror_expanded;
    for (uint64_t x = 0; x < T; x++) uniq[x + 1] += uniq[x];
    const uint64_t E = uniq[T];
    m->edge_begin.assign(uniq.begin(), uniq.end());
    m->edge_n2.resize(E);
    m->edge_weight.resize(E);
    m->edge_out.resize(E);
    m->edge_target.resize(E);
    parallel_ranges(T, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t x = lo; x < hi; x++) {
            const Elem *b = elems.data() + bucket[x], *e = elems.data() + bucket[x + 1];
            uint64_t o = uniq[x];
            for (const Elem *p = b; p < e; p++)
                if (p + 1 == e || p[1].n2 != p->n2) {  // the last insert of the key
                    const uint64_t key = pool[cand_start[p->src] + (p->seq - base[p->src])];
                    m->edge_n2[o] = p->n2;
                    m->edge_weight[o] = (uint32_t)(key >> 32);
                    m->edge_out[o] = out_nodes[p->src];
                    m->edge_target[o] = (uint32_t)key;
                    o++;
                }
        }
    });
    { PodVec<Elem>().swap(elems); }

    // ---- WCCs of the graph as a plain digraph (:545-546), relevant ones numbered by first appearance in node order (:549-565) ----
    PodVec<uint32_t> parent(V);
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) { for (uint64_t n = lo; n < hi; n++) parent[n] = (uint32_t)n; });
    parallel_ranges(g.edge_count(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t e = lo; e < hi; e++) uf_unite(parent.data(), g.e_from[e], g.e_to[e]);
    });
    PodVec<uint64_t> first_node(V);  // per root: the smallest node of the component that has matching nodes
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) { std::fill(first_node.begin() + lo, first_node.begin() + hi, UINT64_MAX); });
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++)
            if (m->id_count[n]) atomic_min_u64(&first_node[uf_find(parent.data(), (uint32_t)n)], n);
    });
    std::vector<uint64_t> firsts;
    for (uint64_t r = 0; r < V; r++)
        if (first_node[r] != UINT64_MAX) firsts.push_back(first_node[r]);
    std::sort(firsts.begin(), firsts.end());
    m->wcc_amount = firsts.size();
    auto wcc_index_of = [&](uint32_t n) {
        const uint64_t f = first_node[uf_find(parent.data(), n)];
        return (uint64_t)(std::lower_bound(firsts.begin(), firsts.end(), f) - firsts.begin());
    };
    // :569-587: input nodes ascending, later writes win -> the larger node of a binode decides
    m->extra_offset.assign(T, UINT64_MAX);
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++) {
            if (!m->id_count[n] || n < g.mirror[n]) continue;
            const uint64_t off = 2 * T + 4 * wcc_index_of((uint32_t)n);
            for (uint32_t a = 0; a < m->id_count[n]; a++) m->extra_offset[m->first_id[n] + a] = off;
        }
    });
    m->matching_node_count = T * 2 + 4 * m->wcc_amount;  // :598
    m->matching_edge_count = E * 2 + T + 4 * T;          // :600
    return m;
}

// matchtigs/mod.rs:591-719. Per first-copy node x (from the first edge's n1 on -- the reference starts its "last_n1" there, :611,
// :638): its edges, then its link to the second copy and its two extra-node edges; then the same for the second copy.
uint64_t write_matching_instance(const MatchingInstance &m, const char *path) {
    FILE *f = std::fopen(path, "w");
    if (!f) MTG_DIE("cannot create %s: %s", path, std::strerror(errno));
    const uint64_t T = m.transformed_node_count, E = m.edge_n2.size(), k = m.k;
    uint64_t written = 0;
    {
        char hdr[64];
        char *p = put_u64(hdr, m.matching_node_count);
        *p++ = ' ';
        p = put_u64(p, m.matching_edge_count);
        *p++ = '\n';
        written += std::fwrite(hdr, 1, (size_t)(p - hdr), f);
    }
    uint64_t first = 0;  // last_n1 of the first edge, or unwrap_or(0)
    if (E)
        while (m.edge_begin[first + 1] == 0) first++;
    const unsigned P = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const uint64_t nodes_per_chunk = 1u << 15;
    std::vector<std::vector<char>> bufs(P);
    for (int copy = 0; copy < 2; copy++) {
        for (uint64_t wave = first; wave < T; wave += nodes_per_chunk * P) {
            parallel_tasks(P, [&](uint64_t c) {
                const uint64_t lo = std::min(T, wave + c * nodes_per_chunk), hi = std::min(T, lo + nodes_per_chunk);
                std::vector<char> &b = bufs[c];
                b.resize((size_t)((m.edge_begin[hi] - m.edge_begin[lo]) * 52 + (hi - lo) * 140 + 16));
                char *p = b.data();
                for (uint64_t x = lo; x < hi; x++) {
                    if (copy == 0) {
                        for (uint64_t e = m.edge_begin[x]; e < m.edge_begin[x + 1]; e++) p = put_line(p, x, m.edge_n2[e], m.edge_weight[e]);
                        p = put_line(p, x, x + T, k - 1);
                        p = put_line(p, x, m.extra_offset[x], 0);
                        p = put_line(p, x, m.extra_offset[x] + 1, 0);
                    } else {
                        for (uint64_t e = m.edge_begin[x]; e < m.edge_begin[x + 1]; e++)
                            p = put_line(p, x + T, (uint64_t)m.edge_n2[e] + T, m.edge_weight[e]);
                        p = put_line(p, x + T, m.extra_offset[x] + 2, 0);
                        p = put_line(p, x + T, m.extra_offset[x] + 3, 0);
                    }
                }
                b.resize((size_t)(p - b.data()));
            }, P);
            for (unsigned c = 0; c < P; c++) {
                if (!bufs[c].empty() && std::fwrite(bufs[c].data(), 1, bufs[c].size(), f) != bufs[c].size())
                    MTG_DIE("write error on %s", path);
                written += bufs[c].size();
                bufs[c].clear();
            }
        }
    }
    if (std::fclose(f) != 0) MTG_DIE("write error on %s", path);
    return written;
}

// matchtigs/mod.rs:746-812: every solution line that is an edge of the first copy (or between the copies) becomes one matched
// pair (original out-node, original in-node, weight), in file order.
std::vector<Pair> read_matching_solution(const MatchingInstance &m, const char *path) {
    FILE *f = std::fopen(path, "r");
    if (!f) MTG_DIE("cannot open %s: %s", path, std::strerror(errno));
    const uint64_t T = m.transformed_node_count;
    std::vector<Pair> pairs;
    char line[256];
    if (!std::fgets(line, sizeof line, f)) MTG_DIE("matcher output %s is empty (matchtigs/mod.rs:756)", path);  // header: only debug-asserted
    while (std::fgets(line, sizeof line, f)) {
        char *end = nullptr;
        errno = 0;
        uint64_t n1 = std::strtoull(line, &end, 10);
        if (end == line || *end != ' ' || errno) MTG_DIE("malformed matcher output line: %s", line);
        char *second = end + 1;
        uint64_t n2 = std::strtoull(second, &end, 10);
        if (end == second || errno) MTG_DIE("malformed matcher output line: %s", line);
        if ((n1 >= T && n2 >= T) || n1 >= 2 * T || n2 >= 2 * T) continue;  // :769-776
        if (n1 >= T) n1 -= T;
        if (n2 >= T) n2 -= T;
        // edges.get(&(n1, n2)): keys are (min, max), the lookup is not normalised (:789)
        const uint32_t *b = m.edge_n2.data() + m.edge_begin[n1], *e = m.edge_n2.data() + m.edge_begin[n1 + 1];
        const uint32_t *p = n2 <= 0xFFFFFFFFull ? std::lower_bound(b, e, (uint32_t)n2) : e;
        if (p == e || *p != n2) {
            if (n1 == n2) continue;
            MTG_DIE("Edge does not exist: (%llu, %llu) (matchtigs/mod.rs:794)", (unsigned long long)n1, (unsigned long long)n2);
        }
        const uint64_t i = (uint64_t)(p - m.edge_n2.data());
        pairs.push_back(Pair{m.edge_out[i], m.edge_target[i], m.edge_weight[i]});
    }
    std::fclose(f);
    return pairs;
}

}  // namespace mtg
