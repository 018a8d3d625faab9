"""Second, independent restatement of the reference path in plain Python (small cases only).

Written directly from the reference lines (not from oracle/mtg_oracle.c) so that the two
restatements can be differenced against each other on random bigraphs.
Citations: file:line into /root/reference/.  Third-party behaviour: SURVEY Appendix A policies.
"""
from __future__ import annotations

import heapq
import os
from dataclasses import dataclass

# The four out-of-tree policies (include/mtg_policy.h: P1 heap tie-break, P2 inclusive bound, P3 adjacency order, P4 union-find tie),
# as a bit mask like mtg_policies() / og_policies(): 0 = the behaviour SURVEY App. A describes, bit i set = P(i+1) flipped. The
# flipped-policy fuzz runs this module, the oracle and the product under the same non-zero mask (tests/test_fuzz_small.py).
POLICY = int(os.environ.get("MTG_POLICY", "0"))
P_HEAP_TIE_DESCENDING, P_BOUND_EXCLUSIVE, P_ADJACENCY_OLDEST_FIRST, P_UNION_TIE_SECOND_UNDER_FIRST = (bool(POLICY & 1), bool(POLICY & 2),
                                                                                                    bool(POLICY & 4), bool(POLICY & 8))
P_EULER_SPLICE_LAST = bool(POLICY & 16)


@dataclass
class Edge:
    frm: int
    to: int
    weight: int
    dummy_id: int
    handle: int
    forwards: bool

    @property
    def is_dummy(self):
        return self.dummy_id != 0  # implementation/mod.rs:291-293


class PyBigraph:
    def __init__(self, n_nodes, mirror):
        self.mirror = list(mirror)
        self.out = [[] for _ in range(n_nodes)]  # edge ids, oldest first; iterate reversed (petgraph newest-first)
        self.inn = [[] for _ in range(n_nodes)]
        self.edges: list[Edge] = []

    @property
    def n(self):
        return len(self.mirror)

    def add_edge(self, f, t, w, dummy_id, handle, forwards):
        self.edges.append(Edge(f, t, w, dummy_id, handle, bool(forwards)))
        e = len(self.edges) - 1
        self.out[f].append(e)
        self.inn[t].append(e)
        return e

    def out_neighbors(self, n):  # iteration order: policy P3
        return iter(self.out[n]) if P_ADJACENCY_OLDEST_FIRST else reversed(self.out[n])

    def mirror_edge(self, e):
        d = self.edges[e]
        rf, rt = self.mirror[d.to], self.mirror[d.frm]
        for m in self.out_neighbors(rf):
            x = self.edges[m]
            if x.to == rt and (x.weight, x.dummy_id, x.handle) == (d.weight, d.dummy_id, d.handle) and x.forwards != d.forwards:
                return m
        return None

    def diff(self, n):  # compute_eulerian_superfluous_out_biedges (App. A.2)
        if self.mirror[n] == n:
            return len(self.out[n]) % 2
        return len(self.out[n]) - len(self.inn[n])


def from_unitig_links(weights, links):
    """clib.rs:97-259 with disjoint-sets 0.4.2 semantics (App. A.4)."""
    n = 4 * len(weights)
    parent, rank = list(range(n)), [0] * n

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    def union(a, b):
        a, b = find(a), find(b)
        if a == b:
            return
        if rank[a] > rank[b]:
            parent[b] = a
        elif rank[b] > rank[a]:
            parent[a] = b
        elif not P_UNION_TIE_SECOND_UNDER_FIRST:  # equal ranks: policy P4
            parent[a] = b
            rank[b] += 1
        else:
            parent[b] = a
            rank[a] += 1

    fi, fo, bi, bo = (lambda u: 4 * u), (lambda u: 4 * u + 2), (lambda u: 4 * u + 3), (lambda u: 4 * u + 1)
    for (ua, sa, ub, sb) in links:  # clib.rs:144-169
        out_a = fo(ua) if sa else bo(ua)
        in_b = fi(ub) if sb else bi(ub)
        mirror_in_a = bi(ua) if sa else fi(ua)
        mirror_out_b = bo(ub) if sb else fo(ub)
        union(out_a, in_b)
        union(mirror_in_a, mirror_out_b)
    reps = sorted(set(find(i) for i in range(n)))
    idx = {r: i for i, r in enumerate(reps)}
    g = PyBigraph(len(reps), [None] * len(reps))
    for u, w in enumerate(weights):
        n1, n2, mn2, mn1 = idx[find(fi(u))], idx[find(fo(u))], idx[find(bi(u))], idx[find(bo(u))]
        g.mirror[n1], g.mirror[mn1] = mn1, n1
        g.mirror[n2], g.mirror[mn2] = mn2, n2
        g.add_edge(n1, n2, w, 0, u, True)
        g.add_edge(mn2, mn1, w, 0, u, False)
    assert all(g.mirror[g.mirror[x]] == x for x in range(g.n)), "verify_node_pairing"
    assert all(g.mirror_edge(e) is not None for e in range(len(g.edges))), "verify_edge_mirror_property"
    return g


def dijkstra(g: PyBigraph, source, live, target_amount, max_weight, stats=None):
    """traitgraph-algo shortest_path_lens (App. A.1), forbid_source_target = true."""
    sgn = -1 if P_HEAP_TIE_DESCENDING else 1  # pop order among equal distances: policy P1
    heap = [(0, sgn * source)]
    dist = {source: 0}
    found = []
    while heap:
        w, n = heapq.heappop(heap)
        n *= sgn
        if dist[n] < w:
            continue
        if (w >= max_weight) if P_BOUND_EXCLUSIVE else (w > max_weight):  # policy P2
            break
        if live[n] and n != source:
            found.append((n, w))
            if len(found) == target_amount:
                break
        if stats is not None:
            stats["settled_nodes"] += 1
        for e in g.out_neighbors(n):
            if stats is not None:
                stats["relaxed_edges"] += 1
            ed = g.edges[e]
            nw = w + ed.weight
            if nw < dist.get(ed.to, 1 << 62):
                dist[ed.to] = nw
                heapq.heappush(heap, (nw, sgn * ed.to))
    return found


def classify(g: PyBigraph):
    out_nodes, live, mult = [], [False] * g.n, [0] * g.n
    for n in range(g.n):  # greedytigs/mod.rs:229-245
        d = g.diff(n)
        if g.mirror[n] == n and d != 0:
            live[n] = True
            mult[n] = d
            out_nodes.append(n)
        elif d > 0:
            live[n] = True
            mult[n] = d
        elif d < 0:
            out_nodes.append(n)
            mult[n] = d
    return out_nodes, live, mult


def greedy_pairs(g: PyBigraph, k, events=None):
    """greedytigs/mod.rs:301-523, single thread. `events` (a dict) counts how often the corner rules fire (tests/fuzz_small.py)."""
    out_nodes, live, mult = classify(g)
    stats = {"settled_nodes": 0, "relaxed_edges": 0}
    res = []

    def ev(name):
        if events is not None:
            events[name] = events.get(name, 0) + 1

    for o in out_nodes:
        o_sm = g.mirror[o] == o
        om = g.mirror[o]
        M = mult[om]
        assert 0 <= M <= 4
        if M == 0:
            continue
        if M >= 3:
            ev("demand_3_or_4")
        if o_sm:
            ev("self_mirror_source")
        while M > 0:
            T = M + 1
            D = dijkstra(g, o, live, T, k - 1, stats)
            if not D:
                break
            abort = len(D) < T
            if len(D) >= 2 and any(D[i][1] == D[i + 1][1] for i in range(len(D) - 1)):
                ev("equal_distance_tie")
            for (t, d) in D:
                self_edge = False
                if t == om:
                    if M < 2:
                        ev("own_mirror_skipped")
                        continue
                    self_edge = True
                    ev("self_mirror_edge")
                t_sm = g.mirror[t] == t
                tm = g.mirror[t]
                M = mult[o] if o_sm else -mult[o]
                if M == 0:
                    break
                if not self_edge and mult[t] == 0:
                    live[t] = False
                    ev("stale_target")
                    continue
                if d == k - 1:
                    ev("distance_k_minus_1")
                if t_sm:
                    ev("self_mirror_target")
                res.append((o, t, d))
                red = 2 if self_edge else 1
                if o_sm:
                    mult[o] -= 1
                else:
                    mult[o] += red
                    mult[om] -= red
                M = -mult[o]
                if not self_edge:
                    mult[t] -= 1
                    if not t_sm:
                        mult[tm] += 1
                if M == 0:
                    live[om] = False
                if not self_edge and mult[t] == 0:
                    live[t] = False
            if abort:
                break
    return res, stats


def insert_pair_edges(g: PyBigraph, pairs):
    did = 0
    for (o, t, d) in pairs:  # greedytigs/mod.rs:678-689
        did += 1
        g.add_edge(o, t, d, did, 0, True)
        g.add_edge(g.mirror[t], g.mirror[o], d, did, 0, False)
    return did


def make_eulerian(g: PyBigraph, dummy_edge_id, k):
    """implementation/mod.rs:392-649 with real ordered maps."""
    nd = []
    for n in range(g.n):  # find_non_eulerian_binodes_with_differences (App. A.2)
        if g.mirror[n] == n:
            if len(g.out[n]) % 2:
                nd.append((n, 0))
        else:
            d = len(g.out[n]) - len(g.inn[n])
            if d:
                nd.append((n, d))
    outd = {n: d for n, d in nd if d < 0}
    ind = {n: d for n, d in nd if d > 0}
    sms = [n for n, d in nd if d == 0]

    def add(o, t):
        nonlocal dummy_edge_id
        dummy_edge_id += 1
        g.add_edge(o, t, k, dummy_edge_id, 0, True)
        g.add_edge(g.mirror[t], g.mirror[o], k, dummy_edge_id, 0, False)

    for p in range(0, len(sms), 2):  # :481-524
        if p + 1 < len(sms):
            add(sms[p], sms[p + 1])
        else:
            t = min(ind)
            add(sms[p], t)
            ind[t] -= 1
            if ind[t] == 0:
                del ind[t]
                del outd[g.mirror[t]]
            else:
                outd[g.mirror[t]] += 1
    while outd:  # :526-645
        o = max(outd)  # Reverse(node) ordering
        od = outd[o]
        keys = sorted(ind)
        t = keys[0]
        if (t == g.mirror[o] and od > -2) or t == o:  # :252-285
            t = keys[1]
        mo, mi = g.mirror[t], g.mirror[o]
        add(o, t)
        outd[o] += 1
        ind[t] -= 1
        if outd[o] == 0:
            del outd[o]
        if ind[t] == 0:
            del ind[t]
        if mo in outd:
            outd[mo] += 1
            if outd[mo] == 0:
                del outd[mo]
        if mi in ind:
            ind[mi] -= 1
            if ind[mi] == 0:
                del ind[mi]
    assert not ind
    return dummy_edge_id


def euler_cycles(g: PyBigraph):
    """bigraph compute_minimum_bidirected_eulerian_cycle_decomposition (App. A.2), literal."""
    used = [False] * len(g.edges)
    cycles = []
    for e0 in range(len(g.edges)):
        if used[e0]:
            continue
        cycle = []
        start = e0
        while start is not None:
            used[start] = True
            used[g.mirror_edge(start)] = True
            start_node = g.edges[start].frm
            cycle.append(start)
            cur = g.edges[start].to
            while True:
                nxt = next((e for e in g.out_neighbors(cur) if not used[e]), None)
                if nxt is None:
                    assert cur == start_node
                    break
                cycle.append(nxt)
                used[nxt] = True
                used[g.mirror_edge(nxt)] = True
                cur = g.edges[nxt].to
            start = None
            for ci in (range(len(cycle) - 1, -1, -1) if P_EULER_SPLICE_LAST else range(len(cycle))):  # policy P5
                e = cycle[ci]
                cand = next((x for x in g.out_neighbors(g.edges[e].frm) if not used[x]), None)
                if cand is not None:
                    start = cand
                    cycle = cycle[ci:] + cycle[:ci]
                    break
        cycles.append(cycle)
    return cycles


def cut_cycles(g: PyBigraph, cycles, k):
    """greedytigs/mod.rs:726-789."""
    tigs = []
    for cyc in cycles:
        lw, li = 0, 0
        for i, e in enumerate(cyc):
            ed = g.edges[e]
            if ed.is_dummy and ed.weight > lw:
                lw, li = ed.weight, i
        if lw > 0:
            cyc = cyc[li:] + cyc[:li]
        off = 0
        for i, e in enumerate(cyc):
            ed = g.edges[e]
            if (ed.weight >= k and ed.is_dummy) or (ed.is_dummy and i == 0):
                if off < i:
                    tigs.append(cyc[off:i])
                off = i + 1
        if off < len(cyc):
            if not g.edges[cyc[-1]].is_dummy:
                tigs.append(cyc[off:])
            elif off < len(cyc) - 1:
                tigs.append(cyc[off:-1])
    return tigs


def compute_greedytigs(g: PyBigraph, k):
    pairs, stats = greedy_pairs(g, k)
    did = insert_pair_edges(g, pairs)
    make_eulerian(g, did, k)
    return cut_cycles(g, euler_cycles(g), k), pairs, stats


def compute_eulertigs(g: PyBigraph, k):
    make_eulerian(g, 0, k)
    return cut_cycles(g, euler_cycles(g), k)


# ---- optimal matchtigs around the external matcher (matchtigs/mod.rs:150-940, threads == 1) ----
class MatchingInstance:
    """matchtigs/mod.rs:150-600: node map (implementation/mod.rs:188-250), edge map, WCC extra offsets."""

    def __init__(self, g: PyBigraph, k):
        self.k = k
        out_nodes, live, _ = classify(g)  # :167-199 (same classification as the greedy path)
        in_count = sum(live)
        self.node_id_map = [[] for _ in range(g.n)]
        self.current_node_id = 0
        self.edges = {}
        self.mirror_biedges = self.mirror_expanded = 0
        for o in out_nodes:  # :225-300
            for t, w in dijkstra(g, o, live, in_count, k - 1):
                assert o != t and w != 0
                is_mirror = o == g.mirror[t] and o != t
                self.mirror_biedges += is_mirror
                for n in (o, t):  # get_or_create_node_indexes
                    if not self.node_id_map[n]:
                        ids = list(range(self.current_node_id, self.current_node_id + abs(g.diff(n))))
                        self.current_node_id += len(ids)
                        self.node_id_map[n] = ids
                        self.node_id_map[g.mirror[n]] = list(ids)
                for c1 in self.node_id_map[o]:
                    for c2 in self.node_id_map[t]:
                        if c1 == c2:
                            assert is_mirror
                            continue
                        key = (min(c1, c2), max(c1, c2))
                        if key not in self.edges and is_mirror:
                            self.mirror_expanded += 1
                        assert key not in self.edges or self.edges[key][0] == w
                        self.edges[key] = (w, o, t)
        T = self.T = self.current_node_id
        # :545-565 WCCs of the plain digraph; relevant ones numbered by first appearance in node order
        comp = list(range(g.n))

        def find(x):
            while comp[x] != x:
                comp[x] = comp[comp[x]]
                x = comp[x]
            return x

        for e in g.edges:
            a, b = find(e.frm), find(e.to)
            if a != b:
                comp[a] = b
        wcc_map = {}
        for n in range(g.n):
            if self.node_id_map[n] and find(n) not in wcc_map:
                wcc_map[find(n)] = len(wcc_map)
        self.wcc_amount = len(wcc_map)
        self.extra = [None] * T  # :569-587
        for n in range(g.n):
            for mnode in self.node_id_map[n]:
                self.extra[mnode] = 2 * T + 4 * wcc_map[find(n)]
        self.matching_node_count = 2 * T + 4 * self.wcc_amount
        self.matching_edge_count = 2 * len(self.edges) + T + 4 * T

    def text(self) -> str:
        """:591-719"""
        T, k, X = self.T, self.k, self.extra
        out = [f"{self.matching_node_count} {self.matching_edge_count}"]
        sorted_edges = sorted((n1, n2, v[0]) for (n1, n2), v in self.edges.items())
        last = None
        for n1, n2, w in sorted_edges:
            if last is not None:
                while last < n1:
                    out += [f"{last} {last + T} {k - 1}", f"{last} {X[last]} 0", f"{last} {X[last] + 1} 0"]
                    last += 1
            out.append(f"{n1} {n2} {w}")
            last = n1
        last = last or 0
        while last < T:
            out += [f"{last} {last + T} {k - 1}", f"{last} {X[last]} 0", f"{last} {X[last] + 1} 0"]
            last += 1
        last = None
        for n1, n2, w in sorted_edges:
            if last is not None:
                while last < n1:
                    out += [f"{last + T} {X[last] + 2} 0", f"{last + T} {X[last] + 3} 0"]
                    last += 1
            last = n1
            out.append(f"{n1 + T} {n2 + T} {w}")
        last = last or 0
        while last < T:
            out += [f"{last + T} {X[last] + 2} 0", f"{last + T} {X[last] + 3} 0"]
            last += 1
        return "\n".join(out) + "\n"

    def apply(self, g: PyBigraph, solution_text: str):
        """:746-940 -> matchtigs (g is mutated)"""
        T, k = self.T, self.k
        did = 0
        for line in solution_text.splitlines()[1:]:
            n1, n2 = (int(x) for x in line.split(" ")[:2])
            if (n1 >= T and n2 >= T) or n1 >= 2 * T or n2 >= 2 * T:
                continue
            n1, n2 = (n1 - T if n1 >= T else n1), (n2 - T if n2 >= T else n2)
            if (n1, n2) not in self.edges:
                assert n1 == n2, f"Edge does not exist: ({n1}, {n2})"
                continue
            w, o1, o2 = self.edges[(n1, n2)]
            did += 1
            g.add_edge(o1, o2, w, did, 0, True)
            g.add_edge(g.mirror[o2], g.mirror[o1], w, did, 0, False)
        make_eulerian(g, did, k)
        cycles = euler_cycles(g)
        for cyc in cycles:
            longest = max([g.edges[e].weight for e in cyc if g.edges[e].is_dummy], default=0)
            assert longest == 0 or longest >= k, "Eulerian bicycle contains at least one dummy edge, but no breaking edge"
        return cut_cycles(g, cycles, k)


def flatten_clib(g: PyBigraph, tigs):
    """clib.rs:393-407."""
    eo, io, lim = [], [], []
    for t in tigs:
        for e in t:
            ed = g.edges[e]
            eo.append(ed.handle * (1 if ed.forwards else -1))
            io.append(0 if not ed.is_dummy else ed.weight)
        lim.append(len(eo))
    return eo, io, lim


_RC = {"A": "T", "C": "G", "G": "C", "T": "A"}


def fasta(g: PyBigraph, tigs, seqs, k):
    """bin.rs:466-606."""
    out = []
    for i, t in enumerate(tigs):
        out.append(f">{i + 1}\n")

        def spell(e, off):
            ed = g.edges[e]
            s = seqs[ed.handle]
            if ed.forwards:
                return s[off:]
            return "".join(_RC[c] for c in reversed(s[: len(s) - off]))

        out.append(spell(t[0], 0))
        prev = t[0]
        for cur in t[1:]:
            if g.edges[cur].is_dummy:
                prev = cur
                continue
            off = k - 1 if not g.edges[prev].is_dummy else k - 1 - g.edges[prev].weight
            out.append(spell(cur, off))
            prev = cur
        out.append("\n")
    return "".join(out)
