"""ctypes binding of the CPU oracle (oracle/libmtg_oracle.so).

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent.parent / "oracle"
_LIB = None


class Pair(C.Structure):
    _fields_ = [("out_node", C.c_uint32), ("in_node", C.c_uint32), ("distance", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_uint64),
        ("unnecessary", C.c_uint64),
        ("settled_nodes", C.c_uint64),
        ("relaxed_edges", C.c_uint64),
        ("queries", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Walks(C.Structure):
    _fields_ = [
        ("n_walks", C.c_uint64),
        ("n_edges", C.c_uint64),
        ("limits", C.POINTER(C.c_uint64)),
        ("edges", C.POINTER(C.c_uint32)),
    ]


def build_oracle(force: bool = False) -> Path:
    """libmtg_oracle.so -- or, when MTG_POLICY names another setting of the five out-of-tree policies (include/mtg_policy.h; the
    flipped-policy fuzz sets 31 = all five flipped), the build of the oracle that follows it."""
    import os

    flipped = int(os.environ.get("MTG_POLICY", "0")) != 0
    name = "libmtg_oracle_flipped.so" if flipped else "libmtg_oracle.so"
    so = ORACLE_DIR / name
    deps = [ORACLE_DIR / "mtg_oracle.c", ORACLE_DIR / "mtg_oracle.h", ORACLE_DIR.parent / "include" / "mtg_policy.h"]
    if force or not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", name], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    import os

    so = build_oracle()
    L = C.CDLL(str(so))
    L.og_policies.restype = C.c_uint
    if int(L.og_policies()) != int(os.environ.get("MTG_POLICY", "0")):
        raise RuntimeError(f"{so} follows policy mask {int(L.og_policies())}, MTG_POLICY asks for {os.environ.get('MTG_POLICY', '0')}")
    vp, u32, u64, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64
    P = C.POINTER
    sig = {
        "og_graph_new": (vp, [u32]),
        "og_graph_free": (None, [vp]),
        "og_graph_from_arrays": (vp, [u32, vp, u32, vp, vp, vp]),
        "og_add_node": (u32, [vp]),
        "og_set_mirror_nodes": (None, [vp, u32, u32]),
        "og_add_edge": (u32, [vp, u32, u32, u64, u64, u64, C.c_int]),
        "og_node_count": (u32, [vp]),
        "og_edge_count": (u32, [vp]),
        "og_mirror_node": (u32, [vp, u32]),
        "og_edge_get": (None, [vp, u32, P(u32), P(u32), P(u64), P(u64), P(u64), P(C.c_int)]),
        "og_out_edges": (u32, [vp, u32, P(u32), u32]),
        "og_mirror_edge": (u32, [vp, u32]),
        "og_verify_node_pairing": (C.c_int, [vp]),
        "og_verify_edge_mirror_property": (C.c_int, [vp]),
        "og_builder_new": (vp, [u64]),
        "og_builder_merge_nodes": (None, [vp, u64, C.c_int, u64, C.c_int]),
        "og_builder_build": (vp, [vp, P(u64)]),
        "og_builder_merge_nodes_many": (None, [vp, u64, P(i64)]),
        "og_superfluous_out_biedges": (i64, [vp, u32]),
        "og_find_non_eulerian": (u32, [vp, P(u32), P(i64)]),
        "og_classify": (u32, [vp, P(u32), P(C.c_uint8), P(i64), P(u32), P(u32)]),
        "og_greedy_pairs": (u64, [vp, u64, P(P(Pair)), P(Stats)]),
        "og_greedy_pairs_prefix": (u64, [vp, u64, u64, P(P(Pair)), P(Stats)]),
        "og_candidate_lists": (u32, [vp, u64, P(P(u32)), P(P(u64)), P(P(u64)), P(Stats)]),
        "og_greedy_pairs_mt": (u64, [vp, u64, u32, P(P(Pair)), P(Stats)]),
        "og_greedy_pairs_mt_prefix": (u64, [vp, u64, u32, u64, P(P(Pair)), P(Stats)]),
        "og_candidate_lists_range": (u32, [vp, u64, u32, u32, P(P(u32)), P(P(u64)), P(P(u64)), P(Stats)]),
        "og_greedy_pairs_given": (u64, [vp, u64, vp, u64, vp, vp, P(P(Pair)), P(Stats)]),
        "og_candidate_lists_given": (None, [vp, u64, vp, u64, vp, P(P(u64)), P(P(u64)), P(Stats)]),
        "og_free": (None, [vp]),
        "og_insert_pair_edges": (u64, [vp, P(Pair), u64]),
        "og_make_eulerian_with_breaking_edges": (None, [vp, P(u64), u64]),
        "og_decomposes_into_eulerian_bicycles": (C.c_int, [vp]),
        "og_no_consecutive_dummy_edges": (C.c_int, [vp, u64]),
        "og_walks_free": (None, [P(Walks)]),
        "og_euler_cycles": (P(Walks), [vp]),
        "og_cut_cycles": (P(Walks), [vp, P(Walks), u64, P(u64)]),
        "og_compute_greedytigs": (P(Walks), [vp, u64, P(Stats)]),
        "og_compute_eulertigs": (P(Walks), [vp, u64]),
        "og_flatten_clib": (u64, [vp, P(Walks), P(i64), P(u64), P(u64)]),
        "og_clib_compute_tigs": (u64, [vp, u64, u64, P(i64), P(u64), P(u64)]),
        "og_write_walks_fasta": (vp, [vp, P(Walks), C.c_char_p, P(u64), u64, P(u64)]),
        "og_matching_instance": (vp, [vp, u64]),
        "og_matching_counts": (None, [vp, P(u64)]),
        "og_matching_write": (None, [vp, C.c_char_p]),
        "og_matching_apply": (P(Walks), [vp, vp, C.c_char_p, u64]),
        "og_matching_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _walks_to_lists(wp) -> list[list[int]]:
    w = wp.contents
    lim = np.ctypeslib.as_array(w.limits, shape=(max(int(w.n_walks), 1),))[: int(w.n_walks)]
    ed = np.ctypeslib.as_array(w.edges, shape=(max(int(w.n_edges), 1),))[: int(w.n_edges)]
    out, b = [], 0
    for l in lim:
        out.append([int(x) for x in ed[b:int(l)]])
        b = int(l)
    return out


class OracleGraph:
    """Owns an og_graph. Build from an explicit bigraph or from clib-style unitig links."""

    def __init__(self, handle):
        self.h = handle
        self.L = lib()

    def __del__(self):
        try:
            if self.h:
                self.L.og_graph_free(self.h)
                self.h = None
        except Exception:
            pass

    # ---- constructors ----
    @classmethod
    def from_bigraph(cls, n_nodes, mirror, edges):
        """edges: iterable of (from, to, weight, dummy_id, handle, forwards) in insertion order."""
        L = lib()
        g = L.og_graph_new(int(n_nodes))
        for a in range(int(n_nodes)):
            b = int(mirror[a])
            L.og_set_mirror_nodes(g, a, b)
        for (f, t, w, d, h, fw) in edges:
            L.og_add_edge(g, int(f), int(t), int(w), int(d), int(h), 1 if fw else 0)
        return cls(g)

    @classmethod
    def from_arrays(cls, mirror, e_from, e_to, e_weight):
        """Original-edge arrays in edge-id order; edge 2u forward of unitig u, edge 2u+1 its mirror."""
        L = lib()
        m = np.ascontiguousarray(mirror, dtype=np.uint32)
        f = np.ascontiguousarray(e_from, dtype=np.uint32)
        t = np.ascontiguousarray(e_to, dtype=np.uint32)
        w = np.ascontiguousarray(e_weight, dtype=np.uint64)
        vp = C.c_void_p
        g = L.og_graph_from_arrays(len(m), m.ctypes.data_as(vp), len(f), f.ctypes.data_as(vp), t.ctypes.data_as(vp),
                                   w.ctypes.data_as(vp))
        return cls(g)

    @classmethod
    def from_unitig_links(cls, unitig_weights, links):
        """links: iterable of (unitig_a, strand_a, unitig_b, strand_b) -- the clib.rs builder."""
        L = lib()
        w = np.ascontiguousarray(unitig_weights, dtype=np.uint64)
        b = L.og_builder_new(len(w))
        for (ua, sa, ub, sb) in links:
            L.og_builder_merge_nodes(b, int(ua), 1 if sa else 0, int(ub), 1 if sb else 0)
        g = L.og_builder_build(b, w.ctypes.data_as(C.POINTER(C.c_uint64)))
        return cls(g)

    @classmethod
    def from_unitig_links_arrays(cls, unitig_weights, links):
        """The same builder over an int array [n, 4] of links (one call instead of one per link)."""
        L = lib()
        w = np.ascontiguousarray(unitig_weights, dtype=np.uint64)
        lk = np.ascontiguousarray(links, dtype=np.int64)
        b = L.og_builder_new(len(w))
        L.og_builder_merge_nodes_many(b, len(lk), lk.ctypes.data_as(C.POINTER(C.c_int64)))
        g = L.og_builder_build(b, w.ctypes.data_as(C.POINTER(C.c_uint64)))
        return cls(g)

    # ---- accessors ----
    @property
    def node_count(self):
        return int(self.L.og_node_count(self.h))

    @property
    def edge_count(self):
        return int(self.L.og_edge_count(self.h))

    def mirror(self):
        return np.array([self.L.og_mirror_node(self.h, n) for n in range(self.node_count)], dtype=np.uint32)

    def edge(self, e):
        f, t = C.c_uint32(), C.c_uint32()
        w, d, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        fw = C.c_int()
        self.L.og_edge_get(self.h, e, C.byref(f), C.byref(t), C.byref(w), C.byref(d), C.byref(h), C.byref(fw))
        return (f.value, t.value, w.value, d.value, h.value, bool(fw.value))

    def edges(self):
        return [self.edge(e) for e in range(self.edge_count)]

    def edge_arrays(self):
        ed = self.edges()
        return (
            np.array([e[0] for e in ed], dtype=np.uint32),
            np.array([e[1] for e in ed], dtype=np.uint32),
            np.array([e[2] for e in ed], dtype=np.uint64),
        )

    def out_edges(self, n):
        buf = (C.c_uint32 * 4096)()
        c = self.L.og_out_edges(self.h, n, buf, 4096)
        return [int(buf[i]) for i in range(min(c, 4096))]

    def mirror_edge(self, e):
        return int(self.L.og_mirror_edge(self.h, e))

    # ---- algorithm stages ----
    def classify(self):
        n = self.node_count
        out = np.zeros(max(n, 1), dtype=np.uint32)
        live = np.zeros(max(n, 1), dtype=np.uint8)
        mult = np.zeros(max(n, 1), dtype=np.int64)
        ni, ns = C.c_uint32(), C.c_uint32()
        no = self.L.og_classify(
            self.h,
            out.ctypes.data_as(C.POINTER(C.c_uint32)),
            live.ctypes.data_as(C.POINTER(C.c_uint8)),
            mult.ctypes.data_as(C.POINTER(C.c_int64)),
            C.byref(ni),
            C.byref(ns),
        )
        return out[:no].copy(), live[:n].copy(), mult[:n].copy(), ni.value, ns.value

    def greedy_pairs(self, k, max_sources=None):
        pp = C.POINTER(Pair)()
        st = Stats()
        if max_sources is None:
            n = self.L.og_greedy_pairs(self.h, k, C.byref(pp), C.byref(st))
        else:
            n = self.L.og_greedy_pairs_prefix(self.h, k, int(max_sources), C.byref(pp), C.byref(st))
        pairs = [(pp[i].out_node, pp[i].in_node, int(pp[i].distance)) for i in range(n)]
        self.L.og_free(pp)
        return pairs, st.as_dict()

    def greedy_pairs_np(self, k, max_sources=None, threads=1):
        pp = C.POINTER(Pair)()
        st = Stats()
        ms = (1 << 64) - 1 if max_sources is None else int(max_sources)
        if threads > 1:   # reference-style worker threads: timing-dependent result, like the reference with -t > 1
            n = self.L.og_greedy_pairs_mt_prefix(self.h, k, int(threads), ms, C.byref(pp), C.byref(st))
        else:
            n = self.L.og_greedy_pairs_prefix(self.h, k, ms, C.byref(pp), C.byref(st))
        dt = np.dtype([("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])
        if n:
            arr = np.frombuffer((C.c_char * (n * C.sizeof(Pair))).from_address(C.addressof(pp.contents)), dtype=dt).copy()
        else:
            arr = np.zeros(0, dtype=dt)
        self.L.og_free(pp)
        return arr, st.as_dict()

    def candidate_lists(self, k, lo=0, hi=0xFFFFFFFF):
        on, off, keys = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        st = Stats()
        n = self.L.og_candidate_lists_range(self.h, k, lo, hi, C.byref(on), C.byref(off), C.byref(keys), C.byref(st))
        out_nodes = np.ctypeslib.as_array(on, shape=(max(n, 1),))[:n].copy()
        offsets = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
        nk = int(offsets[-1])
        ks = np.ctypeslib.as_array(keys, shape=(max(nk, 1),))[:nk].copy()
        self.L.og_free(on)
        self.L.og_free(off)
        self.L.og_free(keys)
        return out_nodes, offsets, ks, st.as_dict()

    def candidate_lists_given(self, k, sources, live):
        """og_candidate_lists_given: the full lists of `sources` (node ids of this graph) under the live map `live` the caller supplies
        (this graph may be a ball subgraph of a larger one, whose classification its own degrees cannot give). -> (offsets, keys)."""
        src = np.ascontiguousarray(sources, np.uint32)
        lv = np.ascontiguousarray(live, np.uint8)
        assert len(lv) == self.node_count
        off, keys = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        st = Stats()
        vp = C.c_void_p
        self.L.og_candidate_lists_given(self.h, k, src.ctypes.data_as(vp), len(src), lv.ctypes.data_as(vp), C.byref(off), C.byref(keys), C.byref(st))
        offsets = np.ctypeslib.as_array(off, shape=(len(src) + 1,)).copy()
        nk = int(offsets[-1])
        ks = np.ctypeslib.as_array(keys, shape=(max(nk, 1),))[:nk].copy()
        self.L.og_free(off)
        self.L.og_free(keys)
        return offsets, ks

    def greedy_pairs_given(self, k, out_nodes, live, mult):
        """og_greedy_pairs_given: the reference's claim loop over `out_nodes` (ascending) with the classification the caller supplies."""
        on = np.ascontiguousarray(out_nodes, np.uint32)
        lv = np.ascontiguousarray(live, np.uint8)
        mu = np.ascontiguousarray(mult, np.int64)
        assert len(lv) == len(mu) == self.node_count
        pp = C.POINTER(Pair)()
        st = Stats()
        vp = C.c_void_p
        n = self.L.og_greedy_pairs_given(self.h, k, on.ctypes.data_as(vp), len(on), lv.ctypes.data_as(vp), mu.ctypes.data_as(vp), C.byref(pp), C.byref(st))
        dt = np.dtype([("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])
        arr = np.frombuffer((C.c_char * (n * C.sizeof(Pair))).from_address(C.addressof(pp.contents)), dtype=dt).copy() if n else np.zeros(0, dtype=dt)
        self.L.og_free(pp)
        return arr, st.as_dict()

    def whole_path_timed(self, k, threads=1):
        """The reference's whole greedy path, stage by stage with wall-clock timers (bench.py cpu_baseline leg).
        Mutates the graph. Returns (seconds per stage, counters). threads > 1 runs the Dijkstra + claim stage with the
        reference's worker-thread scheme (the later stages are sequential in the reference as well)."""
        import time

        t = {}
        t0 = time.perf_counter()
        pairs, st = self.greedy_pairs_np(k, threads=threads)    # greedytigs/mod.rs:222-526 (classification + Dijkstra + claim)
        t["dijkstra_claim"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        arr = np.ascontiguousarray(pairs)
        did = self.L.og_insert_pair_edges(self.h, arr.ctypes.data_as(C.POINTER(Pair)), len(arr))   # :678-689
        d = C.c_uint64(did)
        self.L.og_make_eulerian_with_breaking_edges(self.h, C.byref(d), k)                          # :705
        t["eulerise"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        cyc = self.L.og_euler_cycles(self.h)                    # :722
        t["euler"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tigs = self.L.og_cut_cycles(self.h, cyc, k, None)       # :726-789
        t["cut"] = time.perf_counter() - t0
        n_tigs = int(tigs.contents.n_walks)
        self.L.og_walks_free(cyc)
        self.L.og_walks_free(tigs)
        return t, dict(st, pairs=len(arr), tigs=n_tigs)

    def insert_pair_edges(self, pairs):
        arr = (Pair * max(len(pairs), 1))()
        for i, (o, t, d) in enumerate(pairs):
            arr[i].out_node, arr[i].in_node, arr[i].distance = int(o), int(t), int(d)
        return int(self.L.og_insert_pair_edges(self.h, arr, len(pairs)))

    def make_eulerian(self, k, dummy_edge_id=0):
        d = C.c_uint64(dummy_edge_id)
        self.L.og_make_eulerian_with_breaking_edges(self.h, C.byref(d), k)
        return int(d.value)

    def is_eulerian(self):
        return bool(self.L.og_decomposes_into_eulerian_bicycles(self.h))

    def no_consecutive_dummy_edges(self, k):
        return bool(self.L.og_no_consecutive_dummy_edges(self.h, k))

    def euler_cycles(self):
        w = self.L.og_euler_cycles(self.h)
        out = _walks_to_lists(w)
        self.L.og_walks_free(w)
        return out

    def cut_cycles(self, cycles, k):
        """og_cut_cycles on given closed walks (lists of edge ids)."""
        limits, edges = [], []
        for c in cycles:
            edges.extend(c)
            limits.append(len(edges))
        la = (C.c_uint64 * max(len(limits), 1))(*limits)
        ea = (C.c_uint32 * max(len(edges), 1))(*edges)
        w = Walks(len(limits), len(edges), la, ea)
        t = self.L.og_cut_cycles(self.h, C.byref(w), k, None)
        out = _walks_to_lists(t)
        self.L.og_walks_free(t)
        return out

    def compute_greedytigs(self, k):
        st = Stats()
        w = self.L.og_compute_greedytigs(self.h, k, C.byref(st))
        out = _walks_to_lists(w)
        self.L.og_walks_free(w)
        return out, st.as_dict()

    def compute_eulertigs(self, k):
        w = self.L.og_compute_eulertigs(self.h, k)
        out = _walks_to_lists(w)
        self.L.og_walks_free(w)
        return out

    def clib_compute_tigs(self, algorithm, k):
        ec = self.edge_count
        eo = np.zeros(max(2 * ec, 1), dtype=np.int64)
        io = np.zeros(max(2 * ec, 1), dtype=np.uint64)
        lo = np.zeros(max(ec, 1), dtype=np.uint64)
        n = self.L.og_clib_compute_tigs(
            self.h, algorithm, k,
            eo.ctypes.data_as(C.POINTER(C.c_int64)),
            io.ctypes.data_as(C.POINTER(C.c_uint64)),
            lo.ctypes.data_as(C.POINTER(C.c_uint64)),
        )
        n = int(n)
        total = int(lo[n - 1]) if n else 0
        return n, eo[:total].copy(), io[:total].copy(), lo[:n].copy()

    def matching_instance(self, k):
        return OracleMatching(self, k)

    def fasta(self, tigs, seqs: list[str], k):
        """Spell tigs (list of edge-id lists into *this* graph) as FASTA text."""
        limits, edges = [], []
        for t in tigs:
            edges.extend(t)
            limits.append(len(edges))
        la = (C.c_uint64 * max(len(limits), 1))(*limits)
        ea = (C.c_uint32 * max(len(edges), 1))(*edges)
        w = Walks(len(limits), len(edges), la, ea)
        cat = "".join(seqs).encode()
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        ln = C.c_uint64()
        p = self.L.og_write_walks_fasta(self.h, C.byref(w), cat, off.ctypes.data_as(C.POINTER(C.c_uint64)), k, C.byref(ln))
        s = C.string_at(p, ln.value).decode()
        self.L.og_free(p)
        return s


class OracleMatching:
    """og_matching: the optimal-matchtigs matching instance (matchtigs/mod.rs:150-940 minus the matcher)."""

    NAMES = ["transformed_node_count", "edge_count", "wcc_amount", "matching_node_count", "matching_edge_count", "mirror_biedges",
             "mirror_expanded_biedges"]

    def __init__(self, graph: OracleGraph, k):
        self.L = lib()
        self.g = graph
        self.k = k
        self.h = self.L.og_matching_instance(graph.h, k)

    def stats(self) -> dict:
        a = (C.c_uint64 * 7)()
        self.L.og_matching_counts(self.h, a)
        return {n: int(a[i]) for i, n in enumerate(self.NAMES)}

    def write(self, path):
        self.L.og_matching_write(self.h, str(path).encode())

    def apply(self, solution_path):
        """Mutates the graph the instance was built from; returns the matchtigs."""
        w = self.L.og_matching_apply(self.h, self.g.h, str(solution_path).encode(), self.k)
        out = _walks_to_lists(w)
        self.L.og_walks_free(w)
        return out

    def __del__(self):
        try:
            if self.h:
                self.L.og_matching_free(self.h)
                self.h = None
        except Exception:
            pass
