"""Differential tests on seeded random bigraphs, CPU only:
   oracle (C) vs independent Python restatement vs the PRODUCT's host stages (replay, Euleriser,
   linked-list Hierholzer, cutter, clib flattening, clib builder)."""
import numpy as np
import pytest

import helpers
import pyref
from matchtigs_amd import api, synth


def _cases():
    out = []
    for seed in range(1, 25):
        k = [5, 9, 31][seed % 3]
        out.append((seed, k, dict(n_binodes=40 + seed * 9, seed=seed, k=k, mean_out_degree=1.2 + 0.1 * (seed % 8),
                                  mean_weight=[2.0, 4.0, 8.0][seed % 3], self_mirror_frac=0.05)))
    return out


@pytest.mark.parametrize("seed,k,kw", _cases(), ids=[f"seed{c[0]}-k{c[1]}" for c in _cases()])
def test_three_way_agreement(seed, k, kw, oracle, product_lib):
    bg = synth.g_csr(**kw)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = helpers.oracle_graph(*arrs)
    pairs_o, st_o = og.greedy_pairs(k)
    pairs_p, st_p = pyref.greedy_pairs(helpers.py_graph(*arrs), k)
    assert pairs_o == pairs_p
    assert st_o["relaxed_edges"] == st_p["relaxed_edges"] and st_o["settled_nodes"] == st_p["settled_nodes"]
    # product replay on oracle candidate lists == oracle's truncated-Dijkstra claim loop (T2)
    G = helpers.product_graph(*arrs)
    pr = helpers.product_pairs_from_oracle_lists(G, helpers.oracle_graph(*arrs), k)
    assert [(int(a), int(b), int(c)) for a, b, c in pr] == pairs_o
    # full tig pipelines (T3/T4)
    og2 = helpers.oracle_graph(*arrs)
    tigs_o, _ = og2.compute_greedytigs(k)
    tigs_p, _, _ = pyref.compute_greedytigs(helpers.py_graph(*arrs), k)
    assert tigs_o == tigs_p
    assert G.finish_greedytigs(pr, k) == tigs_o
    # the product's mutated graph equals the oracle's (same dummy edges in the same order)
    ex = G.export()
    oe = og2.edges()
    assert len(oe) == len(ex["edge_from"])
    assert [e[0] for e in oe] == ex["edge_from"].tolist() and [e[1] for e in oe] == ex["edge_to"].tolist()
    assert [e[2] for e in oe] == ex["edge_weight"].tolist() and [e[3] for e in oe] == ex["edge_dummy_id"].tolist()
    # invariants lifted from the reference's asserts
    assert og2.is_eulerian() and og2.no_consecutive_dummy_edges(k)
    for t in tigs_o:
        assert oe[t[0]][3] == 0 and oe[t[-1]][3] == 0            # greedytigs/mod.rs:794-798
    for e in oe:
        if e[3] != 0:
            assert (1 <= e[2] <= k - 1) or e[2] == k               # matched dummies <= k-1, breaking == k
    # clib flattening
    eo, io, lim = G.flatten_clib(tigs_o)
    peo, pio, plim = pyref.flatten_clib(helpers.py_graph(*arrs) if False else _pg_after(arrs, k), tigs_o)
    assert (eo, io, lim) == (peo, pio, plim)
    # reset restores the original graph exactly
    G.reset()
    ex2 = G.export()
    assert ex2["edge_from"].tolist() == bg.edge_from.tolist() and G.edge_count() == bg.n_edges


def _pg_after(arrs, k):
    g = helpers.py_graph(*arrs)
    pyref.compute_greedytigs(g, k)
    return g


@pytest.mark.parametrize("seed", range(1, 13))
def test_eulertigs_three_way(seed, oracle, product_lib):
    k = [5, 9, 31][seed % 3]
    bg = synth.g_csr(60 + 11 * seed, seed=100 + seed, k=k, mean_out_degree=1.0 + 0.15 * (seed % 7), mean_weight=3.0,
                     self_mirror_frac=0.08)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    et_o = helpers.oracle_graph(*arrs).compute_eulertigs(k)
    et_p = pyref.compute_eulertigs(helpers.py_graph(*arrs), k)
    et = api.EulertigAlgorithm.compute_tigs(helpers.product_graph(*arrs), api.EulertigAlgorithmConfiguration(k))
    assert et_o == et_p == et


def test_euler_cycles_linked_list_equals_literal_on_larger_graph(oracle, product_lib):
    """The product's O(E) splice formulation vs the oracle's literal rotate_left formulation, 60k edges."""
    k = 31
    bg = synth.g_csr(20000, seed=77, k=k)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = helpers.oracle_graph(*arrs)
    pairs, _ = og.greedy_pairs_np(k)
    G = helpers.product_graph(*arrs)
    og.insert_pair_edges([(int(a), int(b), int(c)) for a, b, c in pairs])
    G.insert_pair_edges(pairs)
    d1 = og.make_eulerian(k, len(pairs))
    d2 = G.make_eulerian(len(pairs), k)
    assert d1 == d2
    want_cycles = og.euler_cycles()
    assert want_cycles == G.euler_cycles()
    assert want_cycles == G.euler_cycles_records(1)   # 32-byte records (the memory-lean walk, euler_lean.cpp)
    assert want_cycles == G.euler_cycles_records(2)   # 256-byte records seeded from 32-byte ones (the device finish's route)
    assert want_cycles == G.euler_cycles_records(3)   # 128-byte records (two levels), for graphs too large for the 256-byte ones


def test_euler_walk_scratch_is_reused_between_calls_on_one_graph(oracle, product_lib):
    """The walk's large scratch mappings stay with the graph (huge_arena.hpp) and come back with whatever the last call left
    in them: a second and third call on the same graph -- same size, then a different Eulerisation -- must not see any of it."""
    k = 31
    bg = synth.g_csr(20000, seed=78, k=k)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = helpers.oracle_graph(*arrs)
    G = helpers.product_graph(*arrs)
    og.make_eulerian(k, 0)
    G.make_eulerian(0, k)
    want = og.euler_cycles()
    assert G.euler_cycles() == want   # fresh mappings
    assert G.euler_cycles() == want   # reused mappings, identical graph
    G.reset()
    og2 = helpers.oracle_graph(*arrs)
    pairs, _ = og2.greedy_pairs_np(k)
    og2.insert_pair_edges([(int(a), int(b), int(c)) for a, b, c in pairs])
    G.insert_pair_edges(pairs)
    assert og2.make_eulerian(k, len(pairs)) == G.make_eulerian(len(pairs), k)
    assert G.euler_cycles() == og2.euler_cycles()   # reused (or regrown) mappings, other edge set


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clib_builder_matches_oracle_and_pyref(seed, oracle, product_lib):
    ug = synth.g_seq(1500, seed=seed, k=11, haplotypes=3, sub_rate=0.04)
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    pg = pyref.from_unitig_links([int(x) for x in ug.weights], ug.links)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    ex = G.export()
    assert og.node_count == pg.n == G.node_count()
    assert og.mirror().tolist() == pg.mirror == ex["mirror"].tolist()
    oe = og.edges()
    assert [e[0] for e in oe] == [e.frm for e in pg.edges] == ex["edge_from"].tolist()
    assert [e[1] for e in oe] == [e.to for e in pg.edges] == ex["edge_to"].tolist()
    assert [e[4] for e in oe] == ex["edge_unitig"].tolist()
    assert [e[5] for e in oe] == [bool(x) for x in ex["edge_forwards"]]
    # eulertigs and unitigs through the real C-ABI (no GPU needed for ids 3 and 1)
    for alg in (3, 1):
        n, eo, io, lo = api.clib_compute_tigs(ug.weights, ug.links, alg, 1, ug.k)
        n_o, eo_o, io_o, lo_o = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links).clib_compute_tigs(alg, ug.k)
        assert n == n_o and np.array_equal(eo, eo_o) and np.array_equal(io, io_o) and np.array_equal(lo, lo_o)


@pytest.mark.parametrize("kind", ["dbg_like", "random"])
def test_clib_builder_on_host_threads_matches_oracle(kind, oracle, product_lib):
    """matchtigs_build_graph at a size where every pass runs on several host threads (clib.rs:186-259): the links are recorded and
    united at build time, with prefetching, in call order; numbering, mirror assignment and edges are parallel passes. `random` links
    give components without any structure, union-find chains deeper than a de Bruijn graph's, and nodes that are their own mirror
    (the mirror assignment's conflict path)."""
    from matchtigs_amd import synth
    U = 150_000
    rng = np.random.default_rng(11)
    if kind == "dbg_like":
        links = synth.dbg_like_links(U, seed=3)
    else:
        n = int(0.8 * U)  # (the oracle's builder is quadratic in the size of a node: no giant component)
        links = np.stack([rng.integers(0, U, n), rng.integers(0, 2, n), rng.integers(0, U, n), rng.integers(0, 2, n)], axis=1).astype(np.int64)
        links[::7, 2] = links[::7, 0]       # a unitig linked with itself ...
        links[::14, 3] = 1 - links[::14, 1]  # ... on the other strand: nodes that are their own mirror
    weights = rng.integers(1, 50, U).astype(np.uint64)
    og = oracle.OracleGraph.from_unitig_links_arrays(weights, links)
    G = api.Bigraph.from_unitig_links_arrays(weights, links)
    ex = G.export()
    assert og.node_count == G.node_count() and og.edge_count == G.edge_count()
    assert np.array_equal(og.mirror(), ex["mirror"])
    oe = og.edges()
    assert np.array_equal(np.array([e[0] for e in oe], np.uint32), ex["edge_from"])
    assert np.array_equal(np.array([e[1] for e in oe], np.uint32), ex["edge_to"])
    assert np.array_equal(np.array([e[2] for e in oe], np.uint64), ex["edge_weight"])
    assert np.array_equal(np.array([e[4] for e in oe], np.uint64), ex["edge_unitig"])
    assert np.array_equal(np.array([e[5] for e in oe], np.uint8), ex["edge_forwards"])


def test_clib_flattening_on_host_threads_equals_oracle(oracle, product_lib):
    """clib.rs:393-407 at a size where the flattening runs on several host threads (> 2^16 walk edges): eulertigs of a real
    de Bruijn graph through matchtigs_compute_tigs, all three output arrays against the oracle's."""
    ua = synth.g_seq_arrays(1_200_000, seed=5, k=21)
    links = [tuple(int(x) for x in l) for l in ua.links.tolist()] if hasattr(ua.links, "tolist") else ua.links
    n, eo, io, lo = api.clib_compute_tigs(ua.weights, links, 3, 1, 21)
    assert len(eo) > (1 << 16)
    n_o, eo_o, io_o, lo_o = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links).clib_compute_tigs(3, 21)
    assert n == n_o and np.array_equal(eo, eo_o) and np.array_equal(io, io_o) and np.array_equal(lo, lo_o)


def test_edge_cases_empty_and_balanced(oracle, product_lib):
    # empty graph
    G = api.Bigraph.from_edges(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint64))
    assert api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(31)) == []
    # one isolated unitig: nodes 0,1 / mirrors 3,2 ... unitig 0->1, mirror 2... use 2 binodes
    mirror, frm, to, w = helpers.unitigs_to_arrays([1, 0, 3, 2], [[0, 2, 7]])
    et = api.EulertigAlgorithm.compute_tigs(helpers.product_graph(mirror, frm, to, w), api.EulertigAlgorithmConfiguration(5))
    assert et == helpers.oracle_graph(mirror, frm, to, w).compute_eulertigs(5)
    assert len(et) == 1 and len(et[0]) == 1
    # a perfectly balanced cycle of three unitigs: no imbalance, no dummy edges, one circular tig (cut nowhere)
    mirror, frm, to, w = helpers.unitigs_to_arrays([1, 0, 3, 2, 5, 4], [[0, 2, 3], [2, 4, 3], [4, 0, 3]])
    G = helpers.product_graph(mirror, frm, to, w)
    et = api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(5))
    assert et == helpers.oracle_graph(mirror, frm, to, w).compute_eulertigs(5)
    assert G.edge_count() == 6 and len(et) == 1 and len(et[0]) == 3


def test_long_cycle_parallel_cutter_equals_oracle(oracle, product_lib):
    """A closed walk above 2^20 biedges takes the threaded cutter (host_pipeline.cpp); same tigs as the oracle's."""
    from matchtigs_amd import api, synth

    bg = synth.g_csr(750000, seed=3, k=31)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    L = api._lib.load()
    lim, ed = api._take_walks_np(L, L.mtg_compute_eulertigs(G.handle, 31))
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    want = og.compute_eulertigs(31)
    assert np.array_equal(np.cumsum([len(t) for t in want]), lim)
    assert np.array_equal(np.array([e for t in want for e in t], dtype=np.uint32), ed)


def test_oracle_worker_thread_variant_keeps_the_invariants(oracle):
    """og_greedy_pairs_mt (the reference's -t > 1 scheme, used only for the bench's multi-core timing) is timing-dependent,
    so it is checked the way the reference checks itself: matched distances in [1, k-1], the graph Eulerian after
    Eulerisation, tigs covering every unitig once -- and close to the 1-thread pair count."""
    from matchtigs_amd import synth

    bg = synth.g_csr(40000, seed=6, k=31)
    og1 = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    p1, _ = og1.greedy_pairs_np(31)
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    pm, _ = og.greedy_pairs_np(31, threads=4)
    assert abs(len(pm) - len(p1)) <= max(8, len(p1) // 50)
    assert pm["dist"].min() >= 1 and pm["dist"].max() <= 30
    og.insert_pair_edges(list(zip(pm["out"].tolist(), pm["in"].tolist(), pm["dist"].tolist())))
    og.make_eulerian(31, len(pm))
    assert og.is_eulerian() and og.no_consecutive_dummy_edges(31)


@pytest.mark.parametrize("seed,deg,maxdeg", [(4, 5.0, 9), (7, 3.2, 6), (9, 7.0, 12)])
def test_euler_walk_high_degree_nodes_equal_oracle(oracle, product_lib, seed, deg, maxdeg):
    """Nodes with more out-edges than the Euler records copy inline (3 own positions, 3 + 2 copied): the spill arrays and the
    `more` fallbacks of euler_fast.cpp must give the literal algorithm's walks."""
    bg = synth.g_csr(3000, seed=seed, k=15, mean_out_degree=deg, mean_weight=4.0, max_degree=maxdeg, self_mirror_frac=0.02)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, 15)
    og.make_eulerian(15)
    assert G.euler_cycles() == og.euler_cycles()
    G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og2 = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    assert api.EulertigAlgorithm.compute_tigs(G2, api.EulertigAlgorithmConfiguration(15)) == og2.compute_eulertigs(15)


@pytest.mark.parametrize("seed, deg", [(3, 1.5), (4, 1.15)])
def test_walk_forms_agree_beyond_the_long_splice_scans(seed, deg, product_lib):
    """From 2^18 entries on, the 256- / 128-byte-record walk (euler_fast.cpp) narrows a splice scan down with a bitmap of the nodes that
    can still have an unused out-edge, and it emits every cycle run by run; the 32-byte-record walk (euler_lean.cpp) does neither.
    Same closed walks on Eulerised graphs of ~1.5 M darts (the sparser one falls into many components and splices)."""
    k = 31
    bg = synth.g_csr(360000, seed=seed, k=k, mean_out_degree=deg)
    G = helpers.product_graph(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, k)
    lean = G.euler_cycles_records(1)
    assert sum(len(c) for c in lean) == G.edge_count() // 2 >= (1 << 18)
    assert lean == G.euler_cycles()
    assert lean == G.euler_cycles_records(3)
