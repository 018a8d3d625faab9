"""Hand-derived known-answer cases (tests/golden/kats.json) against the oracle, the independent Python
restatement and the product's host stages. CPU only."""
import json
from pathlib import Path

import numpy as np
import pytest

import helpers
import pyref

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())
UNITIG_KATS = [k for k in KATS if "unitigs" in k and "pairs" in k["expect"]]


def test_kat_e_euleriser(oracle, product_lib):
    kat = next(k for k in KATS if "raw_edges" in k)
    k, mirror, exp = kat["k"], kat["mirror"], kat["expect"]
    og = oracle.OracleGraph.from_bigraph(len(mirror), mirror, [tuple(e) for e in kat["raw_edges"]])
    assert og.make_eulerian(k, kat["euleriser_start_dummy_id"]) == exp["final_dummy_id"]
    got = [[e[0], e[1], e[2], e[3], e[5]] for e in og.edges()[len(kat["raw_edges"]):]]
    assert got == exp["breaking_edges"]
    assert og.is_eulerian()
    pg = pyref.PyBigraph(len(mirror), mirror)
    for e in kat["raw_edges"]:
        pg.add_edge(*e)
    assert pyref.make_eulerian(pg, kat["euleriser_start_dummy_id"], k) == exp["final_dummy_id"]
    assert [[e.frm, e.to, e.weight, e.dummy_id, e.forwards] for e in pg.edges[8:]] == exp["breaking_edges"]


@pytest.mark.parametrize("kat", UNITIG_KATS, ids=[k["name"] for k in UNITIG_KATS])
def test_kat_pairs(kat, oracle, product_lib):
    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    want = [tuple(p) for p in exp["pairs"]]
    og = helpers.oracle_graph(mirror, frm, to, w)
    if "out_nodes" in exp:
        on, live, mult, _, _ = og.classify()
        assert on.tolist() == exp["out_nodes"]
        assert mult.tolist() == exp["multiplicity"]
    pairs, _ = og.greedy_pairs(k)
    assert pairs == want, "oracle"
    ppairs, _ = pyref.greedy_pairs(helpers.py_graph(mirror, frm, to, w), k)
    assert ppairs == want, "pyref"
    G = helpers.product_graph(mirror, frm, to, w)
    pr = helpers.product_pairs_from_oracle_lists(G, og, k)
    assert [(int(a), int(b), int(c)) for a, b, c in pr] == want, "product replay"


def test_kat1_tigs(oracle, product_lib):
    from matchtigs_amd import api

    kat = next(k for k in KATS if k["name"].startswith("KAT-1"))
    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    og = helpers.oracle_graph(mirror, frm, to, w)
    tigs, _ = og.compute_greedytigs(k)
    weights = [e[2] for e in og.edges()]
    assert len(tigs) == exp["greedy_tig_count"]
    assert helpers.cumulative_length(tigs, weights, k) == exp["greedy_cumulative_length"]
    brk = [[e[0], e[1]] for e in og.edges() if e[3] != 0 and e[2] >= k and e[5]]
    assert brk == exp["greedy_breaking_edges"]
    og2 = helpers.oracle_graph(mirror, frm, to, w)
    et = og2.compute_eulertigs(k)
    assert len(et) == exp["euler_tig_count"]
    assert helpers.cumulative_length(et, [e[2] for e in og2.edges()], k) == exp["euler_cumulative_length"]
    # product eulertigs (host-only path) agrees edge for edge
    G = helpers.product_graph(mirror, frm, to, w)
    assert api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k)) == et
    # product greedy host stages on the hand-derived pair list
    G2 = helpers.product_graph(mirror, frm, to, w)
    pr = np.array([tuple(p) for p in exp["pairs"]], dtype=[("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])
    assert G2.finish_greedytigs(pr, k) == tigs


T4_KATS = [k for k in KATS if "unitigs" in k and any(x in k["expect"] for x in ("greedy_tigs", "euler_tigs"))]


@pytest.mark.parametrize("kat", T4_KATS, ids=[k["name"] for k in T4_KATS])
def test_kat_t4_euler_order_and_cut(kat, oracle, product_lib):
    """T4: hand-derived Euler cycles (walk order policies of App. A.2/A.3) and tigs (rotation + cut, greedytigs/mod.rs:726-789)
    against the oracle, the independent Python restatement and the product's host stages."""
    from matchtigs_amd import api

    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    pairs = [tuple(p) for p in exp["pairs"]]
    pr = np.array(pairs, dtype=[("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])
    if "greedy_tigs" in exp:
        og = helpers.oracle_graph(mirror, frm, to, w)
        did = og.insert_pair_edges(pairs)
        og.make_eulerian(k, did)
        assert og.euler_cycles() == exp["greedy_euler_cycles"], "oracle cycles"
        assert og.cut_cycles(exp["greedy_euler_cycles"], k) == exp["greedy_tigs"], "oracle cut"
        pg = helpers.py_graph(mirror, frm, to, w)
        pyref.make_eulerian(pg, pyref.insert_pair_edges(pg, pairs), k)
        assert pyref.euler_cycles(pg) == exp["greedy_euler_cycles"], "pyref cycles"
        assert pyref.cut_cycles(pg, exp["greedy_euler_cycles"], k) == exp["greedy_tigs"], "pyref cut"
        G = helpers.product_graph(mirror, frm, to, w)
        G.make_eulerian(G.insert_pair_edges(pr), k)
        assert G.euler_cycles() == exp["greedy_euler_cycles"], "product cycles"
        assert G.euler_cycles_records(1) == exp["greedy_euler_cycles"] and G.euler_cycles_records(2) == exp["greedy_euler_cycles"]
        assert G.euler_cycles_records(3) == exp["greedy_euler_cycles"]
        assert G.cut_cycles(exp["greedy_euler_cycles"], k) == exp["greedy_tigs"], "product cut"
        G2 = helpers.product_graph(mirror, frm, to, w)
        assert G2.finish_greedytigs(pr, k) == exp["greedy_tigs"], "product finish_greedytigs"
    if "euler_tigs" in exp:
        og = helpers.oracle_graph(mirror, frm, to, w)
        og.make_eulerian(k, 0)
        assert og.euler_cycles() == exp["euler_euler_cycles"]
        assert helpers.oracle_graph(mirror, frm, to, w).compute_eulertigs(k) == exp["euler_tigs"]
        assert pyref.compute_eulertigs(helpers.py_graph(mirror, frm, to, w), k) == exp["euler_tigs"]
        G = helpers.product_graph(mirror, frm, to, w)
        assert api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k)) == exp["euler_tigs"]


def test_kat_c_cutter(oracle, product_lib):
    kat = next(k for k in KATS if k["name"].startswith("KAT-C"))
    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    pairs = [tuple(p) for p in kat["dummy_pairs"]]
    og = helpers.oracle_graph(mirror, frm, to, w)
    og.insert_pair_edges(pairs)
    assert og.cut_cycles(kat["cycles"], k) == exp["cut_tigs"], "oracle"
    pg = helpers.py_graph(mirror, frm, to, w)
    pyref.insert_pair_edges(pg, pairs)
    assert pyref.cut_cycles(pg, [list(c) for c in kat["cycles"]], k) == exp["cut_tigs"], "pyref"
    G = helpers.product_graph(mirror, frm, to, w)
    G.insert_pair_edges(np.array(pairs, dtype=[("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)]))
    assert G.cut_cycles(kat["cycles"], k) == exp["cut_tigs"], "product"


CLIB_EULER_KATS = [k for k in KATS if "clib" in k.get("expect", {}) and k["expect"]["pairs"] == []]


@pytest.mark.parametrize("kat", CLIB_EULER_KATS, ids=[k["name"] for k in CLIB_EULER_KATS])
def test_kat_clib_route_eulertigs(kat, oracle, product_lib):
    """The clib.rs route (initialise_graph -> merge_nodes -> build_graph -> compute_tigs(3)) on the balanced KATs: the
    flattened output arrays (signed unitig ids, inserts, exclusive limits; clib.rs:393-407) typed in by hand."""
    from matchtigs_amd import api

    exp = kat["expect"]["clib"]
    uw = np.array([u[2] for u in kat["unitigs"]], dtype=np.uint64)
    links = helpers.links_of_bigraph(kat["mirror"], kat["unitigs"])
    n, eo, io, lo = api.clib_compute_tigs(uw, links, 3, 1, kat["k"])
    assert (n, eo.tolist(), io.tolist(), lo.tolist()) == (len(exp["tigs_out_limits"]), exp["tigs_edge_out"], exp["tigs_insert_out"], exp["tigs_out_limits"])
    n2, eo2, io2, lo2 = oracle.OracleGraph.from_unitig_links(uw, links).clib_compute_tigs(3, kat["k"])
    assert (n2, eo2.tolist(), io2.tolist(), lo2.tolist()) == (n, eo.tolist(), io.tolist(), lo.tolist())
