"""Hand-derived known-answer cases (tests/golden/kats.json) against the oracle, the independent Python
restatement and the product's host stages. CPU only."""
import json
from pathlib import Path

import numpy as np
import pytest

import helpers
import pyref

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())
UNITIG_KATS = [k for k in KATS if "unitigs" in k]


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
