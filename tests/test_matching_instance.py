"""Optimal matchtigs around the external matcher (matchtigs/mod.rs:150-940), CPU only: the oracle's literal
restatement against hand-derived instance files, against the independent Python restatement, and against the PRODUCT's host
stage (bulk, order-free construction from candidate lists) -- file bytes, statistics, and the matchtigs that come out of a
solution. The matcher itself is external in the reference (blossom5); tests/tools/tiny_matcher.py stands in for it."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers
import pyref
from matchtigs_amd import api, synth

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())
MATCHING_KATS = [k for k in KATS if "matching_instance" in k.get("expect", {})]
MATCHER = str(Path(__file__).parent / "tools" / "tiny_matcher.py")


def run_matcher(instance_path):
    r = subprocess.run([sys.executable, MATCHER, "-e", str(instance_path), "-w", str(instance_path) + ".solution"],
                       stderr=subprocess.DEVNULL)
    return r.returncode


def product_instance_from_oracle_lists(G, og, k):
    on, off, keys, _ = og.candidate_lists(k)
    _, _, mult, _, _ = og.classify()
    return api.MatchingInstance.from_lists(G, k, on, mult.astype(np.int32), off[:-1], np.diff(off).astype(np.uint32), keys)


@pytest.mark.parametrize("kat", MATCHING_KATS, ids=[k["name"] for k in MATCHING_KATS])
def test_hand_derived_instances(kat, oracle, product_lib, tmp_path):
    k, exp = kat["k"], kat["expect"]
    arrs = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    og = helpers.oracle_graph(*arrs)
    om = og.matching_instance(k)
    om.write(tmp_path / "o")
    assert (tmp_path / "o").read_text() == exp["matching_instance"]
    assert om.stats() == exp["matching_stats"]
    assert pyref.MatchingInstance(helpers.py_graph(*arrs), k).text() == exp["matching_instance"]
    pm = product_instance_from_oracle_lists(helpers.product_graph(*arrs), helpers.oracle_graph(*arrs), k)
    assert pm.write(tmp_path / "p") == len(exp["matching_instance"])
    assert (tmp_path / "p").read_text() == exp["matching_instance"]
    assert pm.stats() == exp["matching_stats"]


def test_every_kat_graph_three_way(oracle, product_lib, tmp_path):
    """All KAT graphs (two-strand graphs with separate WCCs, balanced graphs with an empty instance, ...)."""
    for i, kat in enumerate(k for k in KATS if "unitigs" in k):
        k = kat["k"]
        arrs = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
        og = helpers.oracle_graph(*arrs)
        om = og.matching_instance(k)
        om.write(tmp_path / f"o{i}")
        text = (tmp_path / f"o{i}").read_text()
        assert text == pyref.MatchingInstance(helpers.py_graph(*arrs), k).text(), kat["name"]
        pm = product_instance_from_oracle_lists(helpers.product_graph(*arrs), helpers.oracle_graph(*arrs), k)
        pm.write(tmp_path / f"p{i}")
        assert (tmp_path / f"p{i}").read_text() == text, kat["name"]
        assert pm.stats() == om.stats(), kat["name"]


def _cases():
    out = []
    for seed in range(1, 17):
        k = [5, 9, 31][seed % 3]
        out.append((seed, k, dict(n_binodes=40 + seed * 9, seed=seed, k=k, mean_out_degree=1.2 + 0.1 * (seed % 8),
                                  mean_weight=[2.0, 4.0, 8.0][seed % 3], self_mirror_frac=0.05)))
    return out


@pytest.mark.parametrize("seed,k,kw", _cases(), ids=[f"seed{c[0]}-k{c[1]}" for c in _cases()])
def test_three_way_on_random_bigraphs(seed, k, kw, oracle, product_lib, tmp_path):
    bg = synth.g_csr(**kw)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og, pg, G = helpers.oracle_graph(*arrs), helpers.py_graph(*arrs), helpers.product_graph(*arrs)
    om, ym = og.matching_instance(k), pyref.MatchingInstance(pg, k)
    pm = product_instance_from_oracle_lists(G, helpers.oracle_graph(*arrs), k)
    om.write(tmp_path / "o")
    pm.write(tmp_path / "p")
    text = (tmp_path / "o").read_text()
    assert text == ym.text()
    assert (tmp_path / "p").read_text() == text
    st = om.stats()
    assert pm.stats() == st
    lines = text.splitlines()
    assert lines[0] == f"{st['matching_node_count']} {st['matching_edge_count']}"
    assert len(lines) - 1 == st["matching_edge_count"]  # header edge count = lines written (node 0 always has an edge here)
    # the matcher's solution applied three ways
    assert run_matcher(tmp_path / "o") == 0
    sol = tmp_path / "o.solution"
    tigs_o = om.apply(sol)
    assert tigs_o == ym.apply(pg, sol.read_text())
    pairs = pm.read_solution(sol)
    assert api.MatchtigAlgorithm.finish(G, pairs, k) == tigs_o
    # every original unitig is covered by some matchtig, tigs start and end on original edges (matchtigs/mod.rs:929-933)
    oe = og.edges()
    covered = {oe[e][4] for t in tigs_o for e in t if oe[e][3] == 0}
    assert covered == set(range(bg.n_edges // 2))
    for t in tigs_o:
        assert oe[t[0]][3] == 0 and oe[t[-1]][3] == 0


def test_exactly_solved_small_instances(oracle, product_lib, tmp_path):
    """Instances of <= 22 nodes, solved exactly by the stand-in matcher: the solution applies identically three ways and its
    cost is what the matched pairs weigh (no comparison with greedy matchtigs: the four extra nodes per WCC force two unmatched
    ids per component, which the greedy path is not bound by)."""
    checked = 0
    for seed in range(1, 60):
        bg = synth.g_csr(n_binodes=14, seed=seed, k=5, mean_out_degree=1.3, mean_weight=2.0, self_mirror_frac=0.1)
        arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
        og, pg, G = helpers.oracle_graph(*arrs), helpers.py_graph(*arrs), helpers.product_graph(*arrs)
        om = og.matching_instance(5)
        st = om.stats()
        if not (0 < st["matching_node_count"] <= 22):
            continue
        om.write(tmp_path / f"i{seed}")
        if run_matcher(tmp_path / f"i{seed}") != 0:
            continue  # two-strand graph whose extra nodes cannot be matched (the reference's instance is infeasible there too)
        sol = tmp_path / f"i{seed}.solution"
        pm = product_instance_from_oracle_lists(G, helpers.oracle_graph(*arrs), 5)
        pairs = pm.read_solution(sol)
        assert all(1 <= int(p["dist"]) <= 4 for p in pairs)
        tigs = om.apply(sol)
        assert tigs == pyref.MatchingInstance(pg, 5).apply(pg, sol.read_text())
        assert api.MatchtigAlgorithm.finish(G, pairs, 5) == tigs
        checked += 1
    assert checked >= 5


def test_parallel_paths_on_a_larger_graph(oracle, product_lib, tmp_path):
    """Big enough that every bulk step of the product's construction runs on several host threads."""
    k = 9
    bg = synth.g_csr(n_binodes=400_000, seed=7, k=k, mean_out_degree=1.4, mean_weight=3.0, self_mirror_frac=0.01)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og, G = helpers.oracle_graph(*arrs), helpers.product_graph(*arrs)
    om = og.matching_instance(k)
    pm = product_instance_from_oracle_lists(G, helpers.oracle_graph(*arrs), k)
    om.write(tmp_path / "o")
    n = pm.write(tmp_path / "p")
    a, b = (tmp_path / "o").read_bytes(), (tmp_path / "p").read_bytes()
    assert n == len(b) and a == b
    assert pm.stats() == om.stats() and om.stats()["transformed_node_count"] > 65536
    assert run_matcher(tmp_path / "o") == 0
    pairs = pm.read_solution(tmp_path / "o.solution")
    assert len(pairs) > 1000
    assert api.MatchtigAlgorithm.finish(G, pairs, k) == om.apply(tmp_path / "o.solution")


def test_solution_with_an_unknown_edge_aborts(oracle, product_lib, tmp_path):
    kat = MATCHING_KATS[0]
    arrs = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    code = (
        "import sys; sys.path[:0] = %r\n"
        "import numpy as np, helpers\n"
        "from matchtigs_amd import api\n"
        "arrs = helpers.unitigs_to_arrays(%r, %r)\n"
        "og = helpers.oracle_graph(*arrs)\n"
        "on, off, keys, _ = og.candidate_lists(%d); mult = og.classify()[2]\n"
        "pm = api.MatchingInstance.from_lists(helpers.product_graph(*arrs), %d, on, mult.astype(np.int32), off[:-1],\n"
        "                                     np.diff(off).astype(np.uint32), keys)\n"
        "pm.read_solution(%r)\n"
    ) % (sys.path[:2], kat["mirror"], kat["unitigs"], kat["k"], kat["k"], str(tmp_path / "bad"))
    (tmp_path / "bad").write_text("14 7\n1 2\n")  # ids 1 and 2 are not joined by an edge in KAT-2's instance
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "Edge does not exist: (1, 2)" in r.stderr
