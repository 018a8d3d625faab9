"""Optimal matchtigs (tig algorithm 4) through the HIP path: the all-targets bounded searches run as the SSSP
kernels, the matching instance is collapsed from their candidate lists, the external matcher runs as a child process of the
library, its solution is applied. Checked against the hand-derived instance files, the oracle's literal restatement of
matchtigs/mod.rs:150-940, and through the reference's own C-ABI (clib.rs:362-376). tests/tools/tiny_matcher.py stands in for
blossom5 (external and not redistributable in the reference as well)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())
MATCHING_KATS = [k for k in KATS if "matching_instance" in k.get("expect", {})]
MATCHER = str(Path(__file__).parent / "tools" / "tiny_matcher.py")


@pytest.fixture(scope="module")
def gpu(product_lib):
    import torch

    if product_lib.mtg_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("these tests need a GPU: the matchtigs_amd hot path has no CPU fallback")
    return torch


@pytest.mark.parametrize("kat", [k for k in KATS if "unitigs" in k], ids=[k["name"] for k in KATS if "unitigs" in k])
def test_kat_instances_through_hip(kat, gpu, oracle, tmp_path):
    from matchtigs_amd import api

    k = kat["k"]
    arrs = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    pm = api.MatchingInstance(helpers.product_graph(*arrs), k)
    pm.write(tmp_path / "p")
    text = (tmp_path / "p").read_text()
    om = helpers.oracle_graph(*arrs).matching_instance(k)
    om.write(tmp_path / "o")
    assert text == (tmp_path / "o").read_text()
    assert pm.stats() == om.stats()
    if "matching_instance" in kat["expect"]:
        assert text == kat["expect"]["matching_instance"] and pm.stats() == kat["expect"]["matching_stats"]


@pytest.mark.parametrize("seed", range(1, 9))
def test_random_bigraphs_whole_algorithm(seed, gpu, oracle, tmp_path):
    from matchtigs_amd import api, synth

    k = [5, 9, 31][seed % 3]
    bg = synth.g_csr(n_binodes=300 + seed * 211, seed=seed, k=k, mean_out_degree=1.2 + 0.1 * (seed % 8),
                     mean_weight=[2.0, 4.0, 8.0][seed % 3], self_mirror_frac=0.05)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G, og = helpers.product_graph(*arrs), helpers.oracle_graph(*arrs)
    prefix = tmp_path / "m"
    cfg = api.MatchtigAlgorithmConfiguration(threads=1, k=k, matching_file_prefix=str(prefix), matcher_path=MATCHER)
    tigs = api.MatchtigAlgorithm.compute_tigs(G, cfg)
    # the library wrote the instance, ran the matcher as a child process and read its solution
    inst, sol = Path(str(prefix) + ".minimalperfectmatching"), Path(str(prefix) + ".minimalperfectmatching.solution")
    om = og.matching_instance(k)
    om.write(tmp_path / "o")
    assert inst.read_bytes() == (tmp_path / "o").read_bytes()
    assert sol.exists()
    assert tigs == om.apply(sol)
    ex, oe = G.export(), og.edges()
    assert [e[0] for e in oe] == ex["edge_from"].tolist() and [e[3] for e in oe] == ex["edge_dummy_id"].tolist()
    covered = {oe[e][4] for t in tigs for e in t if oe[e][3] == 0}
    assert covered == set(range(bg.n_edges // 2))


def test_larger_graph_instance_bytes_and_matchtigs(gpu, oracle, tmp_path):
    from matchtigs_amd import api, synth

    k = 31
    bg = synth.g_csr(n_binodes=400_000, seed=11, k=k)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G, og = helpers.product_graph(*arrs), helpers.oracle_graph(*arrs)
    pm = api.MatchingInstance(G, k)
    om = og.matching_instance(k)
    n = pm.write(tmp_path / "p")
    om.write(tmp_path / "o")
    a, b = (tmp_path / "o").read_bytes(), (tmp_path / "p").read_bytes()
    assert n == len(b) and a == b
    st = pm.stats()
    assert st == om.stats() and st["transformed_node_count"] > 65536
    r = subprocess.run([sys.executable, MATCHER, "-e", str(tmp_path / "p"), "-w", str(tmp_path / "p.solution")])
    assert r.returncode == 0
    pairs = pm.read_solution(tmp_path / "p.solution")
    assert len(pairs) > 1000
    assert api.MatchtigAlgorithm.finish(G, pairs, k) == om.apply(tmp_path / "p.solution")


def test_real_dbg_through_the_reference_cabi(gpu, oracle, tmp_path):
    """clib.rs route with tig_algorithm 4 (clib.rs:362-376): unitig links in, flattened matchtigs out."""
    from matchtigs_amd import api, synth

    k = 31
    ua = synth.g_seq_arrays(300_000, seed=3, k=k, haplotypes=4, sub_rate=0.02)
    prefix = tmp_path / "clib"
    links = [(int(a), bool(b), int(c), bool(d)) for a, b, c, d in ua.links]
    n, eo, io, lo = api.clib_compute_tigs(ua.weights, links, 4, 1, k, matching_file_prefix=str(prefix), matcher_path=MATCHER)
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    om = og.matching_instance(k)
    om.write(tmp_path / "o")
    assert Path(str(prefix) + ".minimalperfectmatching").read_bytes() == (tmp_path / "o").read_bytes()
    tigs = om.apply(str(prefix) + ".minimalperfectmatching.solution")
    oe = og.edges()
    want_e, want_i, want_l = [], [], []
    for t in tigs:  # clib.rs:393-407
        for e in t:
            want_e.append(oe[e][4] * (1 if oe[e][5] else -1))
            want_i.append(oe[e][2] if oe[e][3] else 0)
        want_l.append(len(want_e))
    assert n == len(tigs) and lo.tolist() == want_l
    assert eo.tolist() == want_e and io.tolist() == want_i
    assert n < ua.n_unitigs


def test_cli_matchtigs_fasta_equals_oracle(gpu, oracle, tmp_path):
    """bin.rs route: --bcalm-in + --matchtigs-fa-out + --blossom5-command, FASTA spelled on the GPU."""
    from matchtigs_amd import synth

    k = 31
    ua = synth.g_seq_arrays(200_000, seed=5, k=k, haplotypes=4, sub_rate=0.02)
    inp, out = tmp_path / "unitigs.fa", tmp_path / "matchtigs.fa"
    inp.write_bytes(ua.bcalm2_text())
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "matchtigs_amd", "--bcalm-in", str(inp), "-k", str(k), "--matchtigs-fa-out", str(out),
                        "--blossom5-command", MATCHER], capture_output=True, text=True, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    om = og.matching_instance(k)
    om.write(tmp_path / "o")
    assert Path(str(out) + ".minimalperfectmatching").read_bytes() == (tmp_path / "o").read_bytes()  # bin.rs:1146-1149
    tigs = om.apply(str(out) + ".minimalperfectmatching.solution")
    want = og.fasta(tigs, ua.unitig_list(), k).encode()
    fa = out.read_bytes()
    assert len(fa) == len(want) and fa == want


def test_matcher_failure_aborts_like_the_reference(gpu, tmp_path):
    """A matcher that exits non-zero: "Matcher was unsuccessful" (matchtigs/mod.rs:735), i.e. the process aborts."""
    kat = MATCHING_KATS[0]  # two-strand graph: its instance has no perfect matching, the stand-in matcher exits 1
    code = (
        "import sys; sys.path[:0] = %r\n"
        "import helpers\n"
        "from matchtigs_amd import api\n"
        "arrs = helpers.unitigs_to_arrays(%r, %r)\n"
        "cfg = api.MatchtigAlgorithmConfiguration(1, %d, %r, %r)\n"
        "api.MatchtigAlgorithm.compute_tigs(helpers.product_graph(*arrs), cfg)\n"
    ) % (sys.path[:2], kat["mirror"], kat["unitigs"], kat["k"], str(tmp_path / "f"), MATCHER)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "Matcher was unsuccessful" in r.stderr
    assert os.path.exists(str(tmp_path / "f") + ".minimalperfectmatching")


def test_bare_matcher_name_is_found_through_path(gpu, oracle, tmp_path, monkeypatch):
    """The reference starts the matcher with Command::new(matcher_path) (matchtigs/mod.rs:727), which resolves a bare name such as
    "blossom5" through PATH; so does the library (posix_spawnp), here through the reference's own C-ABI with algorithm 4."""
    from matchtigs_amd import api, synth

    bindir = tmp_path / "bin"
    bindir.mkdir()
    exe = bindir / "blossom5-stand-in"
    exe.write_text(f"#!/bin/sh\nexec {sys.executable} {MATCHER} \"$@\"\n")
    exe.chmod(0o755)
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}{os.environ.get('PATH', '')}")
    k = 31
    ua = synth.g_seq_arrays(60_000, seed=5, k=k, haplotypes=4, sub_rate=0.02)
    links = [(int(a), bool(b), int(c), bool(d)) for a, b, c, d in ua.links]
    prefix = tmp_path / "bare"
    n, eo, io, lo = api.clib_compute_tigs(ua.weights, links, 4, 1, k, matching_file_prefix=str(prefix), matcher_path="blossom5-stand-in")
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    om = og.matching_instance(k)
    tigs = om.apply(str(prefix) + ".minimalperfectmatching.solution")
    assert n == len(tigs) and int(lo[-1]) == sum(len(t) for t in tigs)
