"""The hand-derived known-answer cases (tests/golden/kats.json) through the HIP path: DeviceGraph -> mtg_classify ->
mtg_sssp_candidates -> mtg_replay_claims_device -> mtg_finish_greedytigs, and the one-shot operator / clib.rs C-ABI.
These pin, on the GPU kernels themselves, the tie-break by node index, the inclusive bound, the mirror-of-self
candidate rule, the dead-target rule (T1/T2) and the Euler walk order + rotation/cut policies (T4)."""
import json
from pathlib import Path

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())
PAIR_KATS = [k for k in KATS if "unitigs" in k and "pairs" in k["expect"]]


@pytest.fixture(scope="module")
def gpu(product_lib):
    import torch

    if product_lib.mtg_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("these tests need a GPU: the matchtigs_amd hot path has no CPU fallback")
    return torch


@pytest.mark.parametrize("plan", [0, 1, 2])
@pytest.mark.parametrize("kat", PAIR_KATS, ids=[k["name"] for k in PAIR_KATS])
def test_kat_pairs_through_hip(kat, plan, gpu, oracle):
    from matchtigs_amd import api, torch_glue

    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    G = api.Bigraph.from_edges(mirror, frm, to, w)
    dev = api.DeviceGraph(G, k)
    dev.set_plan(plan)
    stream = torch_glue.current_stream_ptr()
    S = dev.classify(stream)
    on, mu, li = dev.classify_download(stream)
    if "out_nodes" in exp:
        assert on.tolist() == exp["out_nodes"]
        assert mu.tolist() == exp["multiplicity"]
    bufs = torch_glue.run_sssp(dev, 0, S)
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
    assert [(int(a), int(b), int(c)) for a, b, c in pairs] == [tuple(p) for p in exp["pairs"]]
    # the candidate lists themselves (T1) against the oracle's full-ball Dijkstra
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    _, off, keys, _ = helpers.oracle_graph(mirror, frm, to, w).candidate_lists(k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off))
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)]) if S else np.zeros(0, np.uint64)
    assert np.array_equal(got, keys)
    if "greedy_tigs" in exp:
        assert G.finish_greedytigs(pairs, k) == exp["greedy_tigs"]


T4 = [k for k in KATS if "unitigs" in k and "greedy_tigs" in k["expect"]]


@pytest.mark.parametrize("kat", T4, ids=[k["name"] for k in T4])
def test_kat_tigs_one_shot_and_clib(kat, gpu):
    from matchtigs_amd import api

    k, exp = kat["k"], kat["expect"]
    mirror, frm, to, w = helpers.unitigs_to_arrays(kat["mirror"], kat["unitigs"])
    G = api.Bigraph.from_edges(mirror, frm, to, w)
    assert api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration.new(1, k)) == exp["greedy_tigs"]
    if "greedy_cumulative_length" in exp:
        ew = G.export()["edge_weight"]
        assert helpers.cumulative_length(exp["greedy_tigs"], ew, k) == exp["greedy_cumulative_length"]
    if "euler_tigs" in exp:
        G2 = api.Bigraph.from_edges(mirror, frm, to, w)
        assert api.EulertigAlgorithm.compute_tigs(G2, api.EulertigAlgorithmConfiguration(k)) == exp["euler_tigs"]
    if "clib" in exp:   # cases whose result does not depend on the builder's node numbering (no breaking edges)
        uw = np.array([u[2] for u in kat["unitigs"]], dtype=np.uint64)
        links = helpers.links_of_bigraph(kat["mirror"], kat["unitigs"])
        n, eo, io, lo = api.clib_compute_tigs(uw, links, 5, 1, k)
        assert n == len(exp["clib"]["tigs_out_limits"])
        assert eo.tolist() == exp["clib"]["tigs_edge_out"]
        assert io.tolist() == exp["clib"]["tigs_insert_out"]
        assert lo.tolist() == exp["clib"]["tigs_out_limits"]
