"""SURVEY 8e inside the library (mtg_compute_pairs / mtg_config.device_ids): sources block-partitioned by work over several
resident copies of the graph, peer-copy gather on the first, claim replay there. A 1-GPU box can only check the FUNCTION
(every copy lives on GPU 0); the pair list and the tigs must be identical to the single-device path and to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("out", "in", "dist"))


@pytest.mark.parametrize("n_dev", [2, 3])
def test_compute_pairs_over_several_device_copies(n_dev, oracle, product_lib):
    from matchtigs_amd import api, synth

    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    bg = synth.g_csr(40000, seed=5, k=31, mean_out_degree=1.7)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    devs = [api.DeviceGraph(G, bg.k, 0) for _ in range(n_dev)]
    S = [d.classify() for d in devs]
    assert len(set(S)) == 1
    cuts = api.partition_sources(devs[0], n_dev)
    assert cuts[0] == 0 and cuts[-1] == S[0] and all(a <= b for a, b in zip(cuts, cuts[1:]))
    # blocks of near-equal estimated work (1 + out-degree of the source)
    on, _, _ = devs[0].classify_download()
    work = 1 + np.bincount(bg.edge_from, minlength=bg.n_nodes)[on]
    per_block = [int(work[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
    assert max(per_block) - min(per_block) <= 8
    got = api.compute_pairs(devs)
    one = api.compute_pairs(devs[:1])
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).greedy_pairs_np(bg.k)
    assert _pairs_equal(got, one) and _pairs_equal(got, want)


def test_compute_tigs_cfg_with_device_ids(oracle, product_lib):
    """mtg_compute_tigs_cfg with n_devices = 2 (both ids 0 on this box): tigs equal the oracle's."""
    from matchtigs_amd import api, synth

    bg = synth.g_csr(20000, seed=9, k=31)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration(1, bg.k, device_ids=(0, 0)))
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).compute_greedytigs(bg.k)
    assert tigs == want
