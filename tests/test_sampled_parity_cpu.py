"""tests/sampled_parity.py checked against itself on the CPU: on a graph the oracle holds whole, the ball subgraph of a sample of
sources + the full graph's classification must give the oracle the same lists and the same pair prefix as the whole graph does
(torch on the CPU as the calculator; the "GPU side" of the comparison is the whole-graph oracle here)."""
import types

import numpy as np
import pytest


@pytest.mark.parametrize("seed,k,self_mirror_frac", [(1, 31, 0.0), (7, 31, 0.02), (3, 9, 0.05)])
def test_ball_subgraph_reproduces_the_whole_graph_oracle(oracle, product_lib, seed, k, self_mirror_frac):
    import torch

    import sampled_parity
    from matchtigs_amd import api, synth

    bg = synth.g_csr(20000, seed=seed, k=k, mean_weight=8.0 if k == 31 else 2.5, self_mirror_frac=self_mirror_frac)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    on, live, mult, _, _ = og.classify()
    o_on, off, keys, _ = og.candidate_lists(k)
    assert np.array_equal(on, o_on)
    pairs, _ = og.greedy_pairs_np(k)
    bufs = types.SimpleNamespace(n=len(on), start=torch.from_numpy(off[:-1].astype(np.int64)), count=torch.from_numpy(np.diff(off).astype(np.int32)),
                                 pool=torch.from_numpy(keys.view(np.int64) if len(keys) else np.zeros(1, np.int64)))
    res = sampled_parity.check_sampled_lists_and_prefix(torch, oracle, G, bufs, on, mult.astype(np.int32), live, pairs, k, n_prefix=700, n_tail=100,
                                                        n_random=600, device="cpu")
    assert res["sampled"] == 1400 and res["prefix_pairs"] > 100 and res["subgraph_edges"] < bg.n_edges
    # ... and a wrong list is noticed: one key of a sampled source changed
    if len(keys):
        bad = keys.copy()
        first = int(np.nonzero(np.diff(off))[0][0])  # (a source of the prefix region or not: the first one with candidates)
        bad[off[first]] += np.uint64(1 << 32)
        bufs.pool = torch.from_numpy(bad.view(np.int64))
        if first < 700:
            with pytest.raises(AssertionError):
                sampled_parity.check_sampled_lists_and_prefix(torch, oracle, G, bufs, on, mult.astype(np.int32), live, pairs, k, n_prefix=700, n_tail=100,
                                                              n_random=600, device="cpu", log=lambda *_: None)
