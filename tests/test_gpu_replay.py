"""The claim loop on the GPU (mtg_replay_claims_device: deterministic reservations by source index) must reproduce the
sequential 1-thread claim order exactly: compared with the oracle's truncated-Dijkstra claim loop (T2) and with the
product's host replay on the same candidate lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("out", "in", "dist"))


def _run(bg, plan=None, tuning=None):
    import torch  # noqa: F401
    from matchtigs_amd import api, torch_glue

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, bg.k)
    if plan is not None:
        dev.set_plan(plan)
    if tuning:
        dev.set_replay_tuning(**tuning)
    S = dev.classify(torch_glue.current_stream_ptr())
    bufs = torch_glue.run_sssp(dev, 0, S)
    gpu_pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(),
                                         torch_glue.current_stream_ptr())
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    on, mu, li = dev.classify_download()
    host_pairs = G.replay_claims(on, mu, li, start, count, pool)
    return G, dev, gpu_pairs, host_pairs


@pytest.mark.parametrize("case", [
    dict(n_binodes=300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05),
    dict(n_binodes=5000, seed=11, k=9, mean_weight=3.0, mean_out_degree=1.8, self_mirror_frac=0.02),
    dict(n_binodes=30000, seed=1, k=31),
    dict(n_binodes=20000, seed=2, k=31, mean_out_degree=2.2, mean_weight=4.0),
    dict(n_binodes=20000, seed=6, k=31, mean_out_degree=2.6, mean_weight=2.0, self_mirror_frac=0.05),  # dense conflicts
    dict(n_binodes=8000, seed=5, k=63, mean_weight=10.0),
])
def test_gpu_replay_equals_sequential_claim_order(case, oracle, product_lib):
    from matchtigs_amd import synth

    bg = synth.g_csr(**case)
    G, dev, gpu_pairs, host_pairs = _run(bg)
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).greedy_pairs_np(bg.k)
    assert _pairs_equal(host_pairs, want)
    assert _pairs_equal(gpu_pairs, want)
    assert dev.last_replay_rounds() >= 1 or len(want) == 0


def test_gpu_replay_on_real_dbg_with_self_mirrors_and_mirror_candidates(oracle, product_lib):
    """Real tiny dBGs exercise the mirror-of-self candidate rule (:352-358) and binode sharing between a source and the
    mirror of one of its candidates."""
    from matchtigs_amd import api, synth, torch_glue

    for seed, k in ((2, 11), (3, 15), (9, 21)):
        ug = synth.g_seq(6000, seed=seed, k=k, haplotypes=4, sub_rate=0.03)
        G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
        dev = api.DeviceGraph(G, k)
        S = dev.classify(torch_glue.current_stream_ptr())
        bufs = torch_glue.run_sssp(dev, 0, S)
        got = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), torch_glue.current_stream_ptr())
        want, _ = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links).greedy_pairs_np(k)
        assert _pairs_equal(got, want), (seed, k)


def test_gpu_replay_full_bench_size(oracle, product_lib):
    from matchtigs_amd import synth

    bg = synth.g_csr(int((1 << 24) / 1.5 / 2), seed=1, k=31)
    G, dev, gpu_pairs, host_pairs = _run(bg)
    assert _pairs_equal(gpu_pairs, host_pairs)
    assert len(gpu_pairs) == 570447  # the oracle's count for this workload (test_full_bench_size_properties checks the list)
    print("reservation rounds:", dev.last_replay_rounds())


@pytest.mark.parametrize("case", [
    # unit weights, out-degree ~2.6, k = 31: balls of thousands of nodes -- candidate lists of hundreds to thousands of entries
    # (beyond one pass of a wave, beyond the 10-bit indices of a touch record, beyond the 12-bit indices of a packed claims word)
    dict(n_binodes=3000, seed=9, k=31, mean_out_degree=2.6, mean_weight=1.0, self_mirror_frac=0.01),
    dict(n_binodes=9000, seed=4, k=31, mean_out_degree=2.4, mean_weight=1.0, self_mirror_frac=0.0),
    # degrees up to 9: multiplicities above 4, claims go to the spill array
    dict(n_binodes=3000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9, self_mirror_frac=0.01),
])
def test_gpu_replay_long_lists_and_large_demands(case, product_lib):
    """The paths of the claim replay that the bench graph hardly touches, against the host claim loop on the same lists: long
    candidate lists (checked, reserved and admitted by the whole wave) and demands above four (spilled claim indices)."""
    from matchtigs_amd import synth, torch_glue  # noqa: F401

    bg = synth.g_csr(**case)
    G, dev, gpu_pairs, host_pairs = _run(bg)
    assert len(host_pairs) > 100
    assert _pairs_equal(gpu_pairs, host_pairs)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_replay_scrambled_numbering(seed, oracle, product_lib):
    """Mirror nodes NOT numbered next to each other (node ids permuted at random): the paired 16-byte state accesses fall back to
    single words, the source's and the candidates' mirrors are read on their own."""
    from matchtigs_amd import synth
    from test_gpu_finish import scramble

    bg = scramble(synth.g_csr(20000, seed=seed, k=31, mean_out_degree=2.0, self_mirror_frac=0.02), 40 + seed)
    G, dev, gpu_pairs, host_pairs = _run(bg)
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).greedy_pairs_np(bg.k)
    assert _pairs_equal(host_pairs, want)
    assert _pairs_equal(gpu_pairs, want)


@pytest.mark.parametrize("block", [256, 1024])
def test_gpu_replay_both_workgroup_sizes(block, oracle, product_lib):
    """The rounds kernel is instantiated for workgroups of 1024 (rounds that fill the device) and of 256 (small inputs); the
    choice is by input size, mtg_set_replay_tuning forces it: both on the same graphs, against the host loop."""
    from matchtigs_amd import synth

    for case in (dict(n_binodes=20000, seed=6, k=31, mean_out_degree=2.6, mean_weight=2.0, self_mirror_frac=0.05),
                 dict(n_binodes=3000, seed=9, k=31, mean_out_degree=2.6, mean_weight=1.0, self_mirror_frac=0.01)):
        bg = synth.g_csr(**case)
        G, dev, gpu_pairs, host_pairs = _run(bg, tuning=dict(block=block))
        assert _pairs_equal(gpu_pairs, host_pairs), (block, case)


def test_gpu_replay_grid_barrier_forms_and_launch_geometries(oracle, product_lib):
    """The grid barrier's per-XCD stage (one release per XCD: leans on gfx942 / gfx950 hardware) against the plain form (every
    workgroup releases: the memory model alone), and other window counts / role splits / grids -- on a graph of 2^24 nominal edges,
    where the rounds fill the device: the same pair list as the host loop under every setting."""
    from matchtigs_amd import api, synth, torch_glue

    k = 31
    G = synth.g_csr_device(int((1 << 24) / 3), seed=4, k=k)
    dev = api.DeviceGraph(G, k)
    S = dev.classify(torch_glue.current_stream_ptr())
    bufs = torch_glue.run_sssp(dev, 0, S)
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    on, mu, li = dev.classify_download()
    host_pairs = G.replay_claims(on, mu, li, start, count, pool)
    assert len(host_pairs) > 100000
    # (windows: bits 0-15 their number, bits 16-23 the sixteenth of them from which they grow, bits 24-31 by which factor; the engine's
    # own choice -- dict() -- is 36 windows, the second half twice as large)
    for tuning in (dict(), dict(plain_barrier=True), dict(windows=7, role_mod=3), dict(windows=96, plain_barrier=True, role_mod=1),
                   dict(block=256, grid=100), dict(block=256, plain_barrier=True), dict(windows=48), dict(windows=20 | (4 << 16) | (5 << 24)),
                   dict(windows=9 | (15 << 16) | (3 << 24), role_mod=2),
                   # bit 32: the windows adapt to the measured round duration and retry rate, on top of the default list / of 24 windows
                   dict(windows=1 << 32), dict(windows=(1 << 32) | 24, block=1024), dict(windows=(1 << 32) | 12, plain_barrier=True, block=256)):
        dev.set_replay_tuning(**tuning)
        got = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), torch_glue.current_stream_ptr())
        assert _pairs_equal(got, host_pairs), tuning
