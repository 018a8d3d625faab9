"""The finish on the GPU (csrc/finish_device.hip) against the host stages and the oracle.

Reference lines: dummy insertion greedytigs/mod.rs:678-689, Euleriser implementation/mod.rs:392-649 (+ :252-285), rotate + cut
greedytigs/mod.rs:726-789, eulertigs eulertigs/mod.rs:48-198. The device Euleriser must add the SAME breaking edges in the SAME
order as the reference's two-ordered-map loop (bit-exact graph after the call: ids, endpoints, weights, dummy ids), for mirror
numberings where the parallel zip covers everything (mirror = n ^ 1) and for scrambled numberings where most steps fall to the
sequential tail. In reference-order mode the tigs must equal the oracle's; in device Euler mode the invariants + T3.
Also here: the GPU generator of the G-csr input against its numpy twin.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(product_lib):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("these tests need a GPU: the device finish has no CPU fallback")
    return product_lib


def scramble(bg, seed, fraction=1.0):
    """The same bigraph with node ids permuted at random (mirror nodes no longer neighbours); fraction < 1 permutes only that
    share of the nodes among themselves (the Euleriser's parallel prefix then stops somewhere in the middle)."""
    from matchtigs_amd import synth

    rng = np.random.default_rng(seed)
    perm = np.arange(bg.n_nodes, dtype=np.uint32)  # old -> new
    chosen = np.flatnonzero(rng.random(bg.n_nodes) < fraction)
    perm[chosen] = chosen[rng.permutation(len(chosen))].astype(np.uint32)
    mirror = np.empty_like(bg.mirror)
    mirror[perm] = perm[bg.mirror]
    return synth.Bigraph(mirror, perm[bg.edge_from], perm[bg.edge_to], bg.edge_weight.copy(), bg.k)


def _cases():
    from matchtigs_amd import synth

    base = [
        ("k5-selfmirror", synth.g_csr(300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05)),
        ("k5-selfmirror-odd", synth.g_csr(301, seed=8, k=5, mean_weight=2.0, self_mirror_frac=0.05)),
        ("k9", synth.g_csr(5000, seed=11, k=9, mean_weight=3.0, mean_out_degree=1.8, self_mirror_frac=0.01)),
        ("k31", synth.g_csr(30000, seed=1, k=31, self_mirror_frac=0.0)),
        ("k31-dense", synth.g_csr(20000, seed=2, k=31, mean_out_degree=2.2, mean_weight=4.0, self_mirror_frac=0.0)),
        ("k31-sparse", synth.g_csr(20000, seed=4, k=31, mean_out_degree=0.6, self_mirror_frac=0.002)),
        ("k15-high-degree", synth.g_csr(3000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9, self_mirror_frac=0.01)),
        ("tiny", synth.g_csr(12, seed=9, k=7, mean_weight=2.0, self_mirror_frac=0.2)),
    ]
    out = []
    for name, bg in base:
        out.append((name, bg))
        out.append((name + "-scrambled", scramble(bg, 17)))
    return out


def test_euleriser_prefix_stops_in_the_middle(gpu):
    """Partially scrambled numberings (0.1 % to 30 % of the nodes renumbered at random): the first irregular step of the
    parallel Euleriser lies somewhere inside the sequence, the sequential tail takes over from there. Many seeds, graph after
    the finish and tigs equal to the host stages'."""
    from matchtigs_amd import api, synth

    n = 0
    for seed in range(24):
        base = synth.g_csr(400 + 97 * seed, seed=100 + seed, k=[5, 9, 31][seed % 3], mean_out_degree=1.1 + 0.07 * (seed % 9),
                           mean_weight=[1.5, 3.0, 8.0][seed % 3], self_mirror_frac=[0.0, 0.02, 0.1][seed % 3])
        bg = scramble(base, seed, fraction=[0.001, 0.01, 0.05, 0.3][seed % 4])
        pairs = _pairs_of(bg, bg.k)
        H, D = _graphs(bg)
        lim_h, ed_h = api.finish_greedytigs_np(H, pairs, bg.k, finish_stage=api.FinishStage.Host)
        lim_d, ed_d = api.finish_greedytigs_np(D, pairs, bg.k, finish_stage=api.FinishStage.Device)
        _same_graph(H, D)
        assert np.array_equal(lim_h, lim_d) and np.array_equal(ed_h, ed_d), seed
        n += api.last_finish_device_times()["breaking_biedges"]
    assert n > 1000


CASES = None


def case(i):
    global CASES
    if CASES is None:
        CASES = _cases()
    return CASES[i]


N_CASES = 16


def _graphs(bg):
    from matchtigs_amd import api

    return (api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight),
            api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight))


def _pairs_of(bg, k):
    from matchtigs_amd import api

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, k)
    dev.classify()
    return api.compute_pairs([dev])


def _same_graph(a, b):
    ea, eb = a.export(), b.export()
    for key in ea:
        assert np.array_equal(ea[key], eb[key]), key


@pytest.mark.parametrize("idx", range(N_CASES))
def test_device_finish_equals_host_finish_reference_order(gpu, idx):
    """Greedy finish: same graph afterwards (every dummy edge: id, endpoints, weight, dummy id) and the same tigs."""
    from matchtigs_amd import api

    name, bg = case(idx)
    k = bg.k
    pairs = _pairs_of(bg, k)
    H, D = _graphs(bg)
    lim_h, ed_h = api.finish_greedytigs_np(H, pairs, k, finish_stage=api.FinishStage.Host)
    try:
        for records in ("lean", "mid", "wide"):  # the record formats of the reference-order walk (a speed / memory choice only)
            api.set_finish_tuning(records=records)
            lim_d, ed_d = api.finish_greedytigs_np(D, pairs, k, finish_stage=api.FinishStage.Device)
            _same_graph(H, D)
            assert np.array_equal(lim_h, lim_d), (name, records)
            assert np.array_equal(ed_h, ed_d), (name, records)
            if records != "wide":
                D.reset()
    finally:
        api.set_finish_tuning()
    t = api.last_finish_device_times()
    assert t["breaking_biedges"] == (D.edge_count() - bg.n_edges) // 2 - len(pairs)
    # and the graph can be reset and finished again (the dummy edges were appended unlinked)
    D.reset()
    assert D.edge_count() == bg.n_edges
    lim_d2, ed_d2 = api.finish_greedytigs_np(D, pairs, k, finish_stage=api.FinishStage.Device)
    assert np.array_equal(lim_d, lim_d2) and np.array_equal(ed_d, ed_d2)
    # host stages on a graph the device finish left behind: adjacency gets linked on demand
    D.reset()
    lim_h2, ed_h2 = api.finish_greedytigs_np(D, pairs, k, finish_stage=api.FinishStage.Host)
    assert np.array_equal(lim_h, lim_h2) and np.array_equal(ed_h, ed_h2)


@pytest.mark.parametrize("idx", range(N_CASES))
def test_device_finish_vs_oracle(gpu, idx, oracle):
    """Whole operator with the finish forced onto the GPU == the oracle's tigs (greedy and eulertigs)."""
    from matchtigs_amd import api

    name, bg = case(idx)
    k = bg.k
    if "high-degree" in name:
        pytest.skip("the oracle's claim loop asserts de Bruijn degrees (<= 4); host == device covers this graph")
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    want, _ = og.compute_greedytigs(k)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    cfg = api.GreedytigAlgorithmConfiguration(1, k, finish_stage=api.FinishStage.Device)
    got = api.GreedytigAlgorithm.compute_tigs(G, cfg)
    assert got == want, name
    assert G.edge_count() == og.edge_count
    og2 = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    want_e = og2.compute_eulertigs(k)
    G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    got_e = api.EulertigAlgorithm.compute_tigs(G2, api.EulertigAlgorithmConfiguration(k, finish_stage=api.FinishStage.Device))
    assert got_e == want_e, name
    assert G2.edge_count() == og2.edge_count


@pytest.mark.parametrize("idx", range(N_CASES))
def test_device_finish_device_euler_invariants(gpu, idx):
    """Device Euler mode through the all-device finish: valid tigs, every unitig once, same count and cumulative length."""
    from matchtigs_amd import api
    from test_gpu_euler import _tig_invariants

    name, bg = case(idx)
    k = bg.k
    pairs = _pairs_of(bg, k)
    H, D = _graphs(bg)
    lim_h, ed_h = api.finish_greedytigs_np(H, pairs, k, finish_stage=api.FinishStage.Host)
    lim_d, ed_d = api.finish_greedytigs_np(D, pairs, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    _same_graph(H, D)  # the Euleriser does not depend on the Euler mode
    assert len(lim_h) == len(lim_d), name
    ex = D.export()
    w = ex["edge_weight"].astype(np.int64)
    assert w[ed_h].sum() == w[ed_d].sum(), name
    tigs = [ed_d[(lim_d[i - 1] if i else 0):lim_d[i]].tolist() for i in range(len(lim_d))]
    _tig_invariants(ex, tigs, k)
    orig = ed_d[ed_d < bg.n_edges]
    assert np.array_equal(np.sort(orig >> 1), np.arange(bg.n_edges // 2, dtype=orig.dtype))


def _resident(bg, k):
    """Device graph of bg with the claim replay's pairs left in HBM (mtg_replay_claims_resident)."""
    from matchtigs_amd import api, torch_glue

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, k)
    S = dev.classify()
    bufs = torch_glue.run_sssp(dev, 0, S)
    n = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
    return G, dev, n


@pytest.mark.parametrize("idx", range(N_CASES))
def test_resident_pairs_finish_equals_finish_from_host_pairs(gpu, idx):
    """The pairs never leave the GPU between claim replay and finish (what mtg_compute_tigs_cfg does): same pair list as the
    downloading replay, same graph afterwards and same tigs as the finish that is handed host pairs -- both Euler modes."""
    from matchtigs_amd import api

    name, bg = case(idx)
    k = bg.k
    pairs = _pairs_of(bg, k)
    G, dev, n = _resident(bg, k)
    assert n == len(pairs)
    got = dev.download_resident_pairs()
    assert np.array_equal(got, pairs), name
    for mode in (api.EulerMode.HostReferenceOrder, api.EulerMode.Device):
        A, B = _graphs(bg)
        lim_a, ed_a = api.finish_greedytigs_np(A, pairs, k, euler_mode=mode, finish_stage=api.FinishStage.Device)
        lim_b, ed_b = api.finish_greedytigs_resident_np(B, dev, k, euler_mode=mode, finish_stage=api.FinishStage.Device)
        _same_graph(A, B)
        assert np.array_equal(lim_a, lim_b) and np.array_equal(ed_a, ed_b), (name, mode)
    # finish_stage HOST: the resident pairs take the way through the host
    A, B = _graphs(bg)
    lim_a, ed_a = api.finish_greedytigs_np(A, pairs, k, finish_stage=api.FinishStage.Host)
    lim_b, ed_b = api.finish_greedytigs_resident_np(B, dev, k, finish_stage=api.FinishStage.Host)
    _same_graph(A, B)
    assert np.array_equal(lim_a, lim_b) and np.array_equal(ed_a, ed_b), name


@pytest.mark.parametrize("log2_edges, euler", [(14, "host"), (14, "device"), (23, "device"), (23, "host")])
def test_tigs_stay_in_hbm_until_asked_for(gpu, log2_edges, euler):
    """The tigs of a finish on the GPU stay in HBM: the handle answers count() / total_edges() without a copy, arrays()
    brings them to the host once (plain copy at 2^14, through the pinned ring at 2^23), equal to what the array-returning call of the
    same finish delivers; the handle can be dropped without ever being downloaded; flattening (clib.rs:393-407) works on either."""
    from matchtigs_amd import api, synth, torch_glue

    k = 31
    L = gpu
    mode = api.EulerMode.Device if euler == "device" else api.EulerMode.HostReferenceOrder

    G = synth.g_csr_device(int((1 << log2_edges) / 3), seed=70 + log2_edges, k=k)
    E0 = G.edge_count()
    dev = api.DeviceGraph(G, k)
    S = dev.classify()
    bufs = torch_glue.run_sssp(dev, 0, S)

    def run():
        dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
        return api.finish_greedytigs_resident(G, dev, k, mode, 0, api.FinishStage.Device)

    t = run()
    n, tot = t.count(), t.total_edges()
    assert n > 0 and tot >= n
    lim, ed = t.arrays()
    assert len(lim) == n and len(ed) == tot and int(lim[-1]) == tot and np.all(np.diff(lim.astype(np.int64)) > 0)
    lim2, ed2 = t.arrays()  # (a second request returns the same host arrays)
    assert np.array_equal(lim, lim2) and np.array_equal(ed, ed2)
    e1, i1, l1 = np.full(2 * E0, -7, np.int64), np.full(2 * E0, 7, np.uint64), np.full(E0, 7, np.uint64)
    assert L.mtg_flatten_clib(G.handle, t._wp, e1.ctypes.data, i1.ctypes.data, l1.ctypes.data) == n
    assert np.array_equal(l1[:n], lim)
    lim, ed = lim.copy(), ed.copy()
    del t
    G.reset()
    t2 = run()            # the same finish again: same tigs; flattened straight from the handle, never asked for as walks
    assert t2.count() == n and t2.total_edges() == tot
    e2, i2, l2 = np.full(2 * E0, -7, np.int64), np.full(2 * E0, 7, np.uint64), np.full(E0, 7, np.uint64)
    assert L.mtg_flatten_clib(G.handle, t2._wp, e2.ctypes.data, i2.ctypes.data, l2.ctypes.data) == n
    assert np.array_equal(e1[:tot], e2[:tot]) and np.array_equal(i1[:tot], i2[:tot]) and np.array_equal(l1[:n], l2[:n])
    del t2
    G.reset()
    t3 = run()            # ... and dropped without a download
    assert t3.count() == n
    del t3


def test_resident_finish_with_the_dummy_edges_downloaded_beside_the_gpu_stages(gpu):
    """A graph with more than 2^20 dummy darts: in device Euler mode the dummy edges reach the host graph on a side stream from a
    thread of their own, the tig limits travel as 32-bit words. Graph and tigs as after the finish from host pairs, twice."""
    from matchtigs_amd import api, synth, torch_glue

    k = 31
    G = synth.g_csr_device(1 << 21, seed=5, k=k)
    ex = G.export()
    bg = synth.Bigraph(ex["mirror"], ex["edge_from"], ex["edge_to"], ex["edge_weight"], k)
    dev = api.DeviceGraph(G, k)
    S = dev.classify()
    bufs = torch_glue.run_sssp(dev, 0, S)
    n = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
    pairs = dev.download_resident_pairs()
    assert n == len(pairs) > 0
    H = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    lim_h, ed_h = api.finish_greedytigs_np(H, pairs, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    assert H.edge_count() - bg.n_edges >= 1 << 20
    eh = H.export()
    for _ in range(2):
        lim_d, ed_d = api.finish_greedytigs_resident_np(G, dev, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
        eg = G.export()
        for key in eh:
            assert np.array_equal(eh[key], eg[key]), key
        assert np.array_equal(lim_h, lim_d) and np.array_equal(ed_h, ed_d)
        G.reset()
    # ... and the host Euleriser agrees on the graph
    H2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    api.finish_greedytigs_np(H2, pairs, k, finish_stage=api.FinishStage.Host)
    e2 = H2.export()
    for key in eh:
        assert np.array_equal(eh[key], e2[key]), key


@pytest.mark.parametrize("log2_edges, delay_us, records, max_degree", [(22, 0, None, 4), (22, 300, None, 4), (20, 3000, None, 4), (17, 20000, None, 4),
                                                                      (22, 300, "mid", 4), (17, 20000, "mid", 4),
                                                                      # no node with more than three out-edges: no spill arrays, so nothing but the
                                                                      # event behind lean_build_kernel orders the side stream's download of the
                                                                      # 32-byte records after the kernel that writes them
                                                                      (18, 2000, None, 3), (22, 0, None, 3)])
def test_reference_order_walk_starts_while_its_records_arrive(gpu, oracle, log2_edges, delay_us, records, max_degree):
    """The reference-order host walk starts while its 256-byte records still cross PCIe: they arrive in node order (first call on a
    graph: through the pinned ring on a thread of its own; later calls: plain copies into the page-locked arena, followed by a watcher
    thread), a counter says how far they have come, and a step that needs a record beyond that mark takes the node's 32-byte record
    instead. mtg_set_finish_tuning's record_delay_us slows the arrival so that small graphs take that path for most of their steps. Same tigs as the
    host stages' finish (no GPU records at all), and -- on the smallest graph -- as the oracle. `records` = "mid": the 128-byte
    records of the graphs whose 256-byte ones would not fit the host (built and brought down slice by slice beside the walk)."""
    from matchtigs_amd import api, synth

    k = 31
    G = synth.g_csr_device(int((1 << log2_edges) / 3), seed=5 + log2_edges, k=k, max_degree=max_degree)
    dev = api.DeviceGraph(G, k)
    dev.classify()
    pairs = api.compute_pairs([dev])
    del dev
    ref_lim, ref_ed = api.finish_greedytigs_np(G, pairs, k, euler_mode=api.EulerMode.HostReferenceOrder, finish_stage=api.FinishStage.Host)
    ref_lim, ref_ed = ref_lim.copy(), ref_ed.copy()
    G.reset()
    api.set_finish_tuning(records=records, record_delay_us=delay_us)
    results = [(ref_lim, ref_ed)]
    try:
        for _ in range(3):  # call 1: records through the pinned ring; calls 2, 3: the arena is page-locked
            lim, ed = api.finish_greedytigs_np(G, pairs, k, euler_mode=api.EulerMode.HostReferenceOrder, finish_stage=api.FinishStage.Device)
            results.append((lim.copy(), ed.copy()))
            G.reset()
    finally:
        api.set_finish_tuning()
    for lim, ed in results[1:]:
        assert np.array_equal(results[0][0], lim) and np.array_equal(results[0][1], ed)
    if log2_edges <= 17:
        ex = G.export()
        og = oracle.OracleGraph.from_arrays(ex["mirror"], ex["edge_from"], ex["edge_to"], ex["edge_weight"])
        want, _ = og.compute_greedytigs(k)
        lim, ed = results[-1]
        got = [ed[int(a):int(b)].tolist() for a, b in zip(np.concatenate([[0], lim[:-1]]), lim)]
        assert got == want


@pytest.mark.parametrize("log2_edges, algorithm, euler", [(12, 5, "host"), (12, 3, "device"), (18, 5, "host"), (18, 5, "device"), (18, 3, "host"),
                                                         (22, 5, "device"), (23, 3, "device"), (25, 5, "device"), (25, 3, "device")])
def test_compute_tigs_clib_equals_compute_plus_flatten(gpu, oracle, log2_edges, algorithm, euler):
    """mtg_compute_tigs_clib (what matchtigs_compute_tigs runs, clib.rs:280-410): with a finish on the GPU the tigs go from the
    download ring straight into the caller's clib.rs arrays (flattened by the host threads that empty the ring; the cases from 2^22 on
    take the ring -- 2^22 with a single partial slice --, the others the plain copy). Same arrays as mtg_compute_tigs_cfg + mtg_flatten_clib (clib.rs:393-407), and -- in the
    reference's walk order, on the sizes the oracle finishes -- as the oracle's tigs flattened by the same rule."""
    import ctypes as C

    from matchtigs_amd import api, synth

    k = 31
    L = gpu
    G = synth.g_csr_device(int((1 << log2_edges) / 3), seed=40 + log2_edges, k=k)
    E0 = G.edge_count()
    mode = api.EulerMode.Device if euler == "device" else api.EulerMode.HostReferenceOrder
    cfg = api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode).to_c()
    w = L.mtg_compute_tigs_cfg(G.handle, algorithm, C.byref(cfg))
    e1, i1, l1 = np.full(2 * E0, -7, np.int64), np.full(2 * E0, 7, np.uint64), np.full(E0, 7, np.uint64)
    n1 = L.mtg_flatten_clib(G.handle, w, e1.ctypes.data, i1.ctypes.data, l1.ctypes.data)
    L.mtg_walks_free(w)
    ex = G.export() if log2_edges <= 18 and euler == "host" else None
    G.reset()
    e2, i2, l2 = np.full(2 * E0, -7, np.int64), np.full(2 * E0, 7, np.uint64), np.full(E0, 7, np.uint64)
    n2 = L.mtg_compute_tigs_clib(G.handle, algorithm, C.byref(cfg), e2.ctypes.data, i2.ctypes.data, l2.ctypes.data)
    assert n1 == n2 and n1 > 0
    m = int(l1[n1 - 1])
    assert np.array_equal(l1[:n1], l2[:n2])
    assert np.array_equal(e1[:m], e2[:m])
    assert np.array_equal(i1[:m], i2[:m])
    if ex is not None:
        n_orig = E0
        og = oracle.OracleGraph.from_arrays(ex["mirror"], ex["edge_from"][:n_orig], ex["edge_to"][:n_orig], ex["edge_weight"][:n_orig])
        n_o, eo, io, lo = og.clib_compute_tigs(algorithm, k)  # the oracle's own flattening (clib.rs:393-407)
        assert n_o == n2 and np.array_equal(lo[:n_o], l2[:n2])
        assert np.array_equal(eo[:m], e2[:m]) and np.array_equal(io[:m], i2[:m])


def test_kept_device_memory_is_bounded_and_can_be_released(gpu):
    """The finish keeps the work arrays of its LAST call only (a smaller call after a larger one frees what it did not touch), and
    mtg_release_device_memory / mtg_graph_release_device_cache return everything; results do not change."""
    from matchtigs_amd import api, synth

    k = 31
    small = synth.g_csr(3000, seed=22, k=k)
    api.release_device_memory(0)
    assert api.device_memory_held(0) == 0
    GB = synth.g_csr_device(1 << 20, seed=21, k=k)  # work arrays of tens of MB (blocks are kept in units of 1 MB)
    dev = api.DeviceGraph(GB, k)
    dev.classify()
    pb = api.compute_pairs([dev])
    del dev
    want = api.finish_greedytigs_np(GB, pb, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    held_big = api.device_memory_held(0)
    assert held_big > 0
    GS = api.Bigraph.from_edges(small.mirror, small.edge_from, small.edge_to, small.edge_weight)
    ps = _pairs_of(small, k)
    api.finish_greedytigs_np(GS, ps, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    assert api.device_memory_held(0) < held_big  # the large call's arrays are gone
    api.release_device_memory(0)
    assert api.device_memory_held(0) == 0
    GB.reset()
    GB.release_device_cache()
    again = api.finish_greedytigs_np(GB, pb, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    assert np.array_equal(want[0], again[0]) and np.array_equal(want[1], again[1])


@pytest.mark.parametrize("args", [
    dict(n_binodes=12, seed=9, k=7, mean_weight=2.0, self_mirror_frac=0.2),
    dict(n_binodes=300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05),
    dict(n_binodes=30000, seed=1, k=31),
    dict(n_binodes=20000, seed=2, k=31, mean_out_degree=2.2, mean_weight=4.0, self_mirror_frac=0.0),
    dict(n_binodes=3000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9, self_mirror_frac=0.0),
    dict(n_binodes=1 << 20, seed=3, k=31),
    dict(n_binodes=200000, seed=7, k=63, mean_weight=20.0),
])
def test_device_generator_equals_numpy_generator(gpu, args):
    from matchtigs_amd import synth

    bg = synth.g_csr(**args)
    G = synth.g_csr_device(**args)
    ex = G.export()
    assert np.array_equal(ex["mirror"], bg.mirror)
    assert np.array_equal(ex["edge_from"], bg.edge_from)
    assert np.array_equal(ex["edge_to"], bg.edge_to)
    assert np.array_equal(ex["edge_weight"], bg.edge_weight)
    assert not ex["edge_dummy_id"].any()
    assert np.array_equal(ex["edge_unitig"], np.arange(bg.n_edges, dtype=np.uint64) // 2)
    assert np.array_equal(ex["edge_forwards"], (np.arange(bg.n_edges) % 2 == 0).astype(np.uint8))
