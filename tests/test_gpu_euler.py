"""Euler bicycles on the GPU (euler_device.hip).

A parallel Euler decomposition cannot reproduce the sequential tie-breaks of the reference's Hierholzer walk, so this
mode is judged on the guarantees the reference's callers rely on (greedytigs/mod.rs:708-789), not on bytes:
closed walks, consecutive edges adjacent, every biedge exactly once in one orientation, one walk per connected
component (= the number the reference-order walk finds), and -- after the unchanged cutter -- the same number of tigs
and the same cumulative length as the reference-order path (SURVEY 8a invariance note).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(product_lib):
    import torch

    if product_lib.mtg_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("these tests need a GPU: the device Euler mode has no CPU fallback")
    return torch


def _cases():
    from matchtigs_amd import synth

    return [
        ("k5-selfmirror", synth.g_csr(300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05)),
        ("k9", synth.g_csr(5000, seed=11, k=9, mean_weight=3.0, mean_out_degree=1.8, self_mirror_frac=0.01)),
        ("k31", synth.g_csr(30000, seed=1, k=31, self_mirror_frac=0.0)),
        ("k31-dense", synth.g_csr(20000, seed=2, k=31, mean_out_degree=2.2, mean_weight=4.0, self_mirror_frac=0.0)),
        ("k31-sparse-many-components", synth.g_csr(20000, seed=4, k=31, mean_out_degree=0.6, self_mirror_frac=0.0)),
        ("k63", synth.g_csr(8000, seed=5, k=63, mean_weight=10.0)),
        ("k15-high-degree", synth.g_csr(3000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9, self_mirror_frac=0.0)),
    ]


def check_bicycles(ex, limits, edges):
    """Every biedge once, walks closed, consecutive edges adjacent."""
    E = len(ex["edge_from"])
    assert len(edges) == E // 2
    assert np.array_equal(np.sort(edges >> 1), np.arange(E // 2, dtype=edges.dtype))
    frm, to = ex["edge_from"][edges], ex["edge_to"][edges]
    begin = np.concatenate([[0], limits[:-1]]).astype(np.int64)
    end = limits.astype(np.int64)
    assert (end > begin).all()
    nxt = np.arange(1, len(edges) + 1, dtype=np.int64)
    nxt[end - 1] = begin                       # the successor of a walk's last edge is its first
    assert np.array_equal(to, frm[nxt])
    # walks start at the smallest edge id of their component, ascending
    first = edges[begin]
    assert (np.diff(first.astype(np.int64)) > 0).all()
    assert (first % 2 == 0).all()


@pytest.mark.parametrize("idx", range(7))
def test_device_bicycles_on_eulerised_graph(gpu, idx):
    name, bg = _cases()[idx]
    from matchtigs_amd import api

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, bg.k)
    ex = G.export()
    limits, edges = G.euler_cycles_device_np()
    check_bicycles(ex, limits, edges)
    host = G.euler_cycles()
    assert len(limits) == len(host), name      # one closed walk per connected component
    assert sorted(min(e & ~1 for e in c) for c in host) == edges[np.concatenate([[0], limits[:-1]]).astype(np.int64)].tolist()
    assert sorted(len(c) for c in host) == sorted(np.diff(np.concatenate([[0], limits]).astype(np.int64)).tolist())


def _tig_invariants(ex, tigs, k):
    dummy = ex["edge_dummy_id"] != 0
    seen = np.zeros(len(dummy) // 2, np.int64)
    for t in tigs:
        a = np.asarray(t, dtype=np.int64)
        assert not dummy[a[0]] and not dummy[a[-1]]                     # greedytigs/mod.rs:794-798
        assert np.array_equal(ex["edge_to"][a[:-1]], ex["edge_from"][a[1:]])
        assert not (dummy[a[:-1]] & dummy[a[1:]]).any()                 # implementation/mod.rs:319-390
        assert (ex["edge_weight"][a[dummy[a]]] < k).all()               # no breaking edge survives the cut
        np.add.at(seen, a[~dummy[a]] >> 1, 1)
    n_orig = int((~dummy).sum()) // 2
    assert (seen[:n_orig] == 1).all()                                   # every unitig exactly once


@pytest.mark.parametrize("idx", range(7))
def test_greedytigs_with_device_euler_mode(gpu, idx):
    name, bg = _cases()[idx]
    from matchtigs_amd import api

    k = bg.k
    cfg = api.GreedytigAlgorithmConfiguration.new(1, k)
    G0 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    ref_tigs = api.GreedytigAlgorithm.compute_tigs(G0, cfg)            # host walk, reference order
    w0 = G0.export()["edge_weight"]
    dcfg = api.GreedytigAlgorithmConfiguration(1, k, euler_mode=api.EulerMode.Device)
    G1 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    tigs = api.GreedytigAlgorithm.compute_tigs(G1, dcfg)
    G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    etigs = api.EulertigAlgorithm.compute_tigs(G2, api.EulertigAlgorithmConfiguration(k, euler_mode=api.EulerMode.Device))
    ex = G1.export()
    _tig_invariants(ex, tigs, k)
    _tig_invariants(G2.export(), etigs, k)
    cum = lambda ts, w: sum(int(w[np.asarray(t)].sum()) + k - 1 for t in ts)
    has_self_mirror = bool((bg.mirror == np.arange(len(bg.mirror))).any())
    if not has_self_mirror:
        # T3 is invariant under the Euler order when no two breaking edges can become consecutive (SURVEY 8a note)
        assert len(tigs) == len(ref_tigs), name
        assert cum(tigs, ex["edge_weight"]) == cum(ref_tigs, w0), name
    else:
        assert abs(len(tigs) - len(ref_tigs)) <= max(2, len(ref_tigs) // 50), name


def test_device_euler_real_dbg_kmer_set(gpu):
    """Spelled tigs of a real (tiny) de Bruijn graph still contain exactly the input k-mer set."""
    from matchtigs_amd import api, synth

    ug = synth.g_seq(3000, seed=7, k=15)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration(1, ug.k, euler_mode=api.EulerMode.Device))
    fasta = api.write_walks_fasta(G, tigs, ug.unitigs, ug.k).decode()
    seqs = [l for l in fasta.splitlines() if l and not l.startswith(">")]
    assert synth.kmer_set_of_tigs(seqs, ug.k) == ug.kmers


def test_device_euler_full_bench_size(gpu):
    """BASELINE configs[2] shape (Eulertigs only) at |E| = 2^24: the device decomposition is a valid set of bicycles, and
    after the unchanged cutter it gives the same number of tigs and the same cumulative length as the reference-order
    host walk (T3; the graph has no self-mirror nodes, so the invariance note of SURVEY 8a applies exactly)."""
    from matchtigs_amd import api, synth

    k = 31
    bg = synth.g_csr(5_592_405, seed=1, k=k, self_mirror_frac=0.0)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, k)
    limits, edges = G.euler_cycles_device_np()
    ex = G.export()
    check_bicycles(ex, limits, edges)
    G.reset()
    lim_h, ed_h = api.EulertigAlgorithm.compute_tigs_np(G, api.EulertigAlgorithmConfiguration(k))
    w_h = G.export()["edge_weight"]
    G.reset()
    lim_d, ed_d = api.EulertigAlgorithm.compute_tigs_np(G, api.EulertigAlgorithmConfiguration(k, euler_mode=api.EulerMode.Device))
    w_d = G.export()["edge_weight"]
    assert len(lim_d) == len(lim_h)                                                  # T3: #tigs
    assert int(w_d[ed_d].sum()) + (k - 1) * len(lim_d) == int(w_h[ed_h].sum()) + (k - 1) * len(lim_h)  # T3: cumulative length
    n_orig = bg.n_edges
    for lim, ed in ((lim_h, ed_h), (lim_d, ed_d)):
        orig = ed[ed < n_orig]
        assert len(orig) == n_orig // 2 and len(np.unique(orig >> 1)) == n_orig // 2   # every unitig exactly once


@pytest.mark.parametrize("algorithm", ["greedy", "euler"])
def test_device_order_with_self_mirror_nodes_at_bench_size(gpu, algorithm):
    """Device order at |E| = 2^24 on a graph WITH self-mirror nodes (1 % of the binodes), the tigs cut straight from the pairing
    (cut_first_device.hip) and through the closed walks: valid tig sets (tests/gpu_props.py: every unitig once, consecutive edges
    adjacent, ends original, no breaking edge inside, Eulerian graph) whose cumulative length minus (k - 1) per tig -- the unitigs'
    k-mers plus the matched dummies kept -- equals the reference-order result's exactly. The tig COUNT may differ where two breaking
    edges meet at a self-mirror node (the empty stretch between them is no tig: greedytigs/mod.rs:772-774, SURVEY 8a's exception),
    by at most one per self-mirror node; without such nodes it is equal (test_device_euler_full_bench_size)."""
    import gpu_props
    from matchtigs_amd import api, synth

    torch = gpu
    k = 31
    G = synth.g_csr_device(5_592_405, seed=3, k=k, self_mirror_frac=0.01)
    mirror = G.export_mirror()
    n_sm = int((mirror == np.arange(len(mirror), dtype=mirror.dtype)).sum())
    assert n_sm > 50000

    def run(mode, no_cut_first=False):
        api.set_finish_tuning(no_cut_first=no_cut_first)
        try:
            if algorithm == "greedy":
                lim, ed = api.GreedytigAlgorithm.compute_tigs_np(G, api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode))
            else:
                lim, ed = api.EulertigAlgorithm.compute_tigs_np(G, api.EulertigAlgorithmConfiguration(k, euler_mode=mode))
        finally:
            api.set_finish_tuning()
        cum, dummy_kmers = gpu_props.check_tigs(torch, G, lim, ed, k)
        G.reset()
        return len(lim), cum - (k - 1) * len(lim), dummy_kmers, ed

    ref = run(api.EulerMode.HostReferenceOrder)
    cut = run(api.EulerMode.Device)
    walks = run(api.EulerMode.Device, no_cut_first=True)
    again = run(api.EulerMode.Device)
    assert np.array_equal(cut[3], again[3])  # reproducible
    for got in (cut, walks):
        assert got[1] == ref[1] and got[2] == ref[2]
        assert abs(got[0] - ref[0]) <= n_sm
    print(f"{algorithm}: tigs reference order {ref[0]}, cut first {cut[0]}, closed walks {walks[0]}; {n_sm} self-mirror nodes")


@pytest.mark.parametrize("ring", [30_000, 70_000])
def test_device_order_with_a_large_balanced_component(gpu, ring):
    """Trails WITHOUT a breaking dart at size (cut_first_device.hip): a G-csr graph plus a separate ring of `ring` unitigs -- a balanced
    component: no breaking edge, two mirror trails of `ring` darts that no walker passes. 30 000: found through their unmarked local
    minima, listed, and turned into one cyclic tig by the host's splicing step; 70 000: longer than one thread walks (65 536), so the
    step goes through the closed walks of euler_device.hip instead. Either way: valid tigs, the reference-order tig count, every unitig
    once; the ring is one tig of `ring` edges."""
    import gpu_props
    from matchtigs_amd import api, synth

    torch = gpu
    k = 31
    bg = synth.g_csr(2_400_000, seed=8, k=k, self_mirror_frac=0.0)   # ~7.2 M original darts: a 64th of the darts is above 65 536
    V0 = bg.n_nodes
    a = V0 + 2 * np.arange(ring, dtype=np.uint32)            # ring node i and its mirror a + 1
    mirror = np.concatenate([bg.mirror, np.stack([a + 1, a], axis=1).reshape(-1)]).astype(np.uint32)
    nxt = np.roll(a, -1)
    e_from = np.concatenate([bg.edge_from, np.stack([a, nxt + 1], axis=1).reshape(-1)]).astype(np.uint32)   # a_i -> a_{i+1}; mirror(a_{i+1}) -> mirror(a_i)
    e_to = np.concatenate([bg.edge_to, np.stack([nxt, a + 1], axis=1).reshape(-1)]).astype(np.uint32)
    e_w = np.concatenate([bg.edge_weight, np.full(2 * ring, 5, bg.edge_weight.dtype)])
    G = api.Bigraph.from_edges(mirror, e_from, e_to, e_w)
    n_orig = G.original_edge_count()

    def run(mode):
        lim, ed = api.GreedytigAlgorithm.compute_tigs_np(G, api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode))
        cum, _ = gpu_props.check_tigs(torch, G, lim, ed, k)
        G.reset()
        starts = np.concatenate([[0], lim[:-1]]).astype(np.int64)
        lens = lim.astype(np.int64) - starts
        in_ring = ed.astype(np.int64) >= n_orig - 2 * ring
        in_ring &= ed.astype(np.int64) < n_orig
        ring_tigs = np.unique(np.searchsorted(lim.astype(np.int64), np.nonzero(in_ring)[0], side="right"))
        return len(lim), cum, [int(lens[t]) for t in ring_tigs]

    ref = run(api.EulerMode.HostReferenceOrder)
    dev = run(api.EulerMode.Device)
    assert ref[2] == [ring] and dev[2] == [ring]          # the ring: one tig holding each of its unitigs once, in either order
    assert dev[0] == ref[0] and dev[1] == ref[1]          # (no self-mirror node: tig count and cumulative length are invariant)


def test_device_euler_is_reproducible(gpu):
    """No step of the device decomposition depends on thread timing: two runs give identical walks."""
    from matchtigs_amd import api, synth

    bg = synth.g_csr(200000, seed=13, k=31, mean_out_degree=1.2)   # many components of very different sizes
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, bg.k)
    a = G.euler_cycles_device_np()
    for _ in range(3):
        b = G.euler_cycles_device_np()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    check_bicycles(G.export(), a[0], a[1])


def test_device_euler_splitter_mark_and_bitmap_forms_agree(gpu):
    """The segment walks find the next splitter by a mark in bit 31 of its predecessor's successor word (fewer than 2^31 darts) or by
    a bitmap lookup per step (always from 2^31 darts on; forced here by mtg_set_euler_device_tuning): identical walks."""
    from matchtigs_amd import _lib, api, synth

    L = _lib.load()
    for seed, n, deg in ((13, 200000, 1.2), (5, 60000, 2.0), (9, 300, 1.5)):
        bg = synth.g_csr(n, seed=seed, k=31, mean_out_degree=deg)
        G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
        G.make_eulerian(0, bg.k)
        a = G.euler_cycles_device_np()
        L.mtg_set_euler_device_tuning(1)
        try:
            b = G.euler_cycles_device_np()
        finally:
            L.mtg_set_euler_device_tuning(0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (seed, n)
        check_bicycles(G.export(), a[0], a[1])


def test_device_euler_walks_from_the_recorded_sequence_equal_the_second_walk(gpu):
    """The measuring walk records the darts it passes (ballot-packed rows in chunks of a bump-allocated buffer) and the closed walks
    are written from that record; mtg_set_euler_device_tuning(4) writes them by a second walk through the successor array, as until
    round 4: identical walks -- small graphs (a few lanes per wave), both splitter tests, and a graph of 9 M darts."""
    from matchtigs_amd import _lib, api, synth

    L = _lib.load()
    cases = [synth.g_csr(n, seed=seed, k=31, mean_out_degree=deg) for seed, n, deg in ((13, 200000, 1.2), (5, 60000, 2.0), (9, 300, 1.5), (2, 40, 1.0))]
    graphs = []
    for bg in cases:
        G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
        G.make_eulerian(0, bg.k)
        graphs.append(G)
    big = synth.g_csr_device(1 << 21, seed=11, k=31, mean_out_degree=1.4)
    big.make_eulerian(0, 31)
    graphs.append(big)
    for G in graphs:
        a = G.euler_cycles_device_np()
        for flags in (4, 5, 8):  # second walk; second walk + bitmap splitter test; chunk tables of one entry (the overflow fallback)
            L.mtg_set_euler_device_tuning(flags)
            try:
                b = G.euler_cycles_device_np()
            finally:
                L.mtg_set_euler_device_tuning(0)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), flags
        L.mtg_set_euler_device_tuning(1)  # recorded sequence + bitmap splitter test
        try:
            c = G.euler_cycles_device_np()
        finally:
            L.mtg_set_euler_device_tuning(0)
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
        check_bicycles(G.export(), a[0], a[1])


def test_device_euler_two_level_ranking_equals_flat_pointer_jumping(gpu):
    """From 2^16 splitters on the reduced list is ranked in two levels (every 32nd splitter and every root; pointer jumping over
    those only); mtg_set_euler_device_tuning(2) forces the flat form: identical walks on a graph of 9 M darts, Eulertigs and greedy."""
    from matchtigs_amd import _lib, api, synth

    L = _lib.load()
    k = 31
    G = synth.g_csr_device(1 << 21, seed=11, k=k, mean_out_degree=1.4)
    G.make_eulerian(0, k)
    a = G.euler_cycles_device_np()
    assert len(a[1]) >= (1 << 21)  # biedges: twice as many darts, one splitter per 64 of them: >= 2^16 splitters
    for flags in (2, 3):  # flat ranking; flat ranking + bitmap splitter test
        L.mtg_set_euler_device_tuning(flags)
        try:
            b = G.euler_cycles_device_np()
        finally:
            L.mtg_set_euler_device_tuning(0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), flags
    check_bicycles(G.export(), a[0], a[1])
