"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tiers: T1 candidate lists, T2 ordered pair list, T3 #tigs / cumulative length, T4 tig edge sequences.
All integer work: the bar is bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(product_lib):
    import torch

    if product_lib.mtg_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("these tests need a GPU: the matchtigs_amd hot path has no CPU fallback")
    return torch


def _graphs():
    from matchtigs_amd import synth

    return [
        ("csr-small-k5", synth.g_csr(300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05)),
        ("csr-k9", synth.g_csr(5000, seed=11, k=9, mean_weight=3.0, mean_out_degree=1.8, self_mirror_frac=0.01)),
        ("csr-k31", synth.g_csr(30000, seed=1, k=31)),
        ("csr-k31-dense", synth.g_csr(20000, seed=2, k=31, mean_out_degree=2.2, mean_weight=4.0)),
        ("csr-k63", synth.g_csr(8000, seed=5, k=63, mean_weight=10.0)),
    ]


GRAPHS = None


def graphs():
    global GRAPHS
    if GRAPHS is None:
        GRAPHS = _graphs()
    return GRAPHS


def _oracle(oracle, bg):
    return oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)


def _gpu_candidates(bg, plan=0, lo=None, hi=None):
    from matchtigs_amd import api, torch_glue

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, bg.k)
    dev.set_plan(plan)
    S = dev.classify(torch_glue.current_stream_ptr())
    lo = 0 if lo is None else lo
    hi = S if hi is None else hi
    bufs = torch_glue.run_sssp(dev, lo, hi)
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    return G, dev, S, start, count, pool


@pytest.mark.parametrize("idx", range(5))
def test_classification_matches_oracle(gpu, oracle, idx):
    name, bg = graphs()[idx]
    from matchtigs_amd import api

    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, bg.k)
    S = dev.classify()
    on, mu, li = dev.classify_download()
    o_on, o_live, o_mult, _, _ = _oracle(oracle, bg).classify()
    assert S == len(o_on)
    assert np.array_equal(on, o_on), name
    assert np.array_equal(mu.astype(np.int64), o_mult), name
    assert np.array_equal(li, o_live), name


def _pruned_search_units(bg):
    """Independent count of what the goal-directed search visits (mtg_sssp_count_visited): lb(v) = distance to the nearest initial
    in-node by a multi-source Dijkstra over the reversed edges, lb+(v) = min over v's out-edges of w + lb(head) = distance to the
    nearest in-node BEYOND v; a source is searched iff lb+(source) <= k-1; a search settles v over u -> v at distance d + w only if
    d + w + lb(v) <= k-1, and expands it (relaxes its out-edges) only if d + w + lb+(v) <= k-1: an in-node with nothing beyond it is
    recorded from its parent's block and its own block is never read. Returns (searched sources, settled, relaxed edges)."""
    import heapq

    V, K1 = bg.n_nodes, bg.k - 1
    fr, to = bg.edge_from.astype(np.int64), bg.edge_to.astype(np.int64)
    w = np.minimum(bg.edge_weight.astype(np.int64), bg.k)
    mir = bg.mirror.astype(np.int64)

    def csr(a, b):
        order = np.argsort(a, kind="stable")
        row = np.zeros(V + 1, np.int64)
        np.add.at(row, a + 1, 1)
        return np.cumsum(row), b[order], w[order]

    row, col, ww = csr(fr, to)
    rrow, rcol, rww = csr(to, fr)
    odeg = np.diff(row)
    sm = mir == np.arange(V)
    diff = np.where(sm, odeg & 1, odeg - odeg[mir])
    target = diff > 0
    source = np.where(sm, diff != 0, diff < 0)
    INF = 1 << 40
    lb = np.full(V, INF, np.int64)
    heap = [(0, int(t)) for t in np.nonzero(target)[0]]
    lb[target] = 0
    while heap:
        d, u = heapq.heappop(heap)
        if d > lb[u]:
            continue
        for i in range(rrow[u], rrow[u + 1]):
            nd = d + int(rww[i])
            if nd <= K1 and nd < lb[rcol[i]]:
                lb[rcol[i]] = nd
                heapq.heappush(heap, (nd, int(rcol[i])))
    lbp = np.full(V, INF, np.int64)
    np.minimum.at(lbp, fr, w + lb[to])
    searched = settled = relaxed = 0
    for s in np.nonzero(source)[0]:
        if lbp[s] > K1:
            continue
        searched += 1
        dist = {int(s): 0}
        heap = [(0, int(s))]
        while heap:
            d, u = heapq.heappop(heap)
            if d > dist[u]:
                continue
            settled += 1
            if u != s and d + lbp[u] > K1:
                continue
            relaxed += int(row[u + 1] - row[u])
            for i in range(row[u], row[u + 1]):
                v, nd = int(col[i]), d + int(ww[i])
                if nd + lb[v] <= K1 and nd < dist.get(v, INF):
                    dist[v] = nd
                    heapq.heappush(heap, (nd, v))
    return searched, settled, relaxed


@pytest.mark.parametrize("plan", [0, 1, 2, 3, 4, 6])  # 0 = default (path enumeration level + cooperative cascade), 1 = cooperative cascade only,
#   2 / 3 = plan 0 with the quad-cooperative gathers large graphs get / the per-lane ones, 4 / 6 = plans 0 / 2 without the
#   goal-directed pruning (mtg_set_sssp_plan)
@pytest.mark.parametrize("idx", range(5))
def test_t1_candidate_lists(gpu, oracle, idx, plan):
    name, bg = graphs()[idx]
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k)
    assert S == len(o_on)
    assert np.array_equal(count.astype(np.uint64), np.diff(off)), name
    # lists are contiguous per source but sources may land anywhere in the pool
    idx_arr = np.concatenate([np.arange(s, s + c, dtype=np.int64) for s, c in zip(start, count)]) if S else np.zeros(0, np.int64)
    assert np.array_equal(pool[idx_arr], keys), name
    assert int(count.sum()) <= len(pool)  # the pool cursor includes unused tails of block-local chunks
    # unit counters of the counting kernel == oracle's full-ball Dijkstra counters
    cnt = dev.sssp_count(0, S)
    assert cnt["settled_nodes"] == st["settled_nodes"], (name, cnt, st)
    assert cnt["relaxed_edges"] == st["relaxed_edges"], (name, cnt, st)
    assert cnt["emitted"] == len(keys)
    assert dev.prunes() == (bg.k <= 255 and plan in (0, 2, 3))
    if plan == 0:  # ... and the units of the pruned search == an independent count of what the lower bounds leave
        vis = dev.sssp_count_visited(0, S)
        searched, settled, relaxed = _pruned_search_units(bg)
        assert (vis["sources"], vis["settled_nodes"], vis["relaxed_edges"], vis["emitted"]) == (searched, settled, relaxed, len(keys)), (name, vis)
        assert dev.last_searched_sources() == searched
        assert int((count > 0).sum()) <= searched <= S


@pytest.mark.parametrize("plan", [0, 1, 2, 3])
@pytest.mark.parametrize("idx", range(5))
def test_t1_device_graph_without_lower_bounds_and_upgraded(gpu, oracle, idx, plan):
    """mtg_device_create_opts(MTG_DEVICE_NO_LOWER_BOUNDS) -- the device graph mtg_compute_tigs_cfg builds for its one search: plain
    weights, no pruning, full balls -- gives the oracle's candidate lists; mtg_device_build_lower_bounds then rewrites the blocks into
    the 8:8 format (the same kernels as a build with bounds), after which the search prunes and gives the same lists again."""
    from matchtigs_amd import api, torch_glue

    name, bg = graphs()[idx]
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, bg.k, lower_bounds=False)
    dev.set_plan(plan)
    assert not dev.prunes() and dev.lower_bounds_ms() == 0.0

    def lists():
        S = dev.classify(torch_glue.current_stream_ptr())
        start, count, pool = torch_glue.candidates_to_numpy(torch_glue.run_sssp(dev, 0, S))
        idx_arr = np.concatenate([np.arange(s, s + c, dtype=np.int64) for s, c in zip(start, count)]) if S else np.zeros(0, np.int64)
        return S, count, pool[idx_arr]

    S, count, got = lists()
    assert S == len(o_on) and np.array_equal(count.astype(np.uint64), np.diff(off)) and np.array_equal(got, keys), name
    cnt = dev.sssp_count(0, S)
    assert (cnt["settled_nodes"], cnt["relaxed_edges"], cnt["emitted"]) == (st["settled_nodes"], st["relaxed_edges"], len(keys))
    ms = dev.build_lower_bounds()
    if bg.k <= 255:
        assert ms > 0 and dev.prunes() == (plan in (0, 2, 3))
    else:
        assert ms == 0.0 and not dev.prunes()
    S2, count2, got2 = lists()
    assert S2 == S and np.array_equal(count2, count) and np.array_equal(got2, keys), name
    if dev.prunes():  # the same units as a device graph built with its bounds
        ref = api.DeviceGraph(G, bg.k)
        ref.set_plan(plan)
        ref.classify()
        a, b = dev.sssp_count_visited(0, S), ref.sssp_count_visited(0, S)
        for key in ("sources", "settled_nodes", "relaxed_edges", "emitted"):  # (relax_attempts of the label-correcting levels depend on thread timing)
            assert a[key] == b[key], (name, key)


@pytest.mark.parametrize("plan", [0, 2, 4])
def test_t1_long_lists_from_the_enumeration_level(gpu, oracle, plan):
    """Short unitigs and many in-nodes: the enumeration level itself finishes sources with 5-8, 9-16 and more than 16 candidates
    (its three post-pass length classes: work-list compaction + lane-parallel sort) and lists that name a node along two paths."""
    from matchtigs_amd import synth

    bg = synth.g_csr(40000, seed=7, k=31, mean_out_degree=1.25, mean_weight=3.0)
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    levels = dev.last_sssp_levels()
    handed_on = levels[1]["sources"] if len(levels) > 1 else 0
    o_on, off, keys, _ = _oracle(oracle, bg).candidate_lists(bg.k)
    lens = np.diff(off).astype(np.int64)
    # (a source with more than 32 candidates -- a full home block and a full extension block -- cannot finish in the level; the cascade
    # took over fewer of the others than there are lists of 5 to 32 candidates, so the level itself finished lists that long: its
    # post-pass -- the work list, its compaction by length class, the lane-parallel sorts -- ran on them)
    others_handed_on = handed_on - int((lens > 32).sum())
    assert int(((lens >= 5) & (lens <= 32)).sum()) > others_handed_on >= 0, (handed_on, np.bincount(lens)[:34])
    assert np.array_equal(count.astype(np.int64), lens)
    idx_arr = np.concatenate([np.arange(s, s + c, dtype=np.int64) for s, c in zip(start, count)])
    assert np.array_equal(pool[idx_arr], keys)


@pytest.mark.parametrize("plan", [8, 10, 12])  # plans 0 / 2 / 4 with the enumeration level on ONE workgroup (mtg_set_sssp_plan + 8)
@pytest.mark.parametrize("which", ["long lists", "unitig-like"])
def test_t1_one_workgroup_takes_chunk_after_chunk(gpu, oracle, plan, which):
    """What a wave of the enumeration level does BETWEEN its chunks of 64 sources only happens when it takes several of them -- with the
    full grid that needs hundreds of thousands of sources. On one workgroup (four waves) a graph of some ten thousand sources takes every
    wave through dozens of chunks: the record table turns over (the records of a chunk leave together, in the sources' order, two
    chunks later; a source that outlives them stores its own), the key ring wraps and crosses pool chunks (rows of 64 keys, the partial
    row of a chunk that ends, lists longer than the ring), extension blocks go back and forth. Lists compared with the oracle's in full."""
    from matchtigs_amd import synth

    if which == "long lists":
        bg = synth.g_csr(40000, seed=11, k=31, mean_out_degree=1.25, mean_weight=3.0)
    else:
        bg = synth.g_csr(120000, seed=12, k=31)
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    assert dev.set_plan(plan) == plan
    levels = dev.last_sssp_levels()
    assert "sssp_enum_kernel" in levels[0]["kernel"]
    searched = dev.last_searched_sources() if plan != 12 else S
    assert searched > 4 * 64 * 8, searched  # (every wave of the one workgroup took more than eight chunks)
    o_on, off, keys, _ = _oracle(oracle, bg).candidate_lists(bg.k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off))
    idx_arr = np.concatenate([np.arange(s, s + c, dtype=np.int64) for s, c in zip(start, count)])
    assert np.array_equal(pool[idx_arr], keys)


def test_t1_source_subrange(gpu, oracle):
    name, bg = graphs()[2]
    o_on, off, keys, _ = _oracle(oracle, bg).candidate_lists(bg.k)
    S = len(o_on)
    lo, hi = S // 3, S // 3 + 1000
    G, dev, S2, start, count, pool = _gpu_candidates(bg, 0, lo, hi)
    assert S2 == S
    assert np.array_equal(count.astype(np.uint64), np.diff(off)[lo:hi])
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys[int(off[lo]):int(off[hi])])


@pytest.mark.parametrize("idx", range(5))
def test_t2_t3_t4_pairs_and_tigs(gpu, oracle, idx):
    name, bg = graphs()[idx]
    from matchtigs_amd import api

    k = bg.k
    G, dev, S, start, count, pool = _gpu_candidates(bg, 0)
    on, mu, li = dev.classify_download()
    pairs = G.replay_claims(on, mu, li, start, count, pool)
    og = _oracle(oracle, bg)
    o_pairs, _ = og.greedy_pairs_np(k)
    assert len(pairs) == len(o_pairs), name
    for f in ("out", "in", "dist"):
        assert np.array_equal(pairs[f], o_pairs[f]), (name, f)
    # whole path through the one-shot operator API
    G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    tigs = api.GreedytigAlgorithm.compute_tigs(G2, api.GreedytigAlgorithmConfiguration.new(1, k))
    og2 = _oracle(oracle, bg)
    want, _ = og2.compute_greedytigs(k)
    assert len(tigs) == len(want), name          # T3
    assert tigs == want, name                     # T4
    ex = G2.export()
    w = ex["edge_weight"]
    cum = sum(int(w[t].sum()) + k - 1 for t in map(np.array, tigs))
    cum_o = sum(sum(og2.edge(e)[2] for e in t) + k - 1 for t in want)
    assert cum == cum_o
    # invariants lifted from the reference's asserts
    dummy = ex["edge_dummy_id"] != 0
    for t in tigs[:2000]:
        assert not dummy[t[0]] and not dummy[t[-1]]  # greedytigs/mod.rs:794-798
    matched = dummy & (ex["edge_weight"] < k)
    assert (ex["edge_weight"][matched] >= 1).all()


@pytest.mark.parametrize("plan", [0, 1, 2])
def test_overflow_levels_big_balls(gpu, oracle, plan):
    """Unit weights + out-degree ~3 + k=31 make balls far larger than the level-0 budgets: the larger levels must agree."""
    from matchtigs_amd import synth

    bg = synth.g_csr(3000, seed=9, k=31, mean_out_degree=2.6, mean_weight=1.0, self_mirror_frac=0.01)
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    assert len(dev.last_sssp_levels()) > 1, "test graph should overflow level 0"
    cnt = dev.sssp_count(0, S)
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off))
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys)
    assert cnt["settled_nodes"] == st["settled_nodes"] and cnt["relaxed_edges"] == st["relaxed_edges"]


@pytest.mark.parametrize("plan", [0, 1, 2])
def test_deepest_levels_huge_balls(gpu, oracle, plan):
    """Balls above 16384 nodes only fit the last level (table in a global workspace): a unit-weight graph whose
    (k-1)-balls cover most of its 36000 nodes, on a slice of the sources (the oracle would need minutes for all)."""
    from matchtigs_amd import synth

    bg = synth.g_csr(18000, seed=21, k=31, mean_out_degree=2.6, mean_weight=1.0, self_mirror_frac=0.0)
    lo, hi = 100, 148
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan, lo, hi)
    levels = dev.last_sssp_levels()
    assert len(levels) == 5, levels   # level 0 + every later cooperative level down to the global-workspace one
    assert levels[-1]["kernel"].endswith(",global>")
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k, lo, hi)
    assert st["settled_nodes"] > 16384 * 8   # the balls really are that large
    assert np.array_equal(count.astype(np.uint64), np.diff(off)[lo:hi])
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys)


def test_ball_beyond_every_table_level(gpu, oracle):
    """No abort on legal input (the reference's search has no size limit, greedytigs/mod.rs:548-551): a unit-weight graph whose
    (k-1)-balls hold more than the 2^22 entries of the last cooperative level's table; the dense level finishes such sources, and
    the lists equal the oracle's Dijkstra lists. Two sources (every source of this graph has such a ball)."""
    from matchtigs_amd import synth

    bg = synth.g_csr(5_000_000, seed=2, k=120, mean_out_degree=1.7, mean_weight=1.0, self_mirror_frac=0.0)
    lo, hi = 1000, 1002
    G, dev, S, start, count, pool = _gpu_candidates(bg, 0, lo, hi)
    levels = dev.last_sssp_levels()
    assert levels[-1]["kernel"].startswith("dense_relax_kernel") and levels[-1]["sources"] >= 1, levels
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k, lo, hi)
    assert st["settled_nodes"] > (1 << 22)   # the balls really are that large
    assert np.array_equal(count.astype(np.uint64), np.diff(off)[lo:hi])
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys)


def test_many_candidates_in_a_mid_sized_ball(gpu, oracle):
    """A ball that fits the global-workspace level's table but holds far more in-nodes than its quadratic ranking should ever see
    (here ~100 000 per source): such sources go on to the dense level instead. Lists equal the oracle's."""
    from matchtigs_amd import synth

    bg = synth.g_csr(250_000, seed=3, k=100, mean_out_degree=1.8, mean_weight=1.0, self_mirror_frac=0.0)
    lo, hi = 500, 504
    G, dev, S, start, count, pool = _gpu_candidates(bg, 0, lo, hi)
    levels = dev.last_sssp_levels()
    assert levels[-1]["kernel"].startswith("dense_relax_kernel"), levels
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k, lo, hi)
    assert np.diff(off)[lo:hi].max() > 32768
    assert np.array_equal(count.astype(np.uint64), np.diff(off)[lo:hi])
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys)


@pytest.mark.parametrize("k", [70_000, 3_000_000])
def test_k_beyond_16_bits(gpu, oracle, k):
    """k above 65535 (weights still fit 16 bits): distances beyond 15 bits skip the enumeration level, beyond 21 bits every
    source takes the dense level. Candidate lists and tigs equal the oracle's."""
    from matchtigs_amd import api, synth

    bg = synth.g_csr(150, seed=6, k=31, mean_out_degree=1.4, mean_weight=6.0, self_mirror_frac=0.02)
    bg.k = k
    G, dev, S, start, count, pool = _gpu_candidates(bg, 0)
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off))
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)]) if len(start) else np.zeros(0, np.uint64)
    assert np.array_equal(got, keys)
    want, _ = _oracle(oracle, bg).compute_greedytigs(k)
    H = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    assert api.GreedytigAlgorithm.compute_tigs(H, api.GreedytigAlgorithmConfiguration(1, k)) == want


@pytest.mark.parametrize("plan", [0, 2, 1])
@pytest.mark.parametrize("k,mean_weight", [(255, 40.0), (256, 40.0), (300, 50.0), (1000, 150.0), (20000, 2500.0)])
def test_k_around_the_two_weight_formats(gpu, oracle, k, mean_weight, plan):
    """k <= 255: the blocks carry weight | weight + lower bound in one byte each and the enumeration level prunes; 256 <= k < 32768:
    plain 16-bit weights, the enumeration level without pruning. Both sides of the border and well inside the second range, under
    the per-lane and the quad gathers and the plain cascade: candidate lists and full-ball counters equal the oracle's."""
    from matchtigs_amd import synth

    bg = synth.g_csr(4000, seed=k, k=k, mean_out_degree=1.7, mean_weight=mean_weight, self_mirror_frac=0.01)
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    assert dev.prunes() == (k <= 255 and plan != 1)
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off)), (k, plan)
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)]) if S else np.zeros(0, np.uint64)
    assert np.array_equal(got, keys), (k, plan)
    assert len(keys) > 500  # (the bound reaches well beyond single edges)
    levels = dev.last_sssp_levels()
    assert ("sssp_enum_kernel" in levels[0]["kernel"]) == (plan != 1)
    cnt = dev.sssp_count(0, S)
    assert (cnt["settled_nodes"], cnt["relaxed_edges"], cnt["emitted"]) == (st["settled_nodes"], st["relaxed_edges"], len(keys))


@pytest.mark.parametrize("plan", [0, 1, 2])
def test_high_degree_nodes_use_spill_adjacency(gpu, oracle, plan):
    """Nodes with more than 4 out-edges (not a de Bruijn graph, but legal through the C-ABI) take the CSR spill path."""
    from matchtigs_amd import synth

    bg = synth.g_csr(2000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9)
    deg = np.bincount(bg.edge_from, minlength=bg.n_nodes)
    assert deg.max() > 4
    G, dev, S, start, count, pool = _gpu_candidates(bg, plan)
    o_on, off, keys, st = _oracle(oracle, bg).candidate_lists(bg.k)
    assert np.array_equal(count.astype(np.uint64), np.diff(off))
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)])
    assert np.array_equal(got, keys)


def test_clib_abi_end_to_end(gpu, oracle):
    """matchtigs_initialise_graph -> merge_nodes -> build_graph -> compute_tigs(5 / 3 / 1) against the oracle's clib restatement."""
    from matchtigs_amd import api, synth

    ug = synth.g_seq(3000, seed=2, k=15, haplotypes=3, sub_rate=0.03)
    for alg in (5, 3, 1):
        n, eo, io, lo = api.clib_compute_tigs(ug.weights, ug.links, alg, 1, ug.k)
        og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
        n_o, eo_o, io_o, lo_o = og.clib_compute_tigs(alg, ug.k)
        assert n == n_o
        assert np.array_equal(lo, lo_o) and np.array_equal(eo, eo_o) and np.array_equal(io, io_o)


def test_smoke_entry(gpu):
    import __graft_entry__ as ge

    ge.smoke()


def test_full_bench_size_properties(gpu, oracle):
    """At the bench workload's full size (|E| = 2^24 nominal) the oracle's Euler stage is too slow to run in a test, so the
    GPU path is checked through size-independent properties, plus exact pair parity (the oracle's claim loop takes ~30 s)."""
    from matchtigs_amd import api, synth, torch_glue

    k = 31
    bg = synth.g_csr(int((1 << 24) / 1.5 / 2), seed=1, k=k)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    dev = api.DeviceGraph(G, k)
    S = dev.classify(torch_glue.current_stream_ptr())
    bufs = torch_glue.run_sssp(dev, 0, S)
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    on, mu, li = dev.classify_download()
    # candidate lists: strictly ascending keys (sortedness by (distance, node), no duplicates), targets only, bound respected
    tot = int(count.sum())
    cnt64 = count.astype(np.int64)
    seg_begin = np.cumsum(cnt64) - cnt64
    idx = np.repeat(start.astype(np.int64), cnt64) + (np.arange(tot, dtype=np.int64) - np.repeat(seg_begin, cnt64))
    keys = pool[idx]
    seg_first = np.zeros(tot, bool)
    seg_first[seg_begin[cnt64 > 0]] = True
    assert (np.diff(keys.astype(np.int64))[~seg_first[1:]] > 0).all()
    nodes, dist = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64), (keys >> np.uint64(32)).astype(np.int64)
    assert li[nodes].all() and dist.min() >= 1 and dist.max() <= k - 1
    assert (nodes != np.repeat(on.astype(np.int64), cnt64)).all()      # forbid_source_target
    cnt = dev.sssp_count(0, S)
    assert cnt["emitted"] == tot
    # T2 at full size: exact pair list vs the oracle's reference-style claim loop
    pairs = G.replay_claims(on, mu, li, start, count, pool)
    og = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    want, st = og.greedy_pairs_np(k)
    assert len(pairs) == len(want) and all(np.array_equal(pairs[f], want[f]) for f in ("out", "in", "dist"))
    # tigs: every unitig appears exactly once (in one orientation), tigs start/end with unitigs, dummy weights in range
    lim, edges = api.finish_greedytigs_np(G, pairs, k)
    ex = G.export()
    n_orig = bg.n_edges
    orig = edges[edges < n_orig]
    assert len(orig) == n_orig // 2 and len(np.unique(orig >> 1)) == n_orig // 2
    starts = np.r_[0, lim[:-1]].astype(np.int64)
    assert (edges[starts] < n_orig).all() and (edges[lim.astype(np.int64) - 1] < n_orig).all()
    dummies_in_tigs = edges[edges >= n_orig]
    w = ex["edge_weight"][dummies_in_tigs]
    assert (w >= 1).all() and (w <= k - 1).all()                         # only matched dummies survive inside tigs
    # Eulerian after Eulerisation: out-degree == in-degree per non-self-mirror node, even degree for self-mirrors
    outd = np.bincount(ex["edge_from"], minlength=bg.n_nodes)
    ind = np.bincount(ex["edge_to"], minlength=bg.n_nodes)
    sm = ex["mirror"] == np.arange(bg.n_nodes)
    assert (outd[~sm] == ind[~sm]).all() and (outd[sm] % 2 == 0).all()
    # cumulative length identity: sum of unitig k-mers + kept dummy weights + (k-1) * #tigs
    cum = int(ex["edge_weight"][edges].sum()) + (k - 1) * len(lim)
    assert cum == int(bg.edge_weight[0::2].sum()) + int(w.sum()) + (k - 1) * len(lim)


def test_dijkstra_performance_data(gpu, oracle):
    """performance_data_type = Complete (greedytigs/mod.rs:176-197, 647-673): counters in the engine's terms, tied to the
    oracle's full-ball Dijkstra: one query per source, iterations = settled (source, node) pairs."""
    from matchtigs_amd import api

    name, bg = graphs()[2]
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    cfg = api.GreedytigAlgorithmConfiguration(4, bg.k, performance_data_type=api.PerformanceDataType.Complete,
                                              node_weight_array_type=api.NodeWeightArrayType.EpochNodeWeightArray)
    tigs = api.GreedytigAlgorithm.compute_tigs(G, cfg)
    pd = api.last_performance_data()
    og = _oracle(oracle, bg)
    o_on, off, keys, st = og.candidate_lists(bg.k)
    assert pd["dijkstras"] == len(o_on)
    assert pd["iterations"] == st["settled_nodes"] == pd["sum_max_distance_array_size"]
    assert pd["heap_pushes"] >= pd["iterations"] and pd["unnecessary_heap_elements"] == pd["heap_pushes"] - pd["iterations"]
    assert 1 <= pd["max_max_distance_array_size"] <= pd["max_max_heap_size"] <= pd["heap_pushes"]
    want, _ = _oracle(oracle, bg).compute_greedytigs(bg.k)
    assert tigs == want          # the counters never change results
    G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    api.GreedytigAlgorithm.compute_tigs(G2, api.GreedytigAlgorithmConfiguration.new(1, bg.k))
    assert api.last_performance_data()["dijkstras"] == 0   # None: nothing is collected
