"""Fuzz of TINY bigraphs (4-12 nodes): the corner rules of the reference that large random graphs rarely hit -- self-mirror nodes as
sources and targets, multiplicities 3 and 4, a source whose own mirror is a candidate (demand 1: skipped, demand >= 2: the
"self-mirror edge" that costs 2; greedytigs/mod.rs:351-358), stale targets (:454-458), equal-distance ties, distance exactly k-1,
scrambled mirror numberings (the Euleriser's exception rule, implementation/mod.rs:252-285), weights above k.

Three parties per graph: the C oracle, the independent Python restatement (tests/pyref.py) and the product.
  cpu mode: oracle == pyref on pairs, counters, greedy tigs and eulertigs; the product's host stages (claim replay on the oracle's
            lists, host finish, exported graph) == oracle.
  gpu mode: the HIP path == oracle: classification, candidate lists under plans 0 / 1 / 2 (T1), GPU claim replay (T2), device
            finish in reference order (T4) incl. the pairs resident in HBM, the one-shot operator, eulertigs, and the clib.rs C-ABI
            on the same graph given as unitig links.
A graph on which the REFERENCE panics (e.g. an odd number of odd self-mirror nodes and no in-node to pair the last one with,
implementation/mod.rs:496-498) is recognised by the Python restatement raising and is skipped -- the C parties abort() there, like
the reference. Runs as a child process of tests/test_fuzz_small.py, so that an abort() names its seed instead of ending pytest:
    python tests/fuzz_small.py cpu|gpu|cpu_medium|gpu_medium FIRST_SEED N_SEEDS
The *_medium modes run G-csr graphs of 50-3000 binodes with random parameters (k 3..300, degrees, unitig lengths, self-mirror share,
scrambled numbering): the same parties on whole operators, candidate lists and counters.
"""
from __future__ import annotations

import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def tiny_bigraph(seed: int):
    """(k, mirror, unitigs): 4-12 nodes, every node with at most 4 out- and 4 in-edges, weights 1..k (a tenth of them up to k+3)."""
    r = random.Random(0x9E3779B97F4A7C15 ^ seed)
    k = r.randint(3, 9)
    n_pairs = r.randint(1, 5)
    n_sm = r.choice((0, 0, 0, 1, 1, 2, 3))
    while 2 * n_pairs + n_sm < 4:
        n_pairs += 1
    while 2 * n_pairs + n_sm > 12:
        n_pairs -= 1
    V = 2 * n_pairs + n_sm
    ids = list(range(V))
    if r.random() < 0.5:  # mirror nodes not next to each other, self-mirror nodes anywhere
        r.shuffle(ids)
    mirror = [0] * V
    for p in range(n_pairs):
        a, b = ids[2 * p], ids[2 * p + 1]
        mirror[a], mirror[b] = b, a
    for s in range(n_sm):
        mirror[ids[2 * n_pairs + s]] = ids[2 * n_pairs + s]
    odeg = [0] * V
    unitigs = []
    for _ in range(r.randint(max(1, V // 2), 3 * V)):
        a, b = r.randrange(V), r.randrange(V)
        x = r.random()  # short unitigs mostly, so that in-nodes lie within k-1 of out-nodes; a tenth beyond k
        w = r.randint(1, max(1, k // 2)) if x < 0.6 else (r.randint(1, k) if x < 0.9 else r.randint(1, k + 3))
        fa, fb = a, mirror[b]  # the two darts a -> b and mirror(b) -> mirror(a)
        need = {}
        need[fa] = need.get(fa, 0) + 1
        need[fb] = need.get(fb, 0) + 1
        if any(odeg[x] + c > 4 for x, c in need.items()):
            continue
        for x, c in need.items():
            odeg[x] += c
        unitigs.append((a, b, w))
    if not unitigs:
        unitigs.append((0, mirror[0] if mirror[0] != 0 else (1 % V), 1))
    return k, mirror, unitigs


EVENTS: dict = {}  # how often the corner rules fired over the run (counted by the Python restatement's claim loop)


def reference_panics(arrs, k) -> bool:
    import helpers
    import pyref

    try:
        pyref.compute_greedytigs(helpers.py_graph(*arrs), k)
        pyref.compute_eulertigs(helpers.py_graph(*arrs), k)
    except (AssertionError, ValueError, IndexError, KeyError):
        return True
    return False


def run_cpu(seed: int) -> str:
    import helpers
    import pyref

    k, mirror, unitigs = tiny_bigraph(seed)
    arrs = helpers.unitigs_to_arrays(mirror, unitigs)
    if reference_panics(arrs, k):
        return "panic"
    og = helpers.oracle_graph(*arrs)
    pairs_o, st_o = og.greedy_pairs(k)
    pairs_p, st_p = pyref.greedy_pairs(helpers.py_graph(*arrs), k, EVENTS)
    assert pairs_o == pairs_p, "pairs"
    assert st_o["relaxed_edges"] == st_p["relaxed_edges"] and st_o["settled_nodes"] == st_p["settled_nodes"], "dijkstra counters"
    og2 = helpers.oracle_graph(*arrs)
    tigs_o, _ = og2.compute_greedytigs(k)
    tigs_p, _, _ = pyref.compute_greedytigs(helpers.py_graph(*arrs), k)
    assert tigs_o == tigs_p, "greedy tigs"
    assert og2.is_eulerian() and og2.no_consecutive_dummy_edges(k), "invariants"
    et_o = helpers.oracle_graph(*arrs).compute_eulertigs(k)
    assert et_o == pyref.compute_eulertigs(helpers.py_graph(*arrs), k), "eulertigs"
    # the product's host stages
    G = helpers.product_graph(*arrs)
    pr = helpers.product_pairs_from_oracle_lists(G, helpers.oracle_graph(*arrs), k)
    assert [(int(a), int(b), int(c)) for a, b, c in pr] == pairs_o, "product host replay"
    assert G.finish_greedytigs(pr, k) == tigs_o, "product host finish"
    ex, oe = G.export(), og2.edges()
    assert [e[0] for e in oe] == ex["edge_from"].tolist() and [e[1] for e in oe] == ex["edge_to"].tolist(), "product graph"
    assert [e[2] for e in oe] == ex["edge_weight"].tolist() and [e[3] for e in oe] == ex["edge_dummy_id"].tolist(), "product graph"
    return "ok:" + ("pairs" if pairs_o else "nopairs")


def device_order_checks(ex, tigs, k, ref_tigs, mirror, what):
    """A tig set in the device's own order: every unitig exactly once in one orientation, consecutive edges adjacent, tigs start and end
    with original edges, no two dummies in a row, no breaking edge inside (greedytigs/mod.rs:794-798, implementation/mod.rs:319-390);
    tig count and cumulative length equal the reference-order result's when the graph has no self-mirror node."""
    dummy = ex["edge_dummy_id"] != 0
    n_orig = int((~dummy).sum()) // 2
    seen = np.zeros(n_orig, np.int64)
    for t in tigs:
        a = np.asarray(t, dtype=np.int64)
        assert len(a) and not dummy[a[0]] and not dummy[a[-1]], what
        assert np.array_equal(ex["edge_to"][a[:-1]], ex["edge_from"][a[1:]]), what
        assert not (dummy[a[:-1]] & dummy[a[1:]]).any(), what
        assert (ex["edge_weight"][a[dummy[a]]] < k).all(), what
        np.add.at(seen, a[~dummy[a]] >> 1, 1)
    assert (seen == 1).all(), what
    if not any(int(mirror[v]) == v for v in range(len(mirror))):
        cum = lambda ts: sum(int(ex["edge_weight"][np.asarray(t, dtype=np.int64)].sum()) + k - 1 for t in ts)
        assert len(tigs) == len(ref_tigs) and cum(tigs) == cum(ref_tigs), what


def run_gpu(seed: int) -> str:
    import helpers
    from matchtigs_amd import api, torch_glue

    k, mirror, unitigs = tiny_bigraph(seed)
    arrs = helpers.unitigs_to_arrays(mirror, unitigs)
    if reference_panics(arrs, k):
        return "panic"
    og = helpers.oracle_graph(*arrs)
    o_on, o_live, o_mult, _, _ = og.classify()
    _, off, keys, _ = og.candidate_lists(k)
    pairs_o, _ = helpers.oracle_graph(*arrs).greedy_pairs(k)
    tigs_o, _ = helpers.oracle_graph(*arrs).compute_greedytigs(k)
    G = helpers.product_graph(*arrs)
    dev = api.DeviceGraph(G, k)
    S = dev.classify()
    on, mu, li = dev.classify_download()
    assert S == len(o_on) and np.array_equal(on, o_on) and np.array_equal(mu.astype(np.int64), o_mult) and np.array_equal(li, o_live), "classification"
    bufs = None
    for plan in (0, 1, 2):  # T1 under the enumeration level (per-lane and quad gathers) and the plain cascade
        dev.set_plan(plan)
        bufs = torch_glue.run_sssp(dev, 0, S)
        start, count, pool = torch_glue.candidates_to_numpy(bufs)
        assert np.array_equal(count.astype(np.uint64), np.diff(off)), f"list lengths, plan {plan}"
        got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)]) if S else np.zeros(0, np.uint64)
        assert np.array_equal(got, keys), f"candidate lists, plan {plan}"
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())  # T2
    assert [(int(a), int(b), int(c)) for a, b, c in pairs] == pairs_o, "GPU claim replay"
    n = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
    lim, ed = api.finish_greedytigs_resident_np(G, dev, k, finish_stage=api.FinishStage.Device)  # T4, pairs resident in HBM
    assert n == len(pairs_o) and [ed[(lim[i - 1] if i else 0):lim[i]].tolist() for i in range(len(lim))] == tigs_o, "device finish"
    G2 = helpers.product_graph(*arrs)
    assert api.GreedytigAlgorithm.compute_tigs(G2, api.GreedytigAlgorithmConfiguration.new(1, k)) == tigs_o, "operator"
    et_o = helpers.oracle_graph(*arrs).compute_eulertigs(k)
    assert api.EulertigAlgorithm.compute_tigs(helpers.product_graph(*arrs), api.EulertigAlgorithmConfiguration(k)) == et_o, "eulertigs"
    # device Euler mode (the tigs cut straight from the pairing, cut_first_device.hip; tiny graphs are full of trails without a
    # breaking dart -- self-loops, two-cycles, balanced components -- so this is where the splicing of those is exercised), and the
    # same mode through the closed walks: valid tig sets, and the reference's tig count and cumulative length where the order
    # cannot matter (no self-mirror node: SURVEY 8a invariance note)
    for no_cut_first in (False, True):
        api.set_finish_tuning(no_cut_first=no_cut_first)
        for algo in ("greedy", "euler"):
            G3 = helpers.product_graph(*arrs)
            if algo == "greedy":
                dt = api.GreedytigAlgorithm.compute_tigs(G3, api.GreedytigAlgorithmConfiguration(1, k, euler_mode=api.EulerMode.Device))
            else:
                dt = api.EulertigAlgorithm.compute_tigs(G3, api.EulertigAlgorithmConfiguration(k, euler_mode=api.EulerMode.Device))
            device_order_checks(G3.export(), dt, k, tigs_o if algo == "greedy" else et_o, mirror, f"device order, {algo}, no_cut_first={no_cut_first}")
    api.set_finish_tuning()
    # the clib.rs C-ABI on the same graph given as unitig links (its own node numbering: union-find over the unitig ends)
    links = helpers.links_of_bigraph(mirror, unitigs)
    weights = [w for (_, _, w) in unitigs]
    if reference_panics_links(weights, links, k):
        return "ok:clib-panic"
    import oracle_lib

    want = oracle_lib.OracleGraph.from_unitig_links(weights, links).clib_compute_tigs(5, k)
    n_t, eo, io, lo = api.clib_compute_tigs(weights, links, 5, 1, k)
    assert (n_t, list(eo), list(io), list(lo)) == (want[0], list(want[1]), list(want[2]), list(want[3])), "clib C-ABI"
    return "ok:" + ("pairs" if pairs_o else "nopairs")


def medium_bigraph(seed: int):
    """G-csr graphs of 50 to 3000 binodes with every parameter drawn at random (k from 3 to 300: both weight formats of the device
    graph; degrees, unitig lengths, self-mirror share), half of them with scrambled node numbering."""
    from matchtigs_amd import synth

    r = random.Random(0xD1B54A32D192ED03 ^ seed)
    k = r.choice((3, 5, 9, 15, 31, 31, 63, 127, 255, 300))
    bg = synth.g_csr(r.randint(50, 3000), seed=seed, k=k, mean_out_degree=r.uniform(0.6, 2.5), mean_weight=r.uniform(1.0, max(1.5, k / 2.0)),
                     self_mirror_frac=r.choice((0.0, 0.0, 0.01, 0.05, 0.1)), max_degree=4)
    if r.random() < 0.5:
        rng = np.random.default_rng(seed)
        perm = rng.permutation(bg.n_nodes).astype(np.uint32)  # old -> new
        mirror = np.empty_like(bg.mirror)
        mirror[perm] = perm[bg.mirror]
        bg = synth.Bigraph(mirror, perm[bg.edge_from], perm[bg.edge_to], bg.edge_weight.copy(), k)
    return bg


def run_cpu_medium(seed: int) -> str:
    import helpers
    import pyref

    bg = medium_bigraph(seed)
    k, arrs = bg.k, (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    if reference_panics(arrs, k):
        return "panic"
    tigs_o, _ = helpers.oracle_graph(*arrs).compute_greedytigs(k)
    tigs_p, _, _ = pyref.compute_greedytigs(helpers.py_graph(*arrs), k)
    assert tigs_o == tigs_p, "greedy tigs"
    assert helpers.oracle_graph(*arrs).compute_eulertigs(k) == pyref.compute_eulertigs(helpers.py_graph(*arrs), k), "eulertigs"
    G = helpers.product_graph(*arrs)
    pr = helpers.product_pairs_from_oracle_lists(G, helpers.oracle_graph(*arrs), k)
    assert G.finish_greedytigs(pr, k) == tigs_o, "product host stages"
    return "ok:" + ("pairs" if len(pr) else "nopairs")


def run_gpu_medium(seed: int) -> str:
    import helpers
    from matchtigs_amd import api, torch_glue

    bg = medium_bigraph(seed)
    k, arrs = bg.k, (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    if reference_panics(arrs, k):
        return "panic"
    og = helpers.oracle_graph(*arrs)
    _, off, keys, st = og.candidate_lists(k)
    tigs_o, _ = helpers.oracle_graph(*arrs).compute_greedytigs(k)
    G = helpers.product_graph(*arrs)
    dev = api.DeviceGraph(G, k)
    S = dev.classify()
    dev.set_plan(seed % 4)  # 0 / 2 / 3: enumeration level (pruned when k <= 255), 1: the plain cascade
    bufs = torch_glue.run_sssp(dev, 0, S)
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    assert np.array_equal(count.astype(np.uint64), np.diff(off)), "list lengths"
    got = np.concatenate([pool[int(s):int(s) + int(c)] for s, c in zip(start, count)]) if S else np.zeros(0, np.uint64)
    assert np.array_equal(got, keys), "candidate lists"
    cnt = dev.sssp_count(0, S)
    assert (cnt["settled_nodes"], cnt["relaxed_edges"]) == (st["settled_nodes"], st["relaxed_edges"]), "full-ball counters"
    n = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
    lim, ed = api.finish_greedytigs_resident_np(G, dev, k, finish_stage=api.FinishStage.Device)
    assert [ed[(lim[i - 1] if i else 0):lim[i]].tolist() for i in range(len(lim))] == tigs_o, "device finish, reference order"
    # device Euler mode: same number of tigs and cumulative length when no self-mirror node can put two breaking edges next to each other
    G.reset()
    lim_d, ed_d = api.finish_greedytigs_resident_np(G, dev, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    device_order_checks(G.export(), [ed_d[(lim_d[i - 1] if i else 0):lim_d[i]].tolist() for i in range(len(lim_d))], k, tigs_o, bg.mirror, "device Euler mode")
    api.set_finish_tuning(no_cut_first=True)  # ... and through the closed walks (the form before cut_first_device.hip)
    G.reset()
    n = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr())
    lim_w, ed_w = api.finish_greedytigs_resident_np(G, dev, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
    api.set_finish_tuning()
    device_order_checks(G.export(), [ed_w[(lim_w[i - 1] if i else 0):lim_w[i]].tolist() for i in range(len(lim_w))], k, tigs_o, bg.mirror, "device Euler mode, closed walks")
    et_o = helpers.oracle_graph(*arrs).compute_eulertigs(k)
    assert api.EulertigAlgorithm.compute_tigs(helpers.product_graph(*arrs), api.EulertigAlgorithmConfiguration(k)) == et_o, "eulertigs"
    return "ok:" + ("pairs" if n else "nopairs")


def reference_panics_links(weights, links, k) -> bool:
    import pyref

    try:
        pyref.compute_greedytigs(pyref.from_unitig_links(weights, links), k)
    except (AssertionError, ValueError, IndexError, KeyError):
        return True
    return False


def main():
    mode, first, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    fn = {"cpu": run_cpu, "gpu": run_gpu, "cpu_medium": run_cpu_medium, "gpu_medium": run_gpu_medium}[mode]
    tally = {}
    for seed in range(first, first + n):
        print(f"seed {seed}", flush=True)  # (the last line before an abort() names the graph)
        try:
            r = fn(seed)
        except AssertionError as e:
            if mode.endswith("medium"):
                print(f"MISMATCH seed {seed}: {e}", flush=True)
            else:
                k, mirror, unitigs = tiny_bigraph(seed)
                print(f"MISMATCH seed {seed}: {e}; k={k} mirror={mirror} unitigs={unitigs}", flush=True)
            sys.exit(1)
        tally[r] = tally.get(r, 0) + 1
    print("TALLY " + " ".join(f"{kk}={v}" for kk, v in sorted(tally.items())), flush=True)
    print("EVENTS " + " ".join(f"{kk}={v}" for kk, v in sorted(EVENTS.items())), flush=True)


if __name__ == "__main__":
    main()
