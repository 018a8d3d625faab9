"""Exact parity at sizes the CPU oracle cannot hold (SURVEY 8c "maximum sizes"; BASELINE configs[3] / configs[4] stand-ins): the
GPU's classification compared in full, its candidate lists compared for a SAMPLE of sources and its pair list compared for a
PREFIX of sources -- all three against computations that share nothing with the product but the input edge arrays.

 * classification (greedytigs/mod.rs:222-255): degrees by torch scatter-adds over the exported edge arrays, the rule restated here.
 * candidate lists L(s) (greedytigs/mod.rs:324-335): the sample's searches can only ever look at out-edges of nodes within k - 1 of a
   sample source. That subgraph is extracted with torch (a multi-source bounded relaxation over all edges gives the distance to the
   nearest sample source; an edge u -> v of weight w belongs iff that distance of u plus w stays within the bound), its nodes are
   renumbered in order (the (distance, node) order of a list survives), and the ORACLE's Dijkstra (oracle/mtg_oracle.c
   shortest_path_lens, og_candidate_lists_given) runs on it under the full graph's in-node map. Lists must be EQUAL to the GPU's.
 * claim loop (greedytigs/mod.rs:301-523): the first N sources' claims depend on nothing but their own searches and the
   multiplicities / live bits of the nodes those searches reach, so the ORACLE's sequential loop with its own truncated searches
   (og_greedy_pairs_given) over the same subgraph, started from the full graph's classification, must produce exactly the first
   pairs of the GPU's pair list.

torch is the calculator of the extraction (as in tests/gpu_props.py); nothing of the product runs through this module."""
import numpy as np

EDGE_CHUNK = 1 << 27


class EdgeArrays:
    """The ORIGINAL edges of a graph on the GPU as torch tensors (from / to int32 with the bit pattern of the u32 ids, weight clamped to
    255 in uint8 -- only used with bounds below 255), + mirror."""

    def __init__(self, torch, G, device="cuda"):
        self.torch, self.device = torch, device
        E, self.V = G.original_edge_count(), G.node_count()
        self.E = E
        self.frm = torch.empty(E, dtype=torch.int32, device=device)
        self.to = torch.empty(E, dtype=torch.int32, device=device)
        self.w = torch.empty(E, dtype=torch.uint8, device=device)
        for lo in range(0, E, EDGE_CHUNK):
            n = min(EDGE_CHUNK, E - lo)
            ex = G.export_range(lo, n, ("edge_from", "edge_to", "edge_weight"))
            self.frm[lo:lo + n] = torch.from_numpy(ex["edge_from"].view(np.int32)).to(device)
            self.to[lo:lo + n] = torch.from_numpy(ex["edge_to"].view(np.int32)).to(device)
            self.w[lo:lo + n] = torch.from_numpy(np.minimum(ex["edge_weight"], 255).astype(np.uint8)).to(device)
            del ex
        self.mirror = torch.from_numpy(G.export_mirror().view(np.int32)).to(device)

    def ids(self, t):
        """int32 bit patterns of u32 ids -> int64 indices"""
        return t.to(self.torch.int64) & 0xFFFFFFFF


def classification(ea: EdgeArrays):
    """(out_nodes ascending [int64 tensor], live [uint8 tensor, V], mult [int8 tensor, V]) by greedytigs/mod.rs:229-245 with
    compute_eulerian_superfluous_out_biedges = out-degree - in-degree (a self-mirror node: out-degree mod 2)."""
    torch, dev, V = ea.torch, ea.device, ea.V
    outd = torch.zeros(V, dtype=torch.int32, device=dev)
    ind = torch.zeros(V, dtype=torch.int32, device=dev)
    for lo in range(0, ea.E, EDGE_CHUNK):
        hi = min(ea.E, lo + EDGE_CHUNK)
        one = torch.ones(hi - lo, dtype=torch.int32, device=dev)
        outd.index_add_(0, ea.ids(ea.frm[lo:hi]), one)
        ind.index_add_(0, ea.ids(ea.to[lo:hi]), one)
        del one
    sm = ea.ids(ea.mirror) == torch.arange(V, device=dev)
    diff = torch.where(sm, outd % 2, outd - ind)
    del outd, ind
    is_out = torch.where(sm, diff != 0, diff < 0)
    live = (torch.where(sm, diff != 0, diff > 0)).to(torch.uint8)
    out_nodes = torch.nonzero(is_out).flatten()
    mult = diff.to(torch.int8)
    return out_nodes, live, mult


def check_classification(ea: EdgeArrays, on, mu, li):
    """The GPU's classification download (numpy) equals the independent one, entry by entry. Returns the independent tensors."""
    torch = ea.torch
    out_nodes, live, mult = classification(ea)
    assert out_nodes.numel() == len(on), f"{out_nodes.numel()} sources by degrees, the GPU lists {len(on)}"
    CH = 1 << 28
    for lo in range(0, len(on), CH):
        assert bool((out_nodes[lo:lo + CH] == torch.from_numpy(on[lo:lo + CH].astype(np.int64)).to(ea.device)).all()), "out-node list differs"
    for lo in range(0, ea.V, CH):
        assert bool((live[lo:lo + CH] == torch.from_numpy(li[lo:lo + CH]).to(ea.device)).all()), "in-node map differs"
        assert bool((mult[lo:lo + CH].to(torch.int32) == torch.from_numpy(mu[lo:lo + CH]).to(ea.device)).all()), "multiplicities differ"
    return out_nodes, live, mult


def ball_subgraph(ea: EdgeArrays, source_nodes, bound):
    """Every edge u -> v (weight w) with dmin(u) + w <= bound, dmin = distance to the nearest of `source_nodes` (int64 tensor):
    a superset of what any bounded search from one of them can relax. -> numpy (edge ids ascending, from, to, weight)."""
    torch, dev, V = ea.torch, ea.device, ea.V
    assert bound < 255
    INF = 1 << 20
    dmin = torch.full((V,), INF, dtype=torch.int32, device=dev)
    dmin[source_nodes] = 0
    for _ in range(bound + 1):  # (weights are >= 1: a shortest path within the bound has at most `bound` edges)
        before = dmin.clone()
        for lo in range(0, ea.E, EDGE_CHUNK):
            hi = min(ea.E, lo + EDGE_CHUNK)
            cand = dmin[ea.ids(ea.frm[lo:hi])] + ea.w[lo:hi].to(torch.int32)
            ok = cand <= bound
            if bool(ok.any()):
                dmin.scatter_reduce_(0, ea.ids(ea.to[lo:hi][ok]), cand[ok], reduce="amin")
            del cand, ok
        if bool((before == dmin).all()):
            break
    else:
        raise AssertionError("the bounded relaxation did not settle")
    del before
    sel = []
    for lo in range(0, ea.E, EDGE_CHUNK):
        hi = min(ea.E, lo + EDGE_CHUNK)
        ok = dmin[ea.ids(ea.frm[lo:hi])] + ea.w[lo:hi].to(torch.int32) <= bound
        sel.append(torch.nonzero(ok).flatten() + lo)
        del ok
    eid = torch.cat(sel)
    del dmin, sel
    return (eid.cpu().numpy(), ea.ids(ea.frm[eid]).cpu().numpy(), ea.ids(ea.to[eid]).cpu().numpy(), ea.w[eid].cpu().numpy().astype(np.uint64))


class SubOracle:
    """The oracle's graph of a ball subgraph: nodes = endpoints of the selected edges, the given extra nodes and the mirrors of all of
    them, renumbered in ascending order of their ids in the full graph."""

    def __init__(self, oracle_lib, ea: EdgeArrays, edges, extra_nodes, live, mult):
        torch = ea.torch
        eid, frm, to, w = edges
        nodes = np.unique(np.concatenate([frm, to, np.asarray(extra_nodes, np.int64)]))
        mir = ea.ids(ea.mirror[torch.from_numpy(nodes).to(ea.device)]).cpu().numpy()
        nodes = np.unique(np.concatenate([nodes, mir]))
        t_nodes = torch.from_numpy(nodes).to(ea.device)
        self.nodes = nodes
        sub_mirror = np.searchsorted(nodes, ea.ids(ea.mirror[t_nodes]).cpu().numpy())
        assert np.array_equal(nodes[sub_mirror[sub_mirror]], nodes), "mirror is not an involution on the subgraph's nodes"
        self.live = live[t_nodes].cpu().numpy().astype(np.uint8)
        self.mult = mult[t_nodes].cpu().numpy().astype(np.int64)
        self.og = oracle_lib.OracleGraph.from_arrays(sub_mirror.astype(np.uint32), np.searchsorted(nodes, frm).astype(np.uint32),
                                                     np.searchsorted(nodes, to).astype(np.uint32), w)
        self.n_edges = len(eid)

    def local(self, node_ids):
        idx = np.searchsorted(self.nodes, node_ids)
        assert np.array_equal(self.nodes[idx], node_ids)
        return idx.astype(np.uint32)


def sample_indices(n_sources, n_prefix, n_tail, n_random, seed=12345):
    """Source indices (ascending, distinct): the first n_prefix, the last n_tail, n_random drawn without replacement in between."""
    if n_sources <= n_prefix + n_tail + n_random:
        return np.arange(n_sources, dtype=np.int64)
    rng = np.random.default_rng(seed)
    mid = rng.choice(n_sources - n_prefix - n_tail, size=n_random, replace=False) + n_prefix
    return np.unique(np.concatenate([np.arange(n_prefix), mid, np.arange(n_sources - n_tail, n_sources)])).astype(np.int64)


def gpu_lists_of(torch, bufs, idx):
    """The GPU's candidate lists of the sources `idx` (numpy int64, indices into the source list): (offsets, keys) as numpy."""
    dev = bufs.pool.device
    t_idx = torch.from_numpy(idx).to(dev)
    cnt = bufs.count[t_idx].to(torch.int64)
    off = torch.zeros(len(idx) + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(cnt, 0)
    tot = int(off[-1])
    if tot == 0:
        return off.cpu().numpy().astype(np.uint64), np.zeros(0, np.uint64)
    which = torch.repeat_interleave(torch.arange(len(idx), device=dev), cnt)
    pos = bufs.start[t_idx][which] + (torch.arange(tot, device=dev) - off[:-1][which])
    keys = bufs.pool[pos]
    return off.cpu().numpy().astype(np.uint64), keys.cpu().numpy().view(np.uint64)


def check_sampled_lists_and_prefix(torch, oracle_lib, G, bufs, on, mu, li, gpu_pairs, k, n_prefix=20000, n_tail=2000, n_random=80000, log=print,
                                   device="cuda"):
    """The three comparisons of the module's header on one graph. bufs: the GPU's candidate buffers of ALL sources; on / mu / li: its
    classification download; gpu_pairs: its pair list (structured numpy: out, in, dist). Returns a dict of counts."""
    assert k - 1 < 255
    ea = EdgeArrays(torch, G, device)
    out_nodes, live, mult = check_classification(ea, on, mu, li)
    S = int(out_nodes.numel())
    idx = sample_indices(S, n_prefix, n_tail, n_random)
    src_nodes = out_nodes[torch.from_numpy(idx).to(ea.device)]
    edges = ball_subgraph(ea, src_nodes, k - 1)
    sub = SubOracle(oracle_lib, ea, edges, src_nodes.cpu().numpy(), live, mult)
    # T1 on the sample: the oracle's full lists (its own Dijkstra, the full graph's in-node map) == the GPU's
    src_local = sub.local(src_nodes.cpu().numpy())
    o_off, o_keys = sub.og.candidate_lists_given(k, src_local, sub.live)
    o_keys = (o_keys & np.uint64(0xFFFFFFFF00000000)) | sub.nodes[(o_keys & np.uint64(0xFFFFFFFF)).astype(np.int64)].astype(np.uint64)
    g_off, g_keys = gpu_lists_of(torch, bufs, idx)
    assert np.array_equal(g_off, o_off), "sampled candidate lists differ in length"
    assert np.array_equal(g_keys, o_keys), "sampled candidate lists differ"
    # T2 on the prefix: the oracle's sequential claim loop (truncated searches against live bits) over the first sources
    n_pre = min(n_prefix, S)
    assert np.array_equal(idx[:n_pre], np.arange(n_pre))
    o_pairs, st = sub.og.greedy_pairs_given(k, src_local[:n_pre], sub.live, sub.mult)
    last = int(out_nodes[n_pre - 1]) if n_pre else -1
    assert bool(np.all(gpu_pairs["out"][1:] >= gpu_pairs["out"][:-1])), "the pair list is not in source order"
    m = int(np.searchsorted(gpu_pairs["out"], last, side="right"))
    assert m == len(o_pairs), f"{m} pairs from the first {n_pre} sources on the GPU, {len(o_pairs)} by the oracle's loop"
    assert np.array_equal(sub.nodes[o_pairs["out"]], gpu_pairs["out"][:m].astype(np.int64)), "pair prefix: out-nodes differ"
    assert np.array_equal(sub.nodes[o_pairs["in"]], gpu_pairs["in"][:m].astype(np.int64)), "pair prefix: in-nodes differ"
    assert np.array_equal(o_pairs["dist"], gpu_pairs["dist"][:m].astype(np.uint64)), "pair prefix: distances differ"
    res = dict(sources=S, sampled=len(idx), sampled_candidates=len(o_keys), subgraph_nodes=len(sub.nodes), subgraph_edges=sub.n_edges,
               prefix_sources=n_pre, prefix_pairs=m, highest_sampled_node=int(src_nodes.max()) if len(idx) else -1,
               oracle_queries=st["queries"])
    log(f"sampled parity: {res}")
    del ea
    return res
