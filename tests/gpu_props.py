"""Size-independent properties of the hot path's outputs, evaluated with torch on the GPU so that they stay affordable at the
human-like (2^30) and pangenome-like (2^31) G-csr sizes (SURVEY 8c "maximum sizes"). Test infrastructure: the checks of
tests/test_gpu_configs.py and tools/scale_probe.py. torch is the calculator here, nothing under test runs through it.

Candidate lists (greedytigs/mod.rs:324-335 + App. A.1): keys strictly ascending per source, every node an initial in-node,
distance in [1, k-1], the source itself excluded.  Tigs (greedytigs/mod.rs:726-801): every unitig exactly once in one
orientation, consecutive edges of a tig adjacent, tigs start and end with original edges, only matched dummies (weight in [1, k-1]) inside tigs, the graph Eulerian
after Eulerisation (:708-716), cumulative-length identity."""
import numpy as np

CHUNK = 1 << 28


def check_candidates(torch, bufs, out_nodes, is_in_node, k, chunk_sources=1 << 26):
    """bufs: torch_glue.CandidateBuffers (device tensors); out_nodes / is_in_node: numpy classification. Returns #candidates.
    Sources are taken in chunks so that the check's own tensors stay small next to a 110-GB device graph."""
    dev = bufs.pool.device
    live = torch.from_numpy(np.ascontiguousarray(is_in_node)).to(dev)
    total = 0
    for lo in range(0, bufs.n, chunk_sources):
        hi = min(bufs.n, lo + chunk_sources)
        cnt = bufs.count[lo:hi].to(torch.int64)
        tot = int(cnt.sum())
        if tot == 0:
            continue
        total += tot
        seg_begin = torch.cumsum(cnt, 0) - cnt
        src = torch.repeat_interleave(torch.arange(hi - lo, device=dev), cnt)
        idx = bufs.start[lo:hi][src] + (torch.arange(tot, device=dev) - seg_begin[src])
        keys = bufs.pool[idx]
        del idx, seg_begin, cnt
        same = src[1:] == src[:-1]
        assert bool((keys[1:][same] > keys[:-1][same]).all()), "candidate keys not strictly ascending per source"
        del same
        nodes, dist = keys & 0xFFFFFFFF, keys >> 32
        assert bool(live[nodes].bool().all()), "a candidate is not an initial in-node"
        assert int(dist.min()) >= 1 and int(dist.max()) <= k - 1
        on = torch.from_numpy(out_nodes[lo:hi].astype(np.int64)).to(dev)
        assert bool((nodes != on[src]).all()), "a source lists itself"
        del keys, nodes, dist, on, src
    return total


def check_tigs(torch, G, lim, edges, k, device="cuda"):
    """G: api.Bigraph after the finish; lim / edges: flat numpy walks. Returns (cumulative length, matched-dummy k-mers in tigs)."""
    n_orig, E, V = G.original_edge_count(), G.edge_count(), G.node_count()
    t_edges = torch.from_numpy(edges.astype(np.int64)).to(device)
    t_lim = torch.from_numpy(lim.astype(np.int64)).to(device)
    is_orig = t_edges < n_orig
    orig = t_edges[is_orig]
    assert orig.numel() == n_orig // 2
    seen = torch.zeros(n_orig // 2, dtype=torch.uint8, device=device)
    seen[orig >> 1] = 1
    assert bool(seen.all()), "a unitig is missing from the tigs"
    del seen, orig
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), t_lim[:-1]])
    assert bool((t_lim > starts).all())
    assert bool(is_orig[starts].all()) and bool(is_orig[t_lim - 1].all()), "a tig starts or ends with a dummy edge"
    # degrees and weights, streamed over the edge arrays
    outd = torch.zeros(V, dtype=torch.int32, device=device)
    ind = torch.zeros(V, dtype=torch.int32, device=device)
    weight = torch.empty(E, dtype=torch.int16 if k < 32768 else torch.int32, device=device)
    e_from = torch.empty(E, dtype=torch.int32, device=device)  # (bit patterns of the u32 node ids)
    e_to = torch.empty(E, dtype=torch.int32, device=device)
    unitig_kmers = 0
    for lo in range(0, E, CHUNK):
        n = min(CHUNK, E - lo)
        ex = G.export_range(lo, n, ("edge_from", "edge_to", "edge_weight"))
        one = torch.ones(n, dtype=torch.int32, device=device)
        outd.index_add_(0, torch.from_numpy(ex["edge_from"].astype(np.int64)).to(device), one)
        ind.index_add_(0, torch.from_numpy(ex["edge_to"].astype(np.int64)).to(device), one)
        e_from[lo:lo + n] = torch.from_numpy(ex["edge_from"].view(np.int32)).to(device)
        e_to[lo:lo + n] = torch.from_numpy(ex["edge_to"].view(np.int32)).to(device)
        w = torch.from_numpy(ex["edge_weight"].astype(np.int64)).to(device)
        if lo < n_orig:
            m = min(n, n_orig - lo)
            unitig_kmers += int(w[:m][0::2].sum()) if lo % 2 == 0 else int(w[:m][1::2].sum())
        weight[lo:lo + n] = w.to(weight.dtype)
        del ex, one, w
    mirror = torch.from_numpy(G.export_mirror().astype(np.int64)).to(device)
    sm = mirror == torch.arange(V, device=device)
    assert bool((outd[~sm] == ind[~sm]).all()) and bool((outd[sm] % 2 == 0).all()), "not Eulerian after Eulerisation"
    assert bool((outd == ind[mirror]).all())
    del outd, ind, mirror, sm
    # consecutive edges of a tig are adjacent: the head of one is the tail of the next (a tig is a walk)
    inner = torch.ones(t_edges.numel(), dtype=torch.bool, device=device)
    inner[t_lim - 1] = False  # positions whose successor belongs to the next tig
    pos = torch.nonzero(inner).flatten()
    for lo in range(0, pos.numel(), CHUNK):
        p = pos[lo:lo + CHUNK]
        assert bool((e_to[t_edges[p]] == e_from[t_edges[p + 1]]).all()), "consecutive edges of a tig are not adjacent"
    del e_from, e_to, inner, pos
    w_tig = weight[t_edges].to(torch.int64)
    w_dummy = w_tig[~is_orig]
    if w_dummy.numel():
        assert int(w_dummy.min()) >= 1 and int(w_dummy.max()) <= k - 1, "a breaking edge survived the cut"
    dummy_kmers = int(w_dummy.sum())
    cum = int(w_tig.sum()) + (k - 1) * len(lim)
    assert cum == unitig_kmers + dummy_kmers + (k - 1) * len(lim), "cumulative-length identity"
    return cum, dummy_kmers
