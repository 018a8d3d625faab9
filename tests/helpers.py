"""Shared helpers for the test-suite: building the three implementations' graphs from one description."""
from __future__ import annotations

import numpy as np

import oracle_lib
import pyref


def unitigs_to_arrays(mirror, unitigs):
    mirror = np.asarray(mirror, dtype=np.uint32)
    n = len(unitigs)
    frm = np.zeros(2 * n, np.uint32)
    to = np.zeros(2 * n, np.uint32)
    w = np.zeros(2 * n, np.uint64)
    for u, (a, b, wt) in enumerate(unitigs):
        frm[2 * u], to[2 * u] = a, b
        frm[2 * u + 1], to[2 * u + 1] = mirror[b], mirror[a]
        w[2 * u] = w[2 * u + 1] = wt
    return mirror, frm, to, w


def oracle_graph(mirror, frm, to, w):
    return oracle_lib.OracleGraph.from_arrays(mirror, frm, to, w)


def py_graph(mirror, frm, to, w):
    g = pyref.PyBigraph(len(mirror), [int(x) for x in mirror])
    for e in range(len(frm)):
        g.add_edge(int(frm[e]), int(to[e]), int(w[e]), 0, e // 2, e % 2 == 0)
    return g


def product_graph(mirror, frm, to, w):
    from matchtigs_amd import api

    return api.Bigraph.from_edges(mirror, frm, to, w)


def cumulative_length(tigs, weights, k):
    """Sum over tigs of (k-1) + sum of edge weights (original k-mers + kept dummy weights), SURVEY 8a."""
    return sum((k - 1) + sum(int(weights[e]) for e in t) for t in tigs)


def product_pairs_from_oracle_lists(G, og, k):
    """Runs the PRODUCT's host replay on candidate lists produced by the oracle (CPU-only test path)."""
    on, off, keys, _ = og.candidate_lists(k)
    _, live, mult, _, _ = og.classify()
    pr = G.replay_claims(on, mult.astype(np.int32), live, off[:-1], np.diff(off).astype(np.uint32), keys)
    return pr


def links_of_bigraph(mirror, unitigs):
    """clib.rs input form of an abstract bigraph: one link per (unitig end, unitig start) meeting at the same node."""
    ends, starts = {}, {}
    for u, (a, b, _) in enumerate(unitigs):
        starts.setdefault(a, []).append((u, True))
        ends.setdefault(b, []).append((u, True))
        starts.setdefault(mirror[b], []).append((u, False))   # the backwards strand runs mirror(b) -> mirror(a)
        ends.setdefault(mirror[a], []).append((u, False))
    links = []
    for n in sorted(ends):
        for (ua, sa) in ends[n]:
            for (ub, sb) in starts.get(n, []):
                links.append((ua, sa, ub, sb))
    return links
