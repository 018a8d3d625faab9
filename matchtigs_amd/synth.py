"""Seeded synthetic inputs for the greedy-matchtigs hot path.

Two generators, both driven by a counter-based splitmix64 stream so that the same
(seed, parameters) gives the same graph everywhere (container, GPU box, any numpy):

* ``g_csr``  -- direct random edge-centric bigraph for scale ("G-csr" in SURVEY 8d).
* ``g_seq``  -- random genome with haplotype copies -> k-mers -> unitigs + links
               (small sizes only; pure numpy/Python) for end-to-end spelling checks.

The graph model is the reference's: nodes in mirror pairs (plus a few self-mirror nodes),
unitig ``u`` = directed edge ``2u`` (from, to, forwards) and its mirror ``2u+1``
(mirror(to), mirror(from), backwards), both with weight = number of k-mers (clib.rs:236-248).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit values: value i = mix(seed + stream*2^40 + (i+1)*golden). Counter-based, vectorised."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed + (stream << 40)) & 0xFFFFFFFFFFFFFFFF)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(bits: np.ndarray) -> np.ndarray:
    """(0, 1] doubles from 64-bit values."""
    return ((bits >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


@dataclass
class Bigraph:
    """Flat edge-centric bigraph: original edges only, edge 2u / 2u+1 = unitig u forward / mirror."""

    mirror: np.ndarray       # uint32 [V]
    edge_from: np.ndarray    # uint32 [E]
    edge_to: np.ndarray      # uint32 [E]
    edge_weight: np.ndarray  # uint64 [E]
    k: int

    @property
    def n_nodes(self) -> int:
        return int(self.mirror.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.edge_from.shape[0])

    def describe(self) -> dict:
        return {"V": self.n_nodes, "E": self.n_edges, "k": self.k}


def g_csr(n_binodes: int, seed: int = 1, k: int = 31, mean_out_degree: float = 1.5,
          mean_weight: float = 8.0, self_mirror_frac: float = 0.001, max_degree: int = 4) -> Bigraph:
    """G-csr(Nb, d, w, seed): Nb mirror pairs + round(self_mirror_frac*Nb) self-mirror nodes.

    Each unitig picks (from, to) uniformly among all nodes; unitigs that would push any node's
    out-degree above ``max_degree`` (which, by the mirror property, also bounds the in-degree)
    are dropped in unitig order. weight = Geom(1/mean_weight) >= 1, clamped to k.
    """
    n_sm = int(round(self_mirror_frac * n_binodes))
    v = 2 * n_binodes + n_sm
    if v >= 0xFFFFFFFF:
        raise ValueError("too many nodes for u32 ids")
    mirror = np.empty(v, dtype=np.uint32)
    pair = np.arange(2 * n_binodes, dtype=np.uint32)
    mirror[: 2 * n_binodes] = pair ^ np.uint32(1)
    mirror[2 * n_binodes:] = np.arange(2 * n_binodes, v, dtype=np.uint32)

    u = int(round(mean_out_degree * v / 2.0))
    a = (splitmix64(seed, u, 1) % np.uint64(v)).astype(np.uint32)
    b = (splitmix64(seed, u, 2) % np.uint64(v)).astype(np.uint32)
    uni = _uniform01(splitmix64(seed, u, 3))
    p = 1.0 / float(mean_weight)
    w = 1 + np.floor(np.log(uni) / np.log1p(-p)).astype(np.int64) if p < 1.0 else np.ones(u, dtype=np.int64)
    w = np.clip(w, 1, k).astype(np.uint64)

    # directed edges in insertion order: 2u = a->b, 2u+1 = m(b)->m(a)
    frm = np.empty(2 * u, dtype=np.uint32)
    to = np.empty(2 * u, dtype=np.uint32)
    frm[0::2], to[0::2] = a, b
    frm[1::2], to[1::2] = mirror[b], mirror[a]
    # rank of each directed edge among edges leaving the same node (stable, insertion order)
    order = np.argsort(frm, kind="stable")
    sorted_from = frm[order]
    first = np.r_[True, sorted_from[1:] != sorted_from[:-1]]
    start_idx = np.maximum.accumulate(np.where(first, np.arange(2 * u), 0))
    rank_sorted = np.arange(2 * u) - start_idx
    rank = np.empty(2 * u, dtype=np.int64)
    rank[order] = rank_sorted
    keep_u = (rank[0::2] < max_degree) & (rank[1::2] < max_degree)
    keep = np.repeat(keep_u, 2)
    return Bigraph(mirror, frm[keep].copy(), to[keep].copy(), np.repeat(w[keep_u], 2), k)


# --------------------------------------------------------------------------------------
# G-seq: tiny real de Bruijn graphs (unitigs + links), for spelling / k-mer-set tests
# --------------------------------------------------------------------------------------
_COMP = str.maketrans("ACGT", "TGCA")


def revcomp(s: str) -> str:
    return s.translate(_COMP)[::-1]


def canonical(s: str) -> str:
    r = revcomp(s)
    return s if s <= r else r


def random_genome(length: int, seed: int, haplotypes: int = 4, sub_rate: float = 0.02) -> list[str]:
    bases = np.array(list("ACGT"))
    g = (splitmix64(seed, length, 10) % np.uint64(4)).astype(np.int64)
    out = ["".join(bases[g])]
    for h in range(1, haplotypes):
        mut = _uniform01(splitmix64(seed, length, 20 + h)) < sub_rate
        shift = (splitmix64(seed, length, 40 + h) % np.uint64(3)).astype(np.int64) + 1
        gh = np.where(mut, (g + shift) % 4, g)
        out.append("".join(bases[gh]))
    return out


@dataclass
class UnitigGraph:
    """Node-centric compacted dBG in the clib.rs input form."""

    k: int
    unitigs: list[str]                       # forward sequence of each unitig
    links: list[tuple[int, bool, int, bool]]  # (unitig_a, strand_a, unitig_b, strand_b), as BCALM2 L: lines
    kmers: set[str]                           # canonical k-mers

    @property
    def weights(self) -> np.ndarray:
        return np.array([len(u) + 1 - self.k for u in self.unitigs], dtype=np.uint64)


def g_seq(length: int, seed: int = 1, k: int = 31, haplotypes: int = 4, sub_rate: float = 0.02) -> UnitigGraph:
    """Random genome -> canonical k-mer set -> maximal unitigs + all overlaps between unitig ends."""
    seqs = random_genome(length, seed, haplotypes, sub_rate)
    kmers: set[str] = set()
    for s in seqs:
        for i in range(len(s) - k + 1):
            kmers.add(canonical(s[i:i + k]))

    def succ(x: str) -> list[str]:  # oriented successors present in the set
        return [x[1:] + c for c in "ACGT" if canonical(x[1:] + c) in kmers]

    def pred(x: str) -> list[str]:
        return [c + x[:-1] for c in "ACGT" if canonical(c + x[:-1]) in kmers]

    used: set[str] = set()
    unitigs: list[str] = []
    for km in sorted(kmers):
        if km in used:
            continue
        used.add(km)
        path = [km]
        # extend right
        cur = km
        while True:
            s = succ(cur)
            if len(s) != 1:
                break
            nxt = s[0]
            if len(pred(nxt)) != 1 or canonical(nxt) in used:
                break
            used.add(canonical(nxt))
            path.append(nxt)
            cur = nxt
        # extend left
        cur = km
        left: list[str] = []
        while True:
            p = pred(cur)
            if len(p) != 1:
                break
            prv = p[0]
            if len(succ(prv)) != 1 or canonical(prv) in used:
                break
            used.add(canonical(prv))
            left.append(prv)
            cur = prv
        full = left[::-1] + path
        unitigs.append(full[0] + "".join(x[-1] for x in full[1:]))

    # links: unitig a (strand sa) -> unitig b (strand sb) iff last k-1 of oriented a == first k-1 of oriented b
    starts: dict[str, list[tuple[int, bool]]] = {}
    for i, u in enumerate(unitigs):
        starts.setdefault(u[: k - 1], []).append((i, True))
        starts.setdefault(revcomp(u)[: k - 1], []).append((i, False))
    links = []
    for i, u in enumerate(unitigs):
        for sa in (True, False):
            o = u if sa else revcomp(u)
            last_kmer = o[-k:]
            for (j, sb) in starts.get(o[-(k - 1):], []):
                ob = unitigs[j] if sb else revcomp(unitigs[j])
                # the junction k-mer must exist in the graph (it does iff it is an edge of the node-centric dBG)
                if canonical(last_kmer[1:] + ob[k - 1]) in kmers:
                    links.append((i, sa, j, sb))
    return UnitigGraph(k, unitigs, links, kmers)


def kmer_set_of_tigs(tigs: list[str], k: int) -> set[str]:
    out: set[str] = set()
    for t in tigs:
        for i in range(len(t) - k + 1):
            out.add(canonical(t[i:i + k]))
    return out
