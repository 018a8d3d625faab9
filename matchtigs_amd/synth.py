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


def g_csr_weight_thresholds(k: int, mean_weight: float) -> np.ndarray:
    """Integer form of g_csr's weight rule: w = 1 + #{j : x <= T[j]}, x = (bits >> 11) + 1 in [1, 2^53], T descending, len k-1.
    T[j-1] = the largest x whose numpy weight is still >= j + 1 (found by bisection on the very expression g_csr evaluates, which
    is monotone in x), so the device generator reproduces numpy's floating-point result exactly."""
    p = 1.0 / float(mean_weight)
    if p >= 1.0 or k <= 1:
        return np.zeros(0, dtype=np.uint64)
    log1mp = np.log1p(-p)

    def weight_of(x: int) -> int:
        uni = (np.float64(x - 1) + 1.0) * (1.0 / 9007199254740992.0)
        return int(min(max(1 + int(np.floor(np.log(uni) / log1mp)), 1), k))

    out = []
    for j in range(1, k):
        lo, hi = 0, 1 << 53  # weight_of(lo) >= j + 1 (lo = 0: "none"), weight_of(hi) = 1 < j + 1
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if weight_of(mid) >= j + 1:
                lo = mid
            else:
                hi = mid
        out.append(lo)
    return np.array(out, dtype=np.uint64)


def g_csr_device(n_binodes: int, seed: int = 1, k: int = 31, mean_out_degree: float = 1.5, mean_weight: float = 8.0,
                 self_mirror_frac: float = 0.001, max_degree: int = 4, device_id: int = 0):
    """The same graph as ``g_csr`` generated on the GPU (csrc/synth_device.hip), returned as an ``api.Bigraph`` handle: seconds
    instead of minutes at the 2^30 / 2^31 sizes. Needs a GPU; the numpy form above is its twin for CPU tests."""
    import ctypes as C

    from . import _lib, api

    n_sm = int(round(self_mirror_frac * n_binodes))
    v = 2 * n_binodes + n_sm
    u = int(round(mean_out_degree * v / 2.0))
    thr = g_csr_weight_thresholds(k, mean_weight)
    L = _lib.load()
    h = L.mtg_synth_g_csr(n_binodes, n_sm, u, seed & 0xFFFFFFFFFFFFFFFF, k, thr.ctypes.data_as(C.c_void_p) if len(thr) else None,
                          len(thr), max_degree, device_id)
    return api.Bigraph(h)


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


# --------------------------------------------------------------------------------------
# G-seq at scale: the same construction as g_seq, vectorised (2-bit packed k-mers in uint64, k <= 31)
# --------------------------------------------------------------------------------------
@dataclass
class UnitigArrays:
    """Node-centric compacted dBG in the clib.rs input form, as flat arrays (same content as UnitigGraph)."""

    k: int
    seq: np.ndarray      # uint8 ASCII, concatenated forward sequences
    off: np.ndarray      # uint64 [U + 1]
    links: np.ndarray    # int64 [n_links, 4]: (unitig_a, strand_a, unitig_b, strand_b), in g_seq's link order
    kmers: np.ndarray    # sorted canonical k-mer codes (2 bits per base, A < C < G < T)

    @property
    def n_unitigs(self) -> int:
        return int(len(self.off) - 1)

    @property
    def weights(self) -> np.ndarray:
        return (np.diff(self.off.astype(np.int64)) + 1 - self.k).astype(np.uint64)

    def unitig_list(self) -> list[str]:
        s = self.seq.tobytes().decode()
        o = self.off.astype(np.int64)
        return [s[o[i]:o[i + 1]] for i in range(self.n_unitigs)]

    def bcalm2_text(self) -> bytes:
        """BCALM2-style unitig FASTA (`>id LN:i:.. L:<s>:<id>:<s>`), the `--bcalm-in` input of the reference."""
        o = self.off.astype(np.int64)
        lk = self.links
        order = np.argsort(lk[:, 0], kind="stable") if len(lk) else np.zeros(0, np.int64)
        first = np.searchsorted(lk[order, 0], np.arange(self.n_unitigs + 1)) if len(lk) else np.zeros(self.n_unitigs + 1, np.int64)
        s = self.seq.tobytes().decode()
        out = []
        for u in range(self.n_unitigs):
            tags = "".join(f" L:{'+' if lk[j, 1] else '-'}:{int(lk[j, 2])}:{'+' if lk[j, 3] else '-'}" for j in order[first[u]:first[u + 1]])
            out.append(f">{u} LN:i:{o[u + 1] - o[u]}{tags}\n{s[o[u]:o[u + 1]]}\n")
        return "".join(out).encode()


def _kmer_codes(b: np.ndarray, k: int):
    """forward and reverse-complement codes of every k-mer of the base array b (values 0..3)."""
    n = len(b) - k + 1
    fwd = np.zeros(n, np.uint64)
    rc = np.zeros(n, np.uint64)
    b64 = b.astype(np.uint64)
    for j in range(k):
        x = b64[j:j + n]
        fwd |= x << np.uint64(2 * (k - 1 - j))
        rc |= (np.uint64(3) - x) << np.uint64(2 * j)
    return fwd, rc


def kmer_codes_of_sequences(seq: np.ndarray, off: np.ndarray, k: int) -> np.ndarray:
    """Sorted distinct canonical k-mer codes of ASCII sequences (concatenated `seq`, offsets `off`)."""
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    b = lut[seq]
    if (b == 255).any():
        raise ValueError("non-ACGT character")
    fwd, rc = _kmer_codes(b, k)
    o = off.astype(np.int64)
    lens = np.diff(o)
    valid = np.zeros(len(fwd) + 1, np.int64)      # k-mers that do not straddle a sequence boundary
    np.add.at(valid, o[:-1], 1)
    np.add.at(valid, np.maximum(o[1:] - k + 1, o[:-1]), -1)
    ok = np.cumsum(valid[: len(fwd)]) > 0
    ok &= np.repeat(lens >= k, lens)[: len(fwd)] if len(lens) else ok
    return np.unique(np.minimum(fwd, rc)[ok])


def g_seq_arrays(length: int, seed: int = 1, k: int = 31, haplotypes: int = 4, sub_rate: float = 0.02) -> UnitigArrays:
    """g_seq for large genomes: identical unitigs, unitig order, orientations and link order (checked against g_seq in the
    tests), built with array operations. Raises on the degenerate shapes g_seq handles by its visit order (a unitig that
    is a cycle or runs into its own reverse complement); they do not occur in random genomes of useful k."""
    if k > 31 or k % 2 == 0:
        raise ValueError("k must be odd and <= 31")
    g = (splitmix64(seed, length, 10) % np.uint64(4)).astype(np.int64)
    canon_all = []
    for h in range(haplotypes):
        if h == 0:
            gh = g
        else:
            mut = _uniform01(splitmix64(seed, length, 20 + h)) < sub_rate
            shift = (splitmix64(seed, length, 40 + h) % np.uint64(3)).astype(np.int64) + 1
            gh = np.where(mut, (g + shift) % 4, g)
        fwd, rc = _kmer_codes(gh.astype(np.uint8), k)
        canon_all.append(np.minimum(fwd, rc))
        del fwd, rc
    K = np.unique(np.concatenate(canon_all))
    del canon_all
    N = len(K)
    mask = np.uint64((1 << (2 * k)) - 1)

    def revcomp_code(c):  # reverse the 2-bit groups of the complemented word, drop the 64 - 2k pad bits
        x = ~c
        for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
            mm = np.uint64(m)
            x = ((x >> np.uint64(sh)) & mm) | ((x & mm) << np.uint64(sh))
        x = (x >> np.uint64(32)) | (x << np.uint64(32))
        return x >> np.uint64(64 - 2 * k)

    code = np.empty(2 * N, np.uint64)        # oriented k-mer x = 2 * index + (0 canonical orientation, 1 reverse complement)
    code[0::2] = K
    code[1::2] = revcomp_code(K)
    # successors by a merge join: x -> y iff the (k-1)-suffix of x is the (k-1)-prefix of y
    s_order = np.argsort(code)            # oriented k-mers by code = by (prefix, last base)
    S = code[s_order]
    pref = S >> np.uint64(2)
    first = np.r_[True, pref[1:] != pref[:-1]]
    g_start = np.nonzero(first)[0]
    g_pref = pref[g_start]
    g_end = np.r_[g_start[1:], len(S)]
    suf = code & np.uint64((1 << (2 * (k - 1))) - 1)
    a_order = np.argsort(suf)
    pos = np.searchsorted(g_pref, suf[a_order])           # ascending needles
    pos_c = np.minimum(pos, len(g_pref) - 1)
    hit = g_pref[pos_c] == suf[a_order]
    succ = np.full((2 * N, 4), -1, np.int64)              # oriented successor by appended base
    xs = a_order[hit]
    gs = pos_c[hit]
    for t in range(4):
        idx = g_start[gs] + t
        ok = idx < g_end[gs]
        ii = idx[ok]
        succ[xs[ok], (S[ii] & np.uint64(3)).astype(np.int64)] = s_order[ii]
    del S, pref, first, suf, a_order, pos, pos_c, hit, xs, gs
    out_deg = (succ >= 0).sum(axis=1)
    in_deg = out_deg.reshape(N, 2)[:, ::-1].reshape(2 * N)   # predecessors of x = reverse complements of the successors of x ^ 1
    only = np.where(out_deg == 1, succ.max(axis=1), -1)
    internal = (only >= 0) & (in_deg[np.maximum(only, 0)] == 1)
    nxt = np.where(internal, only, -1)
    prv = np.full(2 * N, -1, np.int64)
    prv[nxt[internal]] = np.nonzero(internal)[0]
    if (nxt == (np.arange(2 * N) ^ 1)).any():
        raise NotImplementedError("a unitig runs into its own reverse complement")
    # head, position and the smallest member of every oriented path, by pointer jumping over prv
    jump = prv.copy()
    rank = (prv >= 0).astype(np.int64)      # invariant: rank[x] = steps from x back to jump[x]
    mn = np.arange(2 * N)                   # invariant: mn[x] = smallest member of the path segment (jump[x], x]
    for _ in range(64):
        live = np.nonzero(jump >= 0)[0]
        j = jump[live]
        jj = jump[j]
        upd = jj >= 0
        if not upd.any():
            break
        idx = live[upd]
        rank[idx] += rank[j[upd]]
        mn[idx] = np.minimum(mn[idx], mn[j[upd]])
        jump[idx] = jj[upd]
    else:
        raise NotImplementedError("a unitig is a cycle")
    head = np.where(jump >= 0, jump, np.arange(2 * N))
    is_tail = nxt < 0
    minx = np.full(2 * N, 2 * N, np.int64)  # per head: smallest member of its path
    minx[head[is_tail]] = np.minimum(mn[is_tail], head[is_tail])

    # orientation and order as g_seq: it visits canonical k-mers ascending and spells the unitig of the first unused one in
    # that k-mer's canonical orientation, so a unitig = the oriented path whose smallest member x = 2 * index + orientation is even
    is_head = head == np.arange(2 * N)
    sel_heads = np.nonzero(is_head & (minx % 2 == 0) & (minx < 2 * N))[0]
    sel_heads = sel_heads[np.argsort(minx[sel_heads], kind="stable")]
    U = len(sel_heads)
    uid_of_head = np.full(2 * N, -1, np.int64)
    uid_of_head[sel_heads] = np.arange(U)
    uid = uid_of_head[head]                    # -1 for members of the reverse-complement paths
    members = np.nonzero(uid >= 0)[0]
    if len(members) != N:
        raise NotImplementedError("a unitig runs into its own reverse complement")
    m = np.bincount(uid[members], minlength=U)                 # k-mers per unitig
    off = np.zeros(U + 1, np.int64)
    off[1:] = np.cumsum(m + k - 1)
    seq = np.zeros(int(off[-1]), np.uint8)
    seq[off[uid[members]] + k - 1 + rank[members]] = (code[members] & np.uint64(3)).astype(np.uint8)
    hc = code[sel_heads]
    for t in range(k - 1):
        seq[off[:-1] + t] = ((hc >> np.uint64(2 * (k - 1 - t))) & np.uint64(3)).astype(np.uint8)
    ascii_seq = np.frombuffer(b"ACGT", np.uint8)[seq]

    # links in g_seq's order: for unitig i ascending, strand True then False: every oriented unitig (j, sb) whose first k-mer
    # follows the last k-mer of (i, sa), ordered by (j, True before False)
    tails = np.empty(U, np.int64)
    last = rank[members] == m[uid[members]] - 1
    tails[uid[members[last]]] = members[last]
    start_of = np.full(2 * N, -1, np.int64)
    start_of[sel_heads] = 2 * np.arange(U)
    start_of[tails ^ 1] = 2 * np.arange(U) + 1
    ends = np.empty(2 * U, np.int64)
    ends[0::2] = tails
    ends[1::2] = sel_heads ^ 1
    sy = succ[ends]                                              # [2U, 4]
    so = np.where(sy >= 0, start_of[np.maximum(sy, 0)], -1)
    if ((sy >= 0) & (so < 0)).any():
        raise NotImplementedError("a unitig end is followed by the inside of a unitig (cycle / hairpin shapes)")
    big = np.iinfo(np.int64).max
    so_sorted = np.sort(np.where(so >= 0, so, big), axis=1)
    valid = so_sorted != big
    e_idx = np.repeat(np.arange(2 * U), 4).reshape(2 * U, 4)[valid]
    tgt = so_sorted[valid]
    links = np.stack([e_idx >> 1, (e_idx & 1) == 0, tgt >> 1, (tgt & 1) == 0], axis=1).astype(np.int64)
    return UnitigArrays(k, ascii_seq, off.astype(np.uint64), links, K)


def kmer_codes_of_sequences_torch(seq: np.ndarray, off: np.ndarray, k: int, device: str = "cuda"):
    """``kmer_codes_of_sequences`` with torch as the calculator: (sorted distinct canonical k-mer codes as a uint64 numpy array, number
    of k-mer occurrences they were found in) -- the second number equals the first array's length iff no k-mer is spelled twice."""
    import torch

    dev = torch.device(device)
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    b = torch.from_numpy(lut[seq]).to(dev)
    if bool((b == 255).any()):
        raise ValueError("non-ACGT character")
    b = b.to(torch.int64)
    n = b.numel() - k + 1
    fwd = torch.zeros(n, dtype=torch.int64, device=dev)
    rc = torch.zeros(n, dtype=torch.int64, device=dev)
    for j in range(k):
        x = b[j:j + n]
        fwd |= x << (2 * (k - 1 - j))
        rc |= (3 - x) << (2 * j)
    o = torch.from_numpy(off.astype(np.int64)).to(dev)
    lens = o[1:] - o[:-1]
    # k-mers that start inside [o_i, o_i+1 - k] of a sequence with at least k characters
    start_of = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)[:n]
    ok = (torch.arange(n, device=dev) + k) <= o[1:][start_of]
    canon = torch.minimum(fwd, rc)[ok]
    return torch.unique(canon).cpu().numpy().view(np.uint64), int(canon.numel())


def g_seq_arrays_torch(length: int, seed: int = 1, k: int = 31, haplotypes: int = 4, sub_rate: float = 0.02, device: str = "cuda") -> UnitigArrays:
    """``g_seq_arrays`` with torch as the calculator (on the GPU: seconds instead of minutes at the C. elegans-like 10^8 and the chr1-like
    2.5 * 10^8 of SURVEY 8d): the same k-mer set, unitigs, unitig order, orientations and link order -- held equal to ``g_seq_arrays``
    in the tests. Test / bench infrastructure (a workload generator), not part of the path. 64-bit codes live in int64 tensors: k-mer
    codes have at most 62 bits, so their order is the unsigned one; the generator's unsigned arithmetic (splitmix64, the
    reverse-complement bit tricks) is spelled with wrapping int64 operations and logical shifts."""
    import torch

    if k > 31 or k % 2 == 0:
        raise ValueError("k must be odd and <= 31")
    dev = torch.device(device)
    i64 = torch.int64

    def s64(c: int) -> int:  # a 64-bit pattern as the signed value torch holds
        c &= 0xFFFFFFFFFFFFFFFF
        return c - (1 << 64) if c >= (1 << 63) else c

    def lsr(x, sh: int):  # logical shift right of the 64-bit pattern
        return (x >> sh) & ((1 << (64 - sh)) - 1) if sh else x

    def splitmix(n: int, stream: int):
        z = torch.arange(1, n + 1, dtype=i64, device=dev) * s64(0x9E3779B97F4A7C15) + s64(seed + (stream << 40))
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)

    def kmer_codes(b):  # b: int64 bases 0..3
        n = b.numel() - k + 1
        fwd = torch.zeros(n, dtype=i64, device=dev)
        rc = torch.zeros(n, dtype=i64, device=dev)
        for j in range(k):
            x = b[j:j + n]
            fwd |= x << (2 * (k - 1 - j))
            rc |= (3 - x) << (2 * j)
        return fwd, rc

    g = splitmix(length, 10) & 3
    canon_all = []
    for h in range(haplotypes):
        if h == 0:
            gh = g
        else:
            u01 = (lsr(splitmix(length, 20 + h), 11).to(torch.float64) + 1.0) * (1.0 / 9007199254740992.0)
            z = splitmix(length, 40 + h)
            shift = torch.remainder(torch.remainder(z, 3) + (z < 0).to(i64), 3) + 1  # the unsigned value mod 3 (2^64 = 1 mod 3)
            gh = torch.where(u01 < sub_rate, torch.remainder(g + shift, 4), g)
            del u01, z, shift
        fwd, rc = kmer_codes(gh)
        canon_all.append(torch.minimum(fwd, rc))
        del fwd, rc
    K = torch.unique(torch.cat(canon_all))
    del canon_all
    N = K.numel()

    def revcomp_code(c):
        x = ~c
        for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
            x = (lsr(x, sh) & m) | ((x & m) << sh)
        x = lsr(x, 32) | (x << 32)
        return lsr(x, 64 - 2 * k)

    code = torch.empty(2 * N, dtype=i64, device=dev)
    code[0::2] = K
    code[1::2] = revcomp_code(K)
    ar2n = torch.arange(2 * N, device=dev)
    s_order = torch.argsort(code)
    S = code[s_order]
    pref = S >> 2
    first = torch.ones(2 * N, dtype=torch.bool, device=dev)
    first[1:] = pref[1:] != pref[:-1]
    g_start = torch.nonzero(first).flatten()
    g_pref = pref[g_start]
    g_end = torch.cat([g_start[1:], torch.tensor([2 * N], device=dev)])
    suf = code & ((1 << (2 * (k - 1))) - 1)
    pos_c = torch.clamp(torch.searchsorted(g_pref, suf), max=g_pref.numel() - 1)
    hit = g_pref[pos_c] == suf
    succ = torch.full((2 * N, 4), -1, dtype=i64, device=dev)
    xs = ar2n[hit]
    gs = pos_c[hit]
    for t in range(4):
        idx = g_start[gs] + t
        ok = idx < g_end[gs]
        ii = idx[ok]
        succ[xs[ok], S[ii] & 3] = s_order[ii]
    del S, pref, first, suf, pos_c, hit, xs, gs
    out_deg = (succ >= 0).sum(dim=1)
    in_deg = out_deg.reshape(N, 2).flip(1).reshape(2 * N)
    only = torch.where(out_deg == 1, succ.max(dim=1).values, torch.full_like(out_deg, -1))
    internal = (only >= 0) & (in_deg[torch.clamp(only, min=0)] == 1)
    nxt = torch.where(internal, only, torch.full_like(only, -1))
    prv = torch.full((2 * N,), -1, dtype=i64, device=dev)
    prv[nxt[internal]] = ar2n[internal]
    if bool((nxt == (ar2n ^ 1)).any()):
        raise NotImplementedError("a unitig runs into its own reverse complement")
    jump = prv.clone()
    rank = (prv >= 0).to(i64)
    mn = ar2n.clone()
    for _ in range(64):
        live = torch.nonzero(jump >= 0).flatten()
        j = jump[live]
        jj = jump[j]
        upd = jj >= 0
        if not bool(upd.any()):
            break
        idx = live[upd]
        rank[idx] += rank[j[upd]]
        mn[idx] = torch.minimum(mn[idx], mn[j[upd]])
        jump[idx] = jj[upd]
    else:
        raise NotImplementedError("a unitig is a cycle")
    head = torch.where(jump >= 0, jump, ar2n)
    is_tail = nxt < 0
    minx = torch.full((2 * N,), 2 * N, dtype=i64, device=dev)
    minx[head[is_tail]] = torch.minimum(mn[is_tail], head[is_tail])
    is_head = head == ar2n
    sel_heads = torch.nonzero(is_head & (minx % 2 == 0) & (minx < 2 * N)).flatten()
    sel_heads = sel_heads[torch.argsort(minx[sel_heads])]  # (the keys are distinct)
    U = sel_heads.numel()
    uid_of_head = torch.full((2 * N,), -1, dtype=i64, device=dev)
    uid_of_head[sel_heads] = torch.arange(U, device=dev)
    uid = uid_of_head[head]
    members = torch.nonzero(uid >= 0).flatten()
    if members.numel() != N:
        raise NotImplementedError("a unitig runs into its own reverse complement")
    m = torch.bincount(uid[members], minlength=U)
    off = torch.zeros(U + 1, dtype=i64, device=dev)
    off[1:] = torch.cumsum(m + k - 1, 0)
    seq = torch.zeros(int(off[-1]), dtype=torch.uint8, device=dev)
    seq[off[uid[members]] + k - 1 + rank[members]] = (code[members] & 3).to(torch.uint8)
    hc = code[sel_heads]
    for t in range(k - 1):
        seq[off[:-1] + t] = ((hc >> (2 * (k - 1 - t))) & 3).to(torch.uint8)
    ascii_seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[seq.to(i64)]
    tails = torch.empty(U, dtype=i64, device=dev)
    last = rank[members] == m[uid[members]] - 1
    tails[uid[members[last]]] = members[last]
    start_of = torch.full((2 * N,), -1, dtype=i64, device=dev)
    start_of[sel_heads] = 2 * torch.arange(U, device=dev)
    start_of[tails ^ 1] = 2 * torch.arange(U, device=dev) + 1
    ends = torch.empty(2 * U, dtype=i64, device=dev)
    ends[0::2] = tails
    ends[1::2] = sel_heads ^ 1
    sy = succ[ends]
    so = torch.where(sy >= 0, start_of[torch.clamp(sy, min=0)], torch.full_like(sy, -1))
    if bool(((sy >= 0) & (so < 0)).any()):
        raise NotImplementedError("a unitig end is followed by the inside of a unitig (cycle / hairpin shapes)")
    big = (1 << 63) - 1
    so_sorted = torch.sort(torch.where(so >= 0, so, torch.full_like(so, big)), dim=1).values
    valid = so_sorted != big
    e_idx = torch.arange(2 * U, device=dev).repeat_interleave(4).reshape(2 * U, 4)[valid]
    tgt = so_sorted[valid]
    links = torch.stack([e_idx >> 1, ((e_idx & 1) == 0).to(i64), tgt >> 1, ((tgt & 1) == 0).to(i64)], dim=1)
    return UnitigArrays(k, ascii_seq.cpu().numpy(), off.cpu().numpy().astype(np.uint64), links.cpu().numpy().astype(np.int64),
                        K.cpu().numpy().view(np.uint64))


def unitig_graph_of_arrays(ua: UnitigArrays) -> UnitigGraph:
    """The same graph in g_seq's object form (small sizes: builds Python strings and a set)."""
    bases = "ACGT"

    def decode(c):
        return "".join(bases[(int(c) >> (2 * (ua.k - 1 - j))) & 3] for j in range(ua.k))

    return UnitigGraph(ua.k, ua.unitig_list(), [(int(a), bool(b), int(c), bool(d)) for a, b, c, d in ua.links],
                       {decode(c) for c in ua.kmers})


def dbg_like_links(U: int, seed: int = 1) -> np.ndarray:
    """Links [n, 4] (unitig a, strand a, unitig b, strand b) of a unitig graph with the degree structure of a compacted de Bruijn
    graph: 1.4 node sides per unitig, every end arriving at a side linked with every end leaving it, listed unitig by unitig the way
    BCALM2 does -- input for the clib.rs builder (clib.rs:94-259) at sizes no sequence generator reaches quickly."""
    rng = np.random.default_rng(seed)
    M = int(1.4 * U) & ~1  # node sides; the mirror of side x is x ^ 1
    s = rng.integers(0, M, U)  # where the forward unitig starts
    t = rng.integers(0, M, U)  # where it ends
    u = np.arange(U, dtype=np.int64)
    arr_node = np.concatenate([t, s ^ 1])  # ends arriving at a side: (u, fwd) at t, (u, bwd) at s^1
    arr_id = np.concatenate([u << 1 | 1, u << 1])
    lea_node = np.concatenate([s, t ^ 1])  # ends leaving a side
    lea_id = np.concatenate([u << 1 | 1, u << 1])
    oa = np.argsort(arr_node, kind="stable")
    ol = np.argsort(lea_node, kind="stable")
    arr_node, arr_id = arr_node[oa], arr_id[oa]
    lea_node, lea_id = lea_node[ol], lea_id[ol]
    ca = np.bincount(arr_node, minlength=M)
    cl = np.bincount(lea_node, minlength=M)
    start_l = np.concatenate([[0], np.cumsum(cl)[:-1]])
    # every arrival at side x pairs with every leaving end of x
    rep = cl[arr_node]
    a_rep = np.repeat(arr_id, rep)
    base = np.repeat(start_l[arr_node], rep)
    off = np.arange(len(a_rep)) - np.repeat(np.concatenate([[0], np.cumsum(rep)[:-1]]), rep)
    b_rep = lea_id[base + off]
    links = np.stack([a_rep >> 1, a_rep & 1, b_rep >> 1, b_rep & 1], axis=1).astype(np.int64)
    # BCALM2 lists links unitig by unitig: order by the first unitig
    return np.ascontiguousarray(links[np.argsort(links[:, 0], kind="stable")])
