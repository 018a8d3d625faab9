"""Host-side mirror of the reference's operator interface for the greedy-matchtigs / eulertigs path.

Names, argument meaning and error behaviour follow the reference crate (citations into
/root/reference/src/):

* ``TigAlgorithm.compute_tigs(graph, configuration) -> walks``   implementation/mod.rs:49-59
* ``GreedytigAlgorithm`` / ``GreedytigAlgorithmConfiguration``  implementation/greedytigs/mod.rs:33-90
* ``EulertigAlgorithm`` / ``EulertigAlgorithmConfiguration``    implementation/eulertigs/mod.rs:18-45
* ``HeapType`` / ``NodeWeightArrayType`` / ``PerformanceDataType`` (+ ``from_str``) implementation/mod.rs:61-126
* ``MatchtigEdgeData`` view (``weight / is_dummy / is_original / is_forwards / is_backwards / mirror``)
  implementation/mod.rs:287-317
* the C-ABI of src/clib.rs through ``ClibGraph``.

Everything computes through libmatchtigs.so (HIP). PyTorch is only used by callers that want to own
device buffers (bench.py, multi-GPU); this module itself never imports torch.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib


# ---- configuration enums (implementation/mod.rs:61-126) ---------------------------------------
class _FromStr(enum.Enum):
    @classmethod
    def from_str(cls, s: str):
        for m in cls:
            if m.value == s:
                return m
        raise ValueError(f"Unknown {cls._label}: {s}")  # Err(format!("Unknown ...: {other}"))


class NodeWeightArrayType(_FromStr):
    EpochNodeWeightArray = "EpochNodeWeightArray"
    HashbrownHashMap = "HashbrownHashMap"


NodeWeightArrayType._label = "node weight array type"


class HeapType(_FromStr):
    StdBinaryHeap = "StdBinaryHeap"


HeapType._label = "heap type"


class PerformanceDataType(_FromStr):
    None_ = "None"
    Complete = "Complete"


PerformanceDataType._label = "performance data type"


class EulerMode(enum.IntEnum):
    """Engine-only option (mtg_config.euler_mode): HostReferenceOrder = the reference's walk order (bit-exact tigs);
    Device = parallel Euler bicycles on the GPU (valid walks, same #tigs / cumulative length, different order)."""

    HostReferenceOrder = 0
    Device = 1


class FinishStage(enum.IntEnum):
    """Engine-only option (mtg_config.finish_stage): where dummy insertion, the Euleriser and the cutter run. Auto = on the GPU
    whenever one is visible (same edges and tigs as the host stages)."""

    Auto = 0
    Host = 1
    Device = 2


@dataclass
class GreedytigAlgorithmConfiguration:
    """greedytigs/mod.rs:40-73. heap / node-weight-array / staged-parallelism settings select CPU data
    structures in the reference and never change results; the MI355X engine validates and otherwise ignores them.
    ``threads`` likewise: results always equal the reference's 1-thread order (its only deterministic one).
    ``euler_mode`` / ``device_ids`` are engine-only fields (mtg_config)."""

    threads: int
    k: int
    staged_parallelism_divisor: Optional[float] = None
    resource_limit_factor: int = 0
    node_weight_array_type: NodeWeightArrayType = NodeWeightArrayType.HashbrownHashMap
    heap_type: HeapType = HeapType.StdBinaryHeap
    performance_data_type: PerformanceDataType = PerformanceDataType.None_
    euler_mode: EulerMode = EulerMode.HostReferenceOrder
    device_ids: Sequence[int] = (0,)
    finish_stage: FinishStage = FinishStage.Auto

    @classmethod
    def new(cls, threads: int, k: int) -> "GreedytigAlgorithmConfiguration":
        return cls(threads, k)

    def to_c(self) -> "_lib.MtgConfig":
        c = _lib.MtgConfig()
        _lib.load().mtg_config_init(C.byref(c), self.threads, self.k)
        c.staged_parallelism_divisor = float(self.staged_parallelism_divisor or 0.0)
        c.resource_limit_factor = self.resource_limit_factor
        c.node_weight_array_type = list(NodeWeightArrayType).index(self.node_weight_array_type)
        c.heap_type = list(HeapType).index(self.heap_type)
        c.performance_data_type = list(PerformanceDataType).index(self.performance_data_type)
        c.euler_mode = int(self.euler_mode)
        c.finish_stage = int(self.finish_stage)
        c.n_devices = len(self.device_ids)
        for i, dv in enumerate(self.device_ids):
            c.device_ids[i] = int(dv)
        return c


@dataclass
class EulertigAlgorithmConfiguration:
    """eulertigs/mod.rs:42-45 (+ the engine-only euler_mode / device_id)."""

    k: int
    euler_mode: EulerMode = EulerMode.HostReferenceOrder
    device_id: int = 0
    finish_stage: FinishStage = FinishStage.Auto

    def to_c(self) -> "_lib.MtgConfig":
        return GreedytigAlgorithmConfiguration(1, self.k, euler_mode=self.euler_mode, device_ids=(self.device_id,),
                                               finish_stage=self.finish_stage).to_c()


@dataclass
class MatchtigAlgorithmConfiguration:
    """matchtigs/mod.rs:33-45 (+ the engine-only euler_mode / device_id). ``matcher_path`` is the external blossom5-compatible
    executable (`<matcher> -e <instance> -w <solution>`); the instance goes to ``<matching_file_prefix>.minimalperfectmatching``."""

    threads: int
    k: int
    matching_file_prefix: str
    matcher_path: str
    euler_mode: EulerMode = EulerMode.HostReferenceOrder
    device_id: int = 0

    def to_c(self) -> "_lib.MtgConfig":
        c = GreedytigAlgorithmConfiguration(self.threads, self.k, euler_mode=self.euler_mode, device_ids=(self.device_id,)).to_c()
        self._keep = (str(self.matching_file_prefix).encode(), str(self.matcher_path).encode())  # c_char_p fields borrow these
        c.matching_file_prefix, c.matcher_path = self._keep
        return c


# ---- edge payload view (implementation/mod.rs:287-317; clib.rs:45-85) -------------------------
@dataclass(frozen=True)
class MatchtigEdgeData:
    sequence_handle: int  # unitig id; 0 (= Default) for dummy edges
    forwards: bool
    _weight: int
    dummy_edge_id: int

    def weight(self) -> int:
        return self._weight

    def is_dummy(self) -> bool:
        return self.dummy_edge_id != 0

    def is_original(self) -> bool:
        return not self.is_dummy()

    def is_forwards(self) -> bool:
        return self.forwards

    def is_backwards(self) -> bool:
        return not self.forwards

    def mirror(self) -> "MatchtigEdgeData":
        return MatchtigEdgeData(self.sequence_handle, not self.forwards, self._weight, self.dummy_edge_id)

    @classmethod
    def new(cls, sequence_handle: int, forwards: bool, weight: int, dummy_id: int) -> "MatchtigEdgeData":
        return cls(sequence_handle, forwards, weight, dummy_id)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Bigraph:
    """Edge-centric bigraph handle (replaces ``NodeBigraphWrapper<PetGraph<(), EdgeData>>``).

    compute_tigs mutates it (dummy edges are appended) exactly like the reference mutates its graph
    (greedytigs/mod.rs:678-689, implementation/mod.rs:492-493), and the returned walks index the mutated graph.
    """

    def __init__(self, handle: int):
        self._h = handle
        self._L = _lib.load()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.mtg_graph_free(h)

    @classmethod
    def from_edges(cls, mirror, edge_from, edge_to, edge_weight) -> "Bigraph":
        L = _lib.load()
        m = np.ascontiguousarray(mirror, dtype=np.uint32)
        f = np.ascontiguousarray(edge_from, dtype=np.uint32)
        t = np.ascontiguousarray(edge_to, dtype=np.uint32)
        w = np.ascontiguousarray(edge_weight, dtype=np.uint64)
        if not (len(f) == len(t) == len(w)):
            raise ValueError("edge arrays differ in length")
        return cls(L.mtg_graph_from_edges(len(m), _ptr(m), len(f), _ptr(f), _ptr(t), _ptr(w)))

    @classmethod
    def from_unitig_links(cls, unitig_weights, links: Iterable[Sequence]) -> "Bigraph":
        """The clib.rs builder: links = (unitig_a, strand_a, unitig_b, strand_b)."""
        L = _lib.load()
        w = np.ascontiguousarray(unitig_weights, dtype=np.uint64)
        h = L.mtg_graph_builder_new(len(w))
        for (ua, sa, ub, sb) in links:
            L.mtg_graph_builder_merge(h, int(ua), 1 if sa else 0, int(ub), 1 if sb else 0)
        L.mtg_graph_builder_build(h, _ptr(w))
        return cls(h)

    @classmethod
    def from_unitig_links_arrays(cls, unitig_weights, links) -> "Bigraph":
        """The clib.rs builder over an int array [n, 4] of links (one call instead of one per link)."""
        L = _lib.load()
        w = np.ascontiguousarray(unitig_weights, dtype=np.uint64)
        lk = np.ascontiguousarray(links, dtype=np.int64)
        if lk.ndim != 2 or (len(lk) and lk.shape[1] != 4):
            raise ValueError("links must have shape [n, 4]")
        h = L.mtg_graph_builder_new(len(w))
        L.mtg_graph_builder_merge_links(h, len(lk), _ptr(lk) if len(lk) else None)
        L.mtg_graph_builder_build(h, _ptr(w))
        return cls(h)

    @property
    def handle(self) -> int:
        return self._h

    def release_device_cache(self) -> None:
        """mtg_graph_release_device_cache: the device copy of the original edges (+ their buckets) goes back to the driver."""
        self._L.mtg_graph_release_device_cache(self._h)

    def reset(self) -> None:
        """Drop all dummy edges again (the reference clones its graph instead, bin.rs:1069)."""
        self._L.mtg_graph_reset(self._h)

    def node_count(self) -> int:
        return int(self._L.mtg_graph_node_count(self._h))

    def edge_count(self) -> int:
        return int(self._L.mtg_graph_edge_count(self._h))

    def export(self) -> dict:
        V, E = self.node_count(), self.edge_count()
        out = {
            "mirror": np.zeros(V, np.uint32), "edge_from": np.zeros(E, np.uint32), "edge_to": np.zeros(E, np.uint32),
            "edge_weight": np.zeros(E, np.uint64), "edge_dummy_id": np.zeros(E, np.uint64),
            "edge_unitig": np.zeros(E, np.uint64), "edge_forwards": np.zeros(E, np.uint8),
        }
        self._L.mtg_graph_export(self._h, *[_ptr(out[k]) for k in
                                            ("mirror", "edge_from", "edge_to", "edge_weight", "edge_dummy_id",
                                             "edge_unitig", "edge_forwards")])
        return out

    _EXPORT_FIELDS = (("edge_from", np.uint32), ("edge_to", np.uint32), ("edge_weight", np.uint64), ("edge_dummy_id", np.uint64),
                      ("edge_unitig", np.uint64), ("edge_forwards", np.uint8))

    def export_mirror(self) -> np.ndarray:
        m = np.empty(self.node_count(), np.uint32)
        self._L.mtg_graph_export(self._h, _ptr(m), None, None, None, None, None, None)
        return m

    def original_edge_count(self) -> int:
        return int(self._L.mtg_graph_original_edge_count(self._h))

    def export_range(self, first_edge: int, n_edges: int, fields: Sequence[str]) -> dict:
        """Selected edge arrays of the edges [first_edge, first_edge + n_edges) (mtg_graph_export_range)."""
        out = {name: np.empty(n_edges, dt) for name, dt in self._EXPORT_FIELDS if name in fields}
        self._L.mtg_graph_export_range(self._h, first_edge, n_edges, *[_ptr(out.get(name)) for name, _ in self._EXPORT_FIELDS])
        return out

    def edge_data(self, e: int, _cache={}) -> MatchtigEdgeData:
        ex = self.export()
        return MatchtigEdgeData(int(ex["edge_unitig"][e]), bool(ex["edge_forwards"][e]), int(ex["edge_weight"][e]),
                                int(ex["edge_dummy_id"][e]))

    # ---- host sub-stages (exported for parity tests) ----
    def replay_claims(self, out_nodes, multiplicity, is_in_node, cand_start, cand_count, pool) -> np.ndarray:
        on = np.ascontiguousarray(out_nodes, np.uint32)
        mu = np.ascontiguousarray(multiplicity, np.int32)
        li = np.ascontiguousarray(is_in_node, np.uint8)
        cs = np.ascontiguousarray(cand_start, np.uint64)
        cc = np.ascontiguousarray(cand_count, np.uint32)
        po = np.ascontiguousarray(pool, np.uint64)
        if len(mu) != self.node_count() or len(li) != self.node_count():
            raise ValueError("classification arrays must have node_count entries")
        pp = C.POINTER(_lib.MtgPair)()
        n = self._L.mtg_replay_claims(self._h, len(on), _ptr(on), _ptr(mu), _ptr(li), _ptr(cs), _ptr(cc), _ptr(po), C.byref(pp))
        dt = np.dtype([("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])
        arr = np.zeros(n, dt)
        if n:
            C.memmove(arr.ctypes.data, pp, n * C.sizeof(_lib.MtgPair))
        self._L.mtg_free(pp)
        return arr

    def insert_pair_edges(self, pairs: np.ndarray) -> int:
        p = np.ascontiguousarray(pairs)
        return int(self._L.mtg_insert_pair_edges(self._h, _ptr(p), len(p)))

    def make_eulerian(self, dummy_edge_id: int, k: int) -> int:
        return int(self._L.mtg_make_eulerian(self._h, dummy_edge_id, k))

    def euler_cycles(self) -> list[list[int]]:
        return _take_walks(self._L, self._L.mtg_euler_cycles(self._h))

    def euler_cycles_records(self, record_format: int) -> list[list[int]]:
        """mtg_euler_cycles_records: 0 = wide records from adjacency, 1 = 32-byte records, 2 = wide (256-byte) records seeded from
        32-byte ones, 3 = 128-byte records seeded from 32-byte ones."""
        return _take_walks(self._L, self._L.mtg_euler_cycles_records(self._h, record_format))

    def euler_cycles_device(self, device_id: int = 0) -> list[list[int]]:
        """Euler bicycles on the GPU (valid walks, not the reference's order)."""
        return _take_walks(self._L, self._L.mtg_euler_cycles_device(self._h, device_id))

    def euler_cycles_device_np(self, device_id: int = 0):
        """(limits, edges) arrays of the GPU Euler bicycles."""
        return _take_walks_np(self._L, self._L.mtg_euler_cycles_device(self._h, device_id))

    def cut_cycles(self, cycles: list[list[int]], k: int) -> list[list[int]]:
        """greedytigs/mod.rs:726-789 on given closed walks (edge ids into this graph)."""
        ed = np.fromiter((e for c in cycles for e in c), dtype=np.uint32)
        lim = np.cumsum([len(c) for c in cycles], dtype=np.uint64) if len(cycles) else np.zeros(0, np.uint64)
        w = self._L.mtg_walks_from_arrays(len(lim), _ptr(lim) if len(lim) else None, _ptr(ed) if len(ed) else None)
        try:
            return _take_walks(self._L, self._L.mtg_cut_cycles(self._h, w, k))
        finally:
            self._L.mtg_walks_free(w)

    def finish_greedytigs(self, pairs: np.ndarray, k: int) -> list[list[int]]:
        p = np.ascontiguousarray(pairs)
        return _take_walks(self._L, self._L.mtg_finish_greedytigs(self._h, _ptr(p), len(p), k))

    def flatten_clib(self, tigs: list[list[int]]):
        """clib.rs:393-407 on already-computed walks (recomputed through the C-ABI for algorithm output)."""
        ex = self.export()
        eo, io, lim = [], [], []
        for t in tigs:
            for e in t:
                eo.append(int(ex["edge_unitig"][e]) * (1 if ex["edge_forwards"][e] else -1))
                io.append(0 if ex["edge_dummy_id"][e] == 0 else int(ex["edge_weight"][e]))
            lim.append(len(eo))
        return eo, io, lim


def _take_walks(L, wp) -> list[list[int]]:
    n, tot = int(L.mtg_walks_count(wp)), int(L.mtg_walks_total_edges(wp))
    lim = np.zeros(max(n, 1), np.uint64)
    ed = np.zeros(max(tot, 1), np.uint32)
    L.mtg_walks_export(wp, _ptr(lim), _ptr(ed))
    L.mtg_walks_free(wp)
    out, b = [], 0
    for i in range(n):
        out.append(ed[b:int(lim[i])].tolist())
        b = int(lim[i])
    return out


def _take_walks_np(L, wp):
    """(limits, edges) as numpy VIEWS of the library's walk arrays (no copy: 0.5 GB at the bench size); the walks are freed when
    both views are gone."""
    import weakref

    n, tot = int(L.mtg_walks_count(wp)), int(L.mtg_walks_total_edges(wp))
    if n and tot:
        lp, ep = C.c_void_p(), C.c_void_p()
        L.mtg_walks_data(wp, C.byref(lp), C.byref(ep))
        left = [2]

        def done():
            left[0] -= 1
            if left[0] == 0:
                L.mtg_walks_free(C.c_void_p(wp))

        raw_l = (C.c_char * (n * 8)).from_address(lp.value)
        raw_e = (C.c_char * (tot * 4)).from_address(ep.value)
        weakref.finalize(raw_l, done)
        weakref.finalize(raw_e, done)
        return np.frombuffer(raw_l, dtype=np.uint64), np.frombuffer(raw_e, dtype=np.uint32)
    lim = np.zeros(n, np.uint64)
    ed = np.zeros(tot, np.uint32)
    L.mtg_walks_export(wp, _ptr(lim) if n else None, _ptr(ed) if tot else None)
    L.mtg_walks_free(wp)
    return lim, ed


class DeviceGraph:
    """One GPU's resident copy of a Bigraph (mtg_device). Raises/aborts without a GPU: no CPU path."""

    def __init__(self, graph: Bigraph, k: int, device_id: int = 0, lower_bounds: bool = True, reserve_work: bool = False):
        """lower_bounds=False: mtg_device_create_opts(MTG_DEVICE_NO_LOWER_BOUNDS) -- the device graph of a caller that searches once
        (what mtg_compute_tigs_cfg builds); build_lower_bounds() adds them later. reserve_work=True: MTG_DEVICE_RESERVE_WORK -- the
        device memory the stages of a step take beside the graph is reserved now, in one piece (a caller that steps through the stages)."""
        self._L = _lib.load()
        if self._L.mtg_device_count() <= device_id:
            raise RuntimeError(f"no HIP device {device_id}: the matchtigs_amd device stage has no CPU fallback")
        self.graph = graph
        self.k = k
        self._d = self._L.mtg_device_create_opts(graph.handle, k, device_id, (0 if lower_bounds else 1) | (2 if reserve_work else 0))
        self.n_sources = None

    def build_lower_bounds(self, stream: int = 0) -> float:
        """mtg_device_build_lower_bounds; returns the GPU milliseconds of the precompute (mtg_device_lower_bounds_ms)."""
        self._L.mtg_device_build_lower_bounds(self._d, stream)
        return self.lower_bounds_ms()

    def lower_bounds_ms(self) -> float:
        return float(self._L.mtg_device_lower_bounds_ms(self._d))

    def __del__(self):
        d, self._d = getattr(self, "_d", None), None
        if d:
            self._L.mtg_device_free(d)

    @property
    def handle(self) -> int:
        return self._d

    def graph_bytes(self) -> int:
        return int(self._L.mtg_device_graph_bytes(self._d))

    def classify(self, stream: int = 0) -> int:
        self.n_sources = int(self._L.mtg_classify(self._d, stream))
        return self.n_sources

    def classify_download(self, stream: int = 0):
        V = self.graph.node_count()
        on = np.zeros(self.n_sources, np.uint32)
        mu = np.zeros(V, np.int32)
        li = np.zeros(V, np.uint8)
        self._L.mtg_classify_download(self._d, stream, _ptr(on) if len(on) else None, _ptr(mu) if V else None,
                                      _ptr(li) if V else None)
        return on, mu, li

    def sssp_candidates(self, src_begin: int, src_end: int, d_pool: int, pool_capacity: int, d_cand_start: int,
                        d_cand_count: int, stream: int = 0):
        """Device pointers in, returns (status, pool_needed). status 1 = pool too small."""
        needed = C.c_uint64()
        rc = self._L.mtg_sssp_candidates(self._d, stream, src_begin, src_end, d_pool, pool_capacity, d_cand_start,
                                         d_cand_count, C.byref(needed))
        return int(rc), int(needed.value)

    def last_sssp_kernel_ms(self) -> float:
        return float(self._L.mtg_last_sssp_kernel_ms(self._d))

    def replay_claims_device(self, d_cand_start: int, d_cand_count: int, d_pool: int, stream: int = 0) -> np.ndarray:
        """The claim loop on the GPU over device-resident candidate arrays of ALL sources -> pairs (host numpy)."""
        pp = C.POINTER(_lib.MtgPair)()
        n = self._L.mtg_replay_claims_device(self._d, stream, self.n_sources, d_cand_start, d_cand_count, d_pool, C.byref(pp))
        return _adopt_pairs(self._L, pp, n)

    def replay_claims_resident(self, d_cand_start: int, d_cand_count: int, d_pool: int, stream: int = 0) -> int:
        """The claim loop on the GPU; the pairs stay in HBM for finish_greedytigs_resident_np. Returns their number."""
        return int(self._L.mtg_replay_claims_resident(self._d, stream, self.n_sources, d_cand_start, d_cand_count, d_pool))

    def download_resident_pairs(self) -> np.ndarray:
        pp = C.POINTER(_lib.MtgPair)()
        n = self._L.mtg_download_resident_pairs(self._d, C.byref(pp))
        return _adopt_pairs(self._L, pp, n)

    def set_replay_tuning(self, windows: int = 0, block: int = 0, grid: int = 0, role_mod: int = 0, plain_barrier: bool = False) -> None:
        """mtg_set_replay_tuning: launch geometry of the claim replay (0 = the engine's choice); never changes the pair list."""
        self._L.mtg_set_replay_tuning(self._d, windows, block, grid, role_mod, 1 if plain_barrier else 0)

    def last_replay_ms(self) -> dict:
        out = (C.c_double * 2)()
        self._L.mtg_last_replay_ms(self._d, out)
        return {"rounds_kernel_ms": float(out[0]), "gpu_ms": float(out[1])}

    def last_replay_rounds(self) -> int:
        return int(self._L.mtg_last_replay_rounds(self._d))

    def last_replay_visits(self) -> int:
        return int(self._L.mtg_last_replay_visits(self._d))

    def last_sssp_levels(self) -> list[dict]:
        ms = (C.c_double * 8)()
        src = (C.c_uint64 * 8)()
        n = self._L.mtg_last_sssp_levels(self._d, ms, src, 8)
        return [{"level": i, "ms": float(ms[i]), "sources": int(src[i]),
                 "kernel": self._L.mtg_last_sssp_level_name(self._d, i).decode()} for i in range(n)]

    def sssp_count(self, src_begin: int, src_end: int, stream: int = 0) -> dict:
        st = _lib.MtgSsspStats()
        self._L.mtg_sssp_count(self._d, stream, src_begin, src_end, C.byref(st))
        return st.as_dict()

    def sssp_count_visited(self, src_begin: int, src_end: int, stream: int = 0) -> dict:
        """Units of the search the default plan really runs (goal-directed pruning; mtg_engine.h): sources = sources searched."""
        st = _lib.MtgSsspStats()
        self._L.mtg_sssp_count_visited(self._d, stream, src_begin, src_end, C.byref(st))
        return st.as_dict()

    def prunes(self) -> bool:
        return bool(self._L.mtg_sssp_prunes(self._d))

    def last_searched_sources(self) -> int:
        return int(self._L.mtg_last_sssp_searched_sources(self._d))

    def set_plan(self, plan: int) -> int:
        """0 = default (path enumeration level + cooperative cascade), 1 = cooperative cascade only, 2 / 3 = plan 0 with
        quad-cooperative / per-lane block gathers regardless of the graph size, + 4 = without the goal-directed pruning, + 8 = the
        enumeration level on one workgroup (tests) (mtg_engine.h)."""
        return int(self._L.mtg_set_sssp_plan(self._d, plan))


def compute_pairs(devices: Sequence[DeviceGraph]) -> np.ndarray:
    """mtg_compute_pairs: SSSP sharded over the given (classified) device copies of one graph, gather on the first, claim replay."""
    L = _lib.load()
    arr = (C.c_void_p * len(devices))(*[d.handle for d in devices])
    pp = C.POINTER(_lib.MtgPair)()
    n = L.mtg_compute_pairs(arr, len(devices), C.byref(pp))
    return _adopt_pairs(L, pp, n)


def partition_sources(device: DeviceGraph, parts: int) -> list[int]:
    cuts = (C.c_uint64 * (parts + 1))()
    _lib.load().mtg_partition_sources(device.handle, parts, cuts)
    return [int(x) for x in cuts]


PAIR_DTYPE = np.dtype([("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)])


def _adopt_pairs(L, pp, n: int) -> np.ndarray:
    """A numpy view of the library's malloc'd mtg_pair array (no copy); mtg_free runs when the last view is gone."""
    import weakref

    if not n:
        L.mtg_free(pp)
        return np.zeros(0, PAIR_DTYPE)
    addr = C.cast(pp, C.c_void_p).value
    raw = (C.c_char * (n * C.sizeof(_lib.MtgPair))).from_address(addr)
    weakref.finalize(raw, L.mtg_free, C.c_void_p(addr))
    return np.frombuffer(raw, dtype=PAIR_DTYPE)


class TigAlgorithm:
    """implementation/mod.rs:49-59."""

    Configuration = None

    @classmethod
    def compute_tigs(cls, graph: Bigraph, configuration) -> list[list[int]]:
        raise NotImplementedError


class GreedytigAlgorithm(TigAlgorithm):
    """greedytigs/mod.rs:75-90: walks of edge ids into the (mutated) graph."""

    Configuration = GreedytigAlgorithmConfiguration

    @classmethod
    def compute_tigs(cls, graph: Bigraph, configuration: GreedytigAlgorithmConfiguration):
        L = _lib.load()
        c = configuration.to_c()
        return _take_walks(L, L.mtg_compute_tigs_cfg(graph.handle, 5, C.byref(c)))

    @classmethod
    def compute_tigs_np(cls, graph: Bigraph, configuration: GreedytigAlgorithmConfiguration):
        """The same as flat numpy arrays (exclusive tig ends, edge ids): large graphs."""
        L = _lib.load()
        c = configuration.to_c()
        return _take_walks_np(L, L.mtg_compute_tigs_cfg(graph.handle, 5, C.byref(c)))


class EulertigAlgorithm(TigAlgorithm):
    """eulertigs/mod.rs:18-39."""

    Configuration = EulertigAlgorithmConfiguration

    @classmethod
    def compute_tigs(cls, graph: Bigraph, configuration: EulertigAlgorithmConfiguration):
        L = _lib.load()
        c = configuration.to_c()
        return _take_walks(L, L.mtg_compute_eulertigs_cfg(graph.handle, C.byref(c)))

    @classmethod
    def compute_tigs_np(cls, graph: Bigraph, configuration: EulertigAlgorithmConfiguration):
        L = _lib.load()
        c = configuration.to_c()
        return _take_walks_np(L, L.mtg_compute_eulertigs_cfg(graph.handle, C.byref(c)))


class MatchingInstance:
    """The minimum-perfect-matching instance of optimal matchtigs (matchtigs/mod.rs:150-719), built from the GPU's candidate
    lists: ``write`` produces the matcher's input file, ``read_solution`` turns the matcher's output into matched pairs."""

    def __init__(self, graph: Bigraph, k: int, device_id: int = 0, _handle=None):
        self._L = _lib.load()
        if _handle is not None:
            self._m = _handle
            return
        c = GreedytigAlgorithmConfiguration(1, k, device_ids=(device_id,)).to_c()
        self._m = self._L.mtg_matching_instance(graph.handle, C.byref(c))

    @classmethod
    def from_lists(cls, graph: Bigraph, k: int, out_nodes, multiplicity, cand_start, cand_count, pool) -> "MatchingInstance":
        """The host stage alone over caller-held candidate lists (mtg_matching_instance_from_lists)."""
        L = _lib.load()
        on = np.ascontiguousarray(out_nodes, np.uint32)
        mu = np.ascontiguousarray(multiplicity, np.int32)
        cs = np.ascontiguousarray(cand_start, np.uint64)
        cc = np.ascontiguousarray(cand_count, np.uint32)
        po = np.ascontiguousarray(pool, np.uint64)
        h = L.mtg_matching_instance_from_lists(graph.handle, k, len(on), _ptr(on) if len(on) else None, _ptr(mu),
                                               _ptr(cs) if len(cs) else None, _ptr(cc) if len(cc) else None,
                                               _ptr(po) if len(po) else None)
        return cls(graph, k, _handle=h)

    def stats(self) -> dict:
        st = _lib.MtgMatchingStats()
        self._L.mtg_matching_get_stats(self._m, C.byref(st))
        return st.as_dict()

    def write(self, path: str) -> int:
        return int(self._L.mtg_matching_write(self._m, str(path).encode()))

    def read_solution(self, path: str) -> np.ndarray:
        pp = C.POINTER(_lib.MtgPair)()
        n = self._L.mtg_matching_read_solution(self._m, str(path).encode(), C.byref(pp))
        arr = np.zeros(n, np.dtype([("out", np.uint32), ("in", np.uint32), ("dist", np.uint64)]))
        if n:
            C.memmove(arr.ctypes.data, pp, n * C.sizeof(_lib.MtgPair))
        self._L.mtg_free(pp)
        return arr

    def close(self):
        if self._m:
            self._L.mtg_matching_free(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchtigAlgorithm(TigAlgorithm):
    """matchtigs/mod.rs:47-62: optimal matchtigs; the matching itself is solved by the external matcher the configuration names."""

    Configuration = MatchtigAlgorithmConfiguration

    @classmethod
    def compute_tigs(cls, graph: Bigraph, configuration: MatchtigAlgorithmConfiguration):
        L = _lib.load()
        c = configuration.to_c()
        return _take_walks(L, L.mtg_compute_matchtigs_cfg(graph.handle, C.byref(c)))

    @staticmethod
    def finish(graph: Bigraph, pairs: np.ndarray, k: int):
        """matchtigs/mod.rs:797-935 on already matched pairs (mtg_finish_matchtigs_cfg)."""
        L = _lib.load()
        p = np.ascontiguousarray(pairs)
        c = GreedytigAlgorithmConfiguration(1, k).to_c()
        return _take_walks(L, L.mtg_finish_matchtigs_cfg(graph.handle, _ptr(p) if len(p) else None, len(p), C.byref(c)))


def finish_greedytigs_np(graph: Bigraph, pairs: np.ndarray, k: int, euler_mode: EulerMode = EulerMode.HostReferenceOrder,
                         device_id: int = 0, finish_stage: FinishStage = FinishStage.Auto):
    """mtg_finish_greedytigs_cfg returning flat numpy walks (limits, edges) -- for large graphs."""
    L = _lib.load()
    p = np.ascontiguousarray(pairs)
    c = GreedytigAlgorithmConfiguration(1, k, euler_mode=euler_mode, device_ids=(device_id,), finish_stage=finish_stage).to_c()
    return _take_walks_np(L, L.mtg_finish_greedytigs_cfg(graph.handle, _ptr(p) if len(p) else None, len(p), C.byref(c)))


class Tigs:
    """Handle of a tig set (mtg_walks). After a finish on the GPU the tigs are still in HBM: count() and total_edges() cost nothing,
    arrays() brings them to the host -- once, through the pinned ring -- and returns (limits, edges) as views of the library's arrays
    (valid while this object lives)."""

    def __init__(self, L, wp):
        self._L, self._wp = L, wp

    def __del__(self):
        wp, self._wp = getattr(self, "_wp", None), None
        if wp:
            self._L.mtg_walks_free(C.c_void_p(wp))

    def count(self) -> int:
        return int(self._L.mtg_walks_count(self._wp))

    def total_edges(self) -> int:
        return int(self._L.mtg_walks_total_edges(self._wp))

    def arrays(self):
        n, tot = self.count(), self.total_edges()
        if not (n and tot):
            return np.zeros(n, np.uint64), np.zeros(tot, np.uint32)
        lp, ep = C.c_void_p(), C.c_void_p()
        self._L.mtg_walks_data(self._wp, C.byref(lp), C.byref(ep))
        lim = np.frombuffer((C.c_char * (n * 8)).from_address(lp.value), dtype=np.uint64)
        ed = np.frombuffer((C.c_char * (tot * 4)).from_address(ep.value), dtype=np.uint32)
        self._views = (lim, ed)
        return lim, ed


def finish_greedytigs_resident(graph: Bigraph, device: "DeviceGraph", k: int, euler_mode: EulerMode = EulerMode.HostReferenceOrder,
                               device_id: int = 0, finish_stage: FinishStage = FinishStage.Auto) -> Tigs:
    """mtg_finish_greedytigs_resident as a handle: the tigs of a finish on the GPU stay in HBM until Tigs.arrays() asks for them."""
    L = _lib.load()
    c = GreedytigAlgorithmConfiguration(1, k, euler_mode=euler_mode, device_ids=(device_id,), finish_stage=finish_stage).to_c()
    return Tigs(L, L.mtg_finish_greedytigs_resident(graph.handle, device.handle, C.byref(c)))


def finish_greedytigs_resident_np(graph: Bigraph, device: "DeviceGraph", k: int, euler_mode: EulerMode = EulerMode.HostReferenceOrder,
                                  device_id: int = 0, finish_stage: FinishStage = FinishStage.Auto):
    """mtg_finish_greedytigs_resident: the finish over the pairs the last replay_claims_resident left on the GPU."""
    L = _lib.load()
    c = GreedytigAlgorithmConfiguration(1, k, euler_mode=euler_mode, device_ids=(device_id,), finish_stage=finish_stage).to_c()
    return _take_walks_np(L, L.mtg_finish_greedytigs_resident(graph.handle, device.handle, C.byref(c)))


RECORD_FORMATS = {None: 0, "auto": 0, "lean": 1, "mid": 2, "wide": 3}


def set_finish_tuning(records=None, wait_for_records: bool = False, no_pin: bool = False, no_edge_cache: bool = False,
                      record_delay_us: int = 0, keep_awake: bool = False, no_cut_first: bool = False) -> None:
    """mtg_set_finish_tuning: walk-record format of the reference-order mode ("lean" 32-byte / "mid" 128-byte / "wide" 256-byte
    records, None = the engine's choice), whether the walk waits for all of its records, page-locking, the graph's device cache, and
    a delay per arriving slice of records (tests). Process-wide; never changes a result."""
    flags = (1 if wait_for_records else 0) | (2 if no_pin else 0) | (4 if no_edge_cache else 0) | (8 if keep_awake else 0) | (16 if no_cut_first else 0)
    _lib.load().mtg_set_finish_tuning(RECORD_FORMATS[records], flags, int(record_delay_us))


def set_default_device(device_id: int) -> None:
    """mtg_set_default_device: the GPU whose memory a graph's construction reserves ahead of the call that follows."""
    _lib.load().mtg_set_default_device(device_id)


def set_reserve_ahead(on: bool) -> None:
    """mtg_set_reserve_ahead: False = host-only graph constructors reserve nothing on any GPU (no helper thread)."""
    _lib.load().mtg_set_reserve_ahead(1 if on else 0)


def release_device_memory(device_id: int = 0) -> None:
    """mtg_release_device_memory: the work arrays the finishing stages keep on that GPU between calls."""
    _lib.load().mtg_release_device_memory(device_id)


def device_arena_stats(device_id: int = 0, reset_peak: bool = False) -> dict:
    """mtg_device_arena_stats: bytes in chunks, bytes live, peak of live bytes since the last reset, driver allocations so far."""
    a = (C.c_uint64 * 4)()
    _lib.load().mtg_device_arena_stats(device_id, a, 1 if reset_peak else 0)
    return {"chunk_bytes": int(a[0]), "live_bytes": int(a[1]), "peak_bytes": int(a[2]), "driver_allocations": int(a[3])}


def device_memory_held(device_id: int = 0) -> int:
    return int(_lib.load().mtg_device_memory_held(device_id))


def last_finish_device_times() -> dict:
    """Segments of the last device finish on this thread (mtg_last_finish_device_times)."""
    out = (C.c_double * 6)()
    _lib.load().mtg_last_finish_device_times(out)
    return {"insert_eulerise_s": out[0], "host_graph_s": out[1], "euler_s": out[2], "cut_s": out[3], "euler_kernel_ms": out[4],
            "breaking_biedges": int(out[5])}


def last_finish_device_stage_ms() -> dict:
    """GPU ms per stage of the last device finish (mtg_last_finish_device_stage_ms) + its dart and unit counts."""
    out = (C.c_double * 6)()
    _lib.load().mtg_last_finish_device_stage_ms(out)
    return {"insert_eulerise_ms": out[0], "records_ms": out[1], "decomposition_ms": out[2], "cut_ms": out[3], "darts": int(out[4]),
            "units": int(out[5])}


def last_performance_data() -> dict:
    """greedytigs/mod.rs:647-673 counters of the last greedy run with PerformanceDataType.Complete."""
    out = _lib.MtgDijkstraPerformanceData()
    _lib.load().mtg_last_performance_data(C.byref(out))
    return out.as_dict()


class UnitigStore:
    """Sequence store filled by read_bcalm2 (replaces DefaultSequenceStore<DnaAlphabet>, bin.rs:871)."""

    def __init__(self, handle: int):
        self._h = handle
        self._L = _lib.load()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.mtg_unitigs_free(h)

    @property
    def handle(self) -> int:
        return self._h

    def __len__(self) -> int:
        return int(self._L.mtg_unitigs_count(self._h))

    def sequences(self) -> list[str]:
        n = len(self)
        off = np.ctypeslib.as_array(C.cast(self._L.mtg_unitigs_offsets(self._h), C.POINTER(C.c_uint64)), shape=(n + 1,))
        data = C.string_at(self._L.mtg_unitigs_data(self._h), int(off[n])).decode()
        return [data[int(off[i]):int(off[i + 1])] for i in range(n)]


def read_bcalm2(path: str, k: int):
    """`--bcalm-in path -k k` (bin.rs:902-912): BCALM2/GGCAT unitig FASTA (optionally .gz) -> (Bigraph, UnitigStore)."""
    L = _lib.load()
    st = C.c_void_p()
    g = L.mtg_read_bcalm2(str(path).encode(), k, C.byref(st))
    return Bigraph(g), UnitigStore(st.value)


def compute_tigs_to_fasta_file(graph: Bigraph, store: UnitigStore, algorithm: int, k: int, path: Optional[str],
                               compression_level: int = 6, device_id: int = 0, gfa_path: Optional[str] = None,
                               gfa_header: Optional[str] = None, duplication_bitvector_path: Optional[str] = None,
                               configuration: Optional[GreedytigAlgorithmConfiguration] = None) -> dict:
    """compute (3 = eulertigs, 5 = greedy matchtigs) + spell + write FASTA and/or GFA, all inside the library."""
    import time

    L = _lib.load()
    t0 = time.perf_counter()
    c = (configuration or GreedytigAlgorithmConfiguration(1, k, device_ids=(device_id,))).to_c()
    w = L.mtg_compute_tigs_cfg(graph.handle, algorithm, C.byref(c))
    t1 = time.perf_counter()
    n_tigs = int(L.mtg_walks_count(w))
    nbytes = gbytes = 0
    spell_dev = int(c.device_ids[0])  # spell where the tigs were computed (--device / cfg.device_ids)
    if path:
        nbytes = int(L.mtg_write_tigs_text_file_device(graph.handle, w, k, store.handle, 0, None, str(path).encode(), compression_level, spell_dev))
    if gfa_path:
        gbytes = int(L.mtg_write_tigs_text_file_device(graph.handle, w, k, store.handle, 1, gfa_header.encode() if gfa_header else None,
                                                       str(gfa_path).encode(), compression_level, spell_dev))
    if duplication_bitvector_path:
        L.mtg_write_tigs_duplication_bitvector_file(graph.handle, w, str(duplication_bitvector_path).encode())
    t2 = time.perf_counter()
    L.mtg_walks_free(w)
    return {"tigs": n_tigs, "fasta_bytes": nbytes, "gfa_bytes": gbytes, "compute_s": t1 - t0, "write_s": t2 - t1}


def write_duplication_bitvector(graph: Bigraph, tigs) -> bytes:
    """implementation/mod.rs:668-702 through the C-ABI: per tig a line of '1' (original k-mer) / '0' (duplicate) characters."""
    L = _lib.load()
    if isinstance(tigs, tuple):
        lim, ed = np.ascontiguousarray(tigs[0], np.uint64), np.ascontiguousarray(tigs[1], np.uint32)
    else:
        ed = np.fromiter((e for t in tigs for e in t), dtype=np.uint32)
        lim = np.cumsum([len(t) for t in tigs], dtype=np.uint64) if len(tigs) else np.zeros(0, np.uint64)
    out = C.c_void_p()
    n = L.mtg_write_duplication_bitvector(graph.handle, len(lim), _ptr(lim) if len(lim) else None, _ptr(ed) if len(ed) else None,
                                          C.byref(out))
    data = C.string_at(out, n)
    L.mtg_free(out)
    return data


def write_walks_gfa(graph: Bigraph, tigs, unitigs: Sequence[str], k: int, header: Optional[str] = None) -> bytes:
    """bin.rs:667-818 through the C-ABI: GFA1 text (header line, then one S record per tig)."""
    return write_walks_fasta(graph, tigs, unitigs, k, _gfa=True, _header=header)


def write_walks_text_device(graph: Bigraph, tigs, unitigs, k: int, gfa: bool = False, header: Optional[str] = None,
                            device_id: int = 0) -> bytes:
    """The same text spelled on the GPU (mtg_write_walks_text_device). unitigs: list of str, or (uint8 array, offsets)."""
    L = _lib.load()
    if isinstance(tigs, tuple):
        lim, ed = np.ascontiguousarray(tigs[0], np.uint64), np.ascontiguousarray(tigs[1], np.uint32)
    else:
        ed = np.fromiter((e for t in tigs for e in t), dtype=np.uint32)
        lim = np.cumsum([len(t) for t in tigs], dtype=np.uint64) if len(tigs) else np.zeros(0, np.uint64)
    if isinstance(unitigs, tuple):
        cat = np.ascontiguousarray(unitigs[0], np.uint8).tobytes()
        off = np.ascontiguousarray(unitigs[1], np.uint64)
    else:
        cat = "".join(unitigs).encode()
        off = np.zeros(len(unitigs) + 1, np.uint64)
        off[1:] = np.cumsum([len(u) for u in unitigs])
    out = C.c_void_p()
    n = L.mtg_write_walks_text_device(graph.handle, len(lim), _ptr(lim) if len(lim) else None, _ptr(ed) if len(ed) else None, k, cat,
                                      _ptr(off), 1 if gfa else 0, header.encode() if header else None, device_id, C.byref(out))
    data = C.string_at(out, n)
    L.mtg_free(out)
    return data


def last_spell_kernel() -> dict:
    L = _lib.load()
    return {"ms": float(L.mtg_last_spell_kernel_ms()), "bytes": int(L.mtg_last_spell_bytes())}


def write_walks_fasta(graph: Bigraph, tigs, unitigs: Sequence[str], k: int, _gfa: bool = False,
                      _header: Optional[str] = None) -> bytes:
    """bin.rs:466-606 through the C-ABI: tigs = list of edge-id lists (or (limits, edges) numpy pair) -> FASTA bytes."""
    L = _lib.load()
    if isinstance(tigs, tuple):
        lim, ed = np.ascontiguousarray(tigs[0], np.uint64), np.ascontiguousarray(tigs[1], np.uint32)
    else:
        ed = np.fromiter((e for t in tigs for e in t), dtype=np.uint32)
        lim = np.cumsum([len(t) for t in tigs], dtype=np.uint64) if len(tigs) else np.zeros(0, np.uint64)
    cat = "".join(unitigs).encode()
    off = np.zeros(len(unitigs) + 1, np.uint64)
    off[1:] = np.cumsum([len(u) for u in unitigs])
    out = C.c_void_p()
    if _gfa:
        n = L.mtg_write_walks_gfa(graph.handle, len(lim), _ptr(lim) if len(lim) else None, _ptr(ed) if len(ed) else None, k,
                                  cat, _ptr(off), _header.encode() if _header else None, C.byref(out))
    else:
        n = L.mtg_write_walks_fasta(graph.handle, len(lim), _ptr(lim) if len(lim) else None, _ptr(ed) if len(ed) else None, k,
                                    cat, _ptr(off), C.byref(out))
    data = C.string_at(out, n)
    L.mtg_free(out)
    return data


def last_euler_kernel_ms() -> float:
    return float(_lib.load().mtg_last_euler_kernel_ms())


def last_phase_seconds() -> dict:
    L = _lib.load()
    a = (C.c_double * 8)()
    L.mtg_last_phase_seconds(a)
    names = ["device_build", "classify", "sssp", "download", "replay", "eulerise", "euler", "cut"]
    return {n: float(a[i]) for i, n in enumerate(names)}


# ---- the reference's C-ABI, driven from Python exactly like a C caller would (clib.rs) ----------
def clib_compute_tigs(unitig_weights, links, tig_algorithm: int, threads: int, k: int, matching_file_prefix: str = "",
                      matcher_path: str = ""):
    """matchtigs_initialise_graph -> merge_nodes* -> build_graph -> compute_tigs. Returns
    (n_tigs, tigs_edge_out, tigs_insert_out, tigs_out_limits) trimmed to their used lengths."""
    L = _lib.load()
    w = np.ascontiguousarray(unitig_weights, dtype=np.uint64)
    u = len(w)
    data = L.matchtigs_initialise_graph(u)
    for (ua, sa, ub, sb) in links:
        L.matchtigs_merge_nodes(data, int(ua), bool(sa), int(ub), bool(sb))
    L.matchtigs_build_graph(data, _ptr(w))
    edge_count = 2 * u  # clib.rs:332-348: outputs sized by the edge count before dummies
    eo = np.zeros(max(2 * edge_count, 1), np.int64)
    io = np.zeros(max(2 * edge_count, 1), np.uint64)
    lo = np.zeros(max(edge_count, 1), np.uint64)
    n = int(L.matchtigs_compute_tigs(data, tig_algorithm, threads, k, str(matching_file_prefix).encode(),
                                     str(matcher_path).encode(), _ptr(eo), _ptr(io), _ptr(lo)))
    total = int(lo[n - 1]) if n else 0
    return n, eo[:total].copy(), io[:total].copy(), lo[:n].copy()
