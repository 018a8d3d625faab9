"""ctypes loader for libmatchtigs.so (the C-ABI library; include/matchtigs.h + include/mtg_engine.h).

Fails loudly when the library is missing: there is no Python/CPU fallback for the device stage.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
LIB_PATH = PKG_DIR / "libmatchtigs.so"
INCLUDE_DIR = REPO_DIR / "include"

_lib = None


class MtgPair(C.Structure):
    _fields_ = [("out_node", C.c_uint32), ("in_node", C.c_uint32), ("distance", C.c_uint64)]


class MtgSsspStats(C.Structure):
    _fields_ = [
        ("sources", C.c_uint64),
        ("settled_nodes", C.c_uint64),
        ("relaxed_edges", C.c_uint64),
        ("emitted", C.c_uint64),
        ("relax_attempts", C.c_uint64),
        ("overflow_sources", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


MTG_MAX_DEVICES = 8


class MtgConfig(C.Structure):
    """mtg_config (include/mtg_engine.h): the fields of GreedytigAlgorithmConfiguration + euler_mode / device_ids."""

    _fields_ = [
        ("struct_size", C.c_uint64),
        ("threads", C.c_uint64),
        ("k", C.c_uint64),
        ("staged_parallelism_divisor", C.c_double),
        ("resource_limit_factor", C.c_uint64),
        ("node_weight_array_type", C.c_int32),
        ("heap_type", C.c_int32),
        ("performance_data_type", C.c_int32),
        ("euler_mode", C.c_int32),
        ("n_devices", C.c_int32),
        ("device_ids", C.c_int32 * MTG_MAX_DEVICES),
        ("matching_file_prefix", C.c_char_p),
        ("matcher_path", C.c_char_p),
        ("finish_stage", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class MtgMatchingStats(C.Structure):
    _fields_ = [
        ("transformed_node_count", C.c_uint64),
        ("edge_count", C.c_uint64),
        ("wcc_amount", C.c_uint64),
        ("matching_node_count", C.c_uint64),
        ("matching_edge_count", C.c_uint64),
        ("mirror_biedges", C.c_uint64),
        ("mirror_expanded_biedges", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class MtgDijkstraPerformanceData(C.Structure):
    _fields_ = [
        ("dijkstras", C.c_uint64),
        ("iterations", C.c_uint64),
        ("heap_pushes", C.c_uint64),
        ("unnecessary_heap_elements", C.c_uint64),
        ("max_max_heap_size", C.c_uint64),
        ("max_max_distance_array_size", C.c_uint64),
        ("sum_max_heap_size", C.c_uint64),
        ("sum_max_distance_array_size", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def declared_symbols() -> list[str]:
    """Every function name declared in include/*.h (used by the symbol-export test)."""
    names: list[str] = []
    for h in sorted(INCLUDE_DIR.glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names += re.findall(r"\b((?:mtg|matchtigs)_[a-z0-9_]+)\s*\(", text)
    # (include/mtg_policy.h defines static inline helpers, mtg_policy_*: compiled into their users, not exported)
    return sorted(n for n in set(names) if not n.startswith("mtg_policy_"))


def _share_hip_runtime_with_torch() -> None:
    """Process-level plumbing. PyTorch wheels bundle their own libamdhip64 and request it as
    'libamdhip64.so', while libmatchtigs.so requests ROCm's 'libamdhip64.so.7'. If libmatchtigs is loaded
    first, torch later maps a SECOND HIP runtime into the process and neither sees the other's device
    pointers/streams. Loading torch's copy first makes the dynamic loader satisfy our NEEDED entry with
    it (same SONAME), so both share one runtime. A pure C caller never goes through here."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load():
    global _lib
    if _lib is not None:
        return _lib
    import os

    lib_path = Path(os.environ["MATCHTIGS_LIBRARY"]) if os.environ.get("MATCHTIGS_LIBRARY") else LIB_PATH  # (another build of the library, e.g. `make flipped`)
    if not lib_path.exists():
        raise RuntimeError(
            f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C matchtigs_amd/csrc`. matchtigs_amd has no fallback without its HIP library."
        )
    _share_hip_runtime_with_torch()
    L = C.CDLL(str(lib_path))
    vp, u32, u64, i64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_int32
    P = C.POINTER
    sig = {
        "mtg_version": (C.c_char_p, []),
        "mtg_device_count": (C.c_int, []),
        "mtg_policies": (C.c_uint, []),
        "mtg_graph_from_edges": (vp, [u64, vp, u64, vp, vp, vp]),
        "mtg_graph_builder_new": (vp, [u64]),
        "mtg_graph_builder_merge": (None, [vp, u64, C.c_int, u64, C.c_int]),
        "mtg_graph_builder_merge_links": (None, [vp, u64, vp]),
        "mtg_graph_builder_build": (None, [vp, vp]),
        "mtg_graph_free": (None, [vp]),
        "mtg_graph_reset": (None, [vp]),
        "mtg_graph_node_count": (u64, [vp]),
        "mtg_graph_edge_count": (u64, [vp]),
        "mtg_graph_export": (None, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "mtg_graph_export_range": (None, [vp, u64, u64, vp, vp, vp, vp, vp, vp]),
        "mtg_graph_original_edge_count": (u64, [vp]),
        "mtg_device_create": (vp, [vp, u64, C.c_int]),
        "mtg_device_free": (None, [vp]),
        "mtg_device_graph_bytes": (u64, [vp]),
        "mtg_classify": (u64, [vp, vp]),
        "mtg_classify_download": (None, [vp, vp, vp, vp, vp]),
        "mtg_classify_d_out_nodes": (vp, [vp]),
        "mtg_sssp_candidates": (C.c_int, [vp, vp, u64, u64, vp, u64, vp, vp, P(u64)]),
        "mtg_last_sssp_kernel_ms": (C.c_double, [vp]),
        "mtg_last_sssp_levels": (C.c_int, [vp, P(C.c_double), P(u64), C.c_int]),
        "mtg_last_sssp_level_name": (C.c_char_p, [vp, C.c_int]),
        "mtg_sssp_count": (None, [vp, vp, u64, u64, P(MtgSsspStats)]),
        "mtg_sssp_count_visited": (None, [vp, vp, u64, u64, P(MtgSsspStats)]),
        "mtg_sssp_prunes": (C.c_int, [vp]),
        "mtg_last_sssp_searched_sources": (u64, [vp]),
        "mtg_set_sssp_plan": (C.c_int, [vp, C.c_int]),
        "mtg_replay_claims_device": (u64, [vp, vp, u64, vp, vp, vp, P(P(MtgPair))]),
        "mtg_last_replay_rounds": (C.c_int, [vp]),
        "mtg_last_replay_visits": (u64, [vp]),
        "mtg_compute_pairs": (u64, [P(vp), C.c_int, P(P(MtgPair))]),
        "mtg_last_gather_ms": (C.c_double, []),
        "mtg_partition_sources": (None, [vp, C.c_int, P(u64)]),
        "mtg_replay_claims": (u64, [vp, u64, vp, vp, vp, vp, vp, vp, P(P(MtgPair))]),
        "mtg_free": (None, [vp]),
        "mtg_finish_greedytigs": (vp, [vp, vp, u64, u64]),
        "mtg_compute_eulertigs": (vp, [vp, u64]),
        "mtg_insert_pair_edges": (u64, [vp, vp, u64]),
        "mtg_make_eulerian": (u64, [vp, u64, u64]),
        "mtg_euler_cycles": (vp, [vp]),
        "mtg_euler_cycles_records": (vp, [vp, C.c_int]),
        "mtg_cut_cycles": (vp, [vp, vp, u64]),
        "mtg_euler_cycles_device": (vp, [vp, C.c_int]),
        "mtg_config_init": (None, [P(MtgConfig), u64, u64]),
        "mtg_finish_greedytigs_cfg": (vp, [vp, vp, u64, P(MtgConfig)]),
        "mtg_compute_eulertigs_cfg": (vp, [vp, P(MtgConfig)]),
        "mtg_matching_instance": (vp, [vp, P(MtgConfig)]),
        "mtg_matching_instance_from_lists": (vp, [vp, u64, u64, vp, vp, vp, vp, vp]),
        "mtg_matching_get_stats": (None, [vp, P(MtgMatchingStats)]),
        "mtg_matching_write": (u64, [vp, C.c_char_p]),
        "mtg_matching_read_solution": (u64, [vp, C.c_char_p, P(P(MtgPair))]),
        "mtg_matching_free": (None, [vp]),
        "mtg_finish_matchtigs_cfg": (vp, [vp, vp, u64, P(MtgConfig)]),
        "mtg_compute_matchtigs_cfg": (vp, [vp, P(MtgConfig)]),
        "mtg_compute_tigs_cfg": (vp, [vp, u64, P(MtgConfig)]),
        "mtg_compute_tigs_clib": (u64, [vp, u64, P(MtgConfig), vp, vp, vp]),
        "mtg_last_performance_data": (None, [P(MtgDijkstraPerformanceData)]),
        "mtg_last_euler_kernel_ms": (C.c_double, []),
        "mtg_set_euler_device_tuning": (None, [C.c_int]),
        "mtg_walks_count": (u64, [vp]),
        "mtg_walks_total_edges": (u64, [vp]),
        "mtg_walks_export": (None, [vp, vp, vp]),
        "mtg_walks_free": (None, [vp]),
        "mtg_walks_data": (None, [vp, P(vp), P(vp)]),
        "mtg_walks_from_arrays": (vp, [u64, vp, vp]),
        "mtg_flatten_clib": (u64, [vp, vp, vp, vp, vp]),
        "mtg_write_walks_fasta": (u64, [vp, u64, vp, vp, u64, C.c_char_p, vp, P(vp)]),
        "mtg_read_bcalm2": (vp, [C.c_char_p, u64, P(vp)]),
        "mtg_unitigs_count": (u64, [vp]),
        "mtg_unitigs_data": (vp, [vp]),
        "mtg_unitigs_offsets": (vp, [vp]),
        "mtg_unitigs_free": (None, [vp]),
        "mtg_write_tigs_fasta_file": (u64, [vp, vp, u64, vp, C.c_char_p, C.c_int]),
        "mtg_write_tigs_text_file_device": (u64, [vp, vp, u64, vp, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
        "mtg_write_walks_text_device": (u64, [vp, u64, vp, vp, u64, C.c_char_p, vp, C.c_int, C.c_char_p, C.c_int, P(vp)]),
        "mtg_last_spell_kernel_ms": (C.c_double, []),
        "mtg_last_spell_bytes": (u64, []),
        "mtg_write_duplication_bitvector": (u64, [vp, u64, vp, vp, P(vp)]),
        "mtg_write_tigs_duplication_bitvector_file": (u64, [vp, vp, C.c_char_p]),
        "mtg_write_walks_gfa": (u64, [vp, u64, vp, vp, u64, C.c_char_p, vp, C.c_char_p, P(vp)]),
        "mtg_write_tigs_gfa_file": (u64, [vp, vp, u64, vp, C.c_char_p, C.c_char_p, C.c_int]),
        "mtg_compute_tigs": (vp, [vp, u64, u64, C.c_int]),
        "mtg_finish_device": (vp, [vp, vp, u64, P(MtgConfig)]),
        "mtg_finish_greedytigs_resident": (vp, [vp, vp, P(MtgConfig)]),
        "mtg_release_device_memory": (None, [C.c_int]),
        "mtg_device_memory_held": (u64, [C.c_int]),
        "mtg_device_arena_stats": (None, [C.c_int, P(u64), C.c_int]),
        "mtg_graph_release_device_cache": (None, [vp]),
        "mtg_set_default_device": (None, [C.c_int]),
        "mtg_set_reserve_ahead": (None, [C.c_int]),
        "mtg_device_create_opts": (vp, [vp, u64, C.c_int, C.c_int]),
        "mtg_device_build_lower_bounds": (None, [vp, vp]),
        "mtg_device_lower_bounds_ms": (C.c_double, [vp]),
        "mtg_set_finish_tuning": (None, [C.c_int, C.c_int, i64]),
        "mtg_replay_claims_resident": (u64, [vp, vp, u64, vp, vp, vp]),
        "mtg_last_replay_ms": (None, [vp, P(C.c_double)]),
        "mtg_set_replay_tuning": (None, [vp, u64, C.c_int, C.c_int, C.c_int, C.c_int]),
        "mtg_resident_pairs": (vp, [vp, P(u64)]),
        "mtg_download_resident_pairs": (u64, [vp, P(P(MtgPair))]),
        "mtg_last_finish_device_times": (None, [P(C.c_double)]),
        "mtg_last_finish_device_stage_ms": (None, [P(C.c_double)]),
        "mtg_synth_g_csr": (vp, [u64, u64, u64, u64, u64, vp, u64, C.c_int, C.c_int]),
        "mtg_last_phase_seconds": (None, [P(C.c_double)]),
        "matchtigs_initialise": (None, []),
        "matchtigs_initialise_graph": (vp, [C.c_size_t]),
        "matchtigs_merge_nodes": (None, [vp, C.c_size_t, C.c_bool, C.c_size_t, C.c_bool]),
        "matchtigs_build_graph": (None, [vp, vp]),
        "matchtigs_compute_tigs": (C.c_size_t, [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_char_p, C.c_char_p, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _ = (i64, i32, u32)
    _lib = L
    return L
