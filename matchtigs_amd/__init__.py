"""matchtigs_amd -- MI355X-native engine for the greedy-matchtigs / eulertigs hot path of algbio/matchtigs.

The product is the C-ABI library ``matchtigs_amd/libmatchtigs.so`` (HIP kernels + C++ host stages; headers in
``include/``). This package is the thin Python mirror of the reference's operator interface over it.
"""
from .api import (  # noqa: F401
    Bigraph,
    DeviceGraph,
    EulertigAlgorithm,
    EulertigAlgorithmConfiguration,
    GreedytigAlgorithm,
    GreedytigAlgorithmConfiguration,
    HeapType,
    MatchtigEdgeData,
    NodeWeightArrayType,
    PerformanceDataType,
    TigAlgorithm,
    clib_compute_tigs,
    last_phase_seconds,
)

__all__ = [n for n in dir() if not n.startswith("_")]
