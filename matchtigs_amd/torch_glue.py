"""PyTorch plumbing only: device buffers, streams and torch.distributed for callers that drive the staged
engine API themselves (bench.py, the multi-GPU path, the GPU parity tests). No compute happens here."""
from __future__ import annotations

import numpy as np
import torch

from .api import DeviceGraph


def current_stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class CandidateBuffers:
    """Device buffers for one SSSP call over sources [src_begin, src_end): pool of u64 keys + (start, count)."""

    def __init__(self, n_sources: int, pool_capacity: int, device="cuda"):
        self.n = n_sources
        self.pool = torch.empty(max(pool_capacity, 1), dtype=torch.int64, device=device)
        self.start = torch.empty(max(n_sources, 1), dtype=torch.int64, device=device)
        self.count = torch.empty(max(n_sources, 1), dtype=torch.int32, device=device)
        self.used = 0

    @property
    def capacity(self) -> int:
        return int(self.pool.numel())

    def grow(self, needed: int):
        self.pool = torch.empty(int(needed + needed // 8 + 1024), dtype=torch.int64, device=self.pool.device)


def run_sssp(dev: DeviceGraph, src_begin: int, src_end: int, bufs: CandidateBuffers | None = None,
             initial_keys_per_source: int = 4) -> CandidateBuffers:
    """Calls mtg_sssp_candidates on the current torch stream, growing the pool until it fits."""
    n = src_end - src_begin
    if bufs is None:
        bufs = CandidateBuffers(n, max(1024, initial_keys_per_source * n))
    while True:
        rc, needed = dev.sssp_candidates(src_begin, src_end, bufs.pool.data_ptr(), bufs.capacity, bufs.start.data_ptr(),
                                         bufs.count.data_ptr(), current_stream_ptr())
        if rc == 0:
            bufs.used = needed
            return bufs
        bufs.grow(needed)


def candidates_to_numpy(bufs: CandidateBuffers):
    """Host copies; the start of an empty list (left untouched by the engine) is reported as 0."""
    count = bufs.count[: bufs.n].cpu().numpy().view(np.uint32)
    start = bufs.start[: bufs.n].cpu().numpy().view(np.uint64).copy()
    start[count == 0] = 0
    pool = bufs.pool[: bufs.used].cpu().numpy().view(np.uint64)
    return start, count, pool
