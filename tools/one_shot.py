"""The consuming one-shot call as a caller of the reference's C-ABI sees it (clib.rs:280-291 consumes the graph: every real call is a
first call): host edge arrays -> mtg_graph_from_edges -> mtg_compute_tigs_clib (= what matchtigs_compute_tigs runs once it has built
its configuration: the whole path into clib.rs-sized output arrays) -> graph freed.

Device memory: the library's arena is NOT released between the calls (--release-between does, for comparison: every release adds
driver calls -- the frees, then fresh allocations -- to the next call, and single driver calls sporadically stall for 0.5-5 s on
this pool's shared hosts: tools/alloc_probe.hip; a fresh hipMalloc normally costs 0.3 ms whatever its size, which is
what a first call in a fresh process pays).

The input arrays come from the GPU generator (a graph is generated, exported to numpy arrays, freed): the generator is NOT timed.
`--calls 2` repeats the call on the same arrays: call 0 is the FIRST call of the process (the HIP runtime's queues, the pinned
transfer ring and the kernels' code objects come into being inside it), call 1 a later first-call-on-a-graph.

usage: python tools/one_shot.py [--log2-edges 27] [--euler device|host] [--calls 2]
Prints one JSON line per call. bench.py runs this as a child process for its `one_shot` block."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from matchtigs_amd import _lib, api, synth  # noqa: E402


def one_call(L, arrays, k: int, mode) -> dict:
    mirror, ef, et, ew = arrays
    n_edges = len(ef)
    # caller-allocated outputs sized as clib.rs:332-348 (untouched memory, like a real caller's)
    eo, io, lo = np.empty(2 * n_edges, np.int64), np.empty(2 * n_edges, np.uint64), np.empty(n_edges, np.uint64)
    t0 = time.perf_counter()
    G = api.Bigraph.from_edges(mirror, ef, et, ew)
    t1 = time.perf_counter()
    cfg = api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode).to_c()
    n = L.mtg_compute_tigs_clib(G.handle, 5, C.byref(cfg), eo.ctypes.data, io.ctypes.data, lo.ctypes.data)
    t2 = time.perf_counter()
    del G
    t3 = time.perf_counter()
    ph = api.last_phase_seconds()
    n_e = int(lo[n - 1]) if n else 0
    # phases of the compute call: device graph (upload + build kernels + lower bounds), classification, search + claim replay,
    # insertion + Euleriser, Euler bicycles, rotate + cut + the flattened tigs into the caller's arrays
    return {"graph_build_s": round(t1 - t0, 4), "compute_tigs_clib_s": round(t2 - t1, 4), "graph_free_s": round(t3 - t2, 4),
            "total_s": round(t2 - t0, 4), "total_with_free_s": round(t3 - t0, 4), "tigs": int(n), "tig_edges": n_e,
            "checksum": int(np.bitwise_xor.reduce(eo[:n_e].view(np.uint64)) ^ np.bitwise_xor.reduce(io[:n_e]) ^ np.bitwise_xor.reduce(lo[:n])) if n else 0,
            "phases_s": {kk: round(v, 4) for kk, v in ph.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-edges", type=int, default=27)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--euler", choices=["host", "device"], default="device")
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--release-between", action="store_true", help="mtg_release_device_memory (and a second of rest) before every call")
    args = ap.parse_args()
    L = _lib.load()
    k = args.k
    mode = api.EulerMode.Device if args.euler == "device" else api.EulerMode.HostReferenceOrder
    g = synth.g_csr_device(int((1 << args.log2_edges) / 1.5 / 2), seed=args.seed, k=k, device_id=args.device)
    V, E = g.node_count(), g.edge_count()
    mirror = g.export_mirror()
    ex = g.export_range(0, E, ("edge_from", "edge_to", "edge_weight"))
    arrays = (mirror, ex["edge_from"], ex["edge_to"], ex["edge_weight"])
    del g, ex
    for i in range(args.calls):
        if args.release_between:
            api.release_device_memory(args.device)
            time.sleep(1.0)
        r = one_call(L, arrays, k, mode)
        r.update(call=i, euler_mode=args.euler, V=int(V), E=int(E), log2_edges=args.log2_edges)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
