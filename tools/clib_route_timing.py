"""clib_route_timing.py -- what the reference's OWN way in costs (clib.rs:94-410): matchtigs_initialise_graph, one matchtigs_merge_nodes
per link (here mtg_graph_builder_merge_links: the same function called from one C loop instead of once per ctypes call),
matchtigs_build_graph, then the computation into clib.rs-sized output arrays (mtg_compute_tigs_clib: what matchtigs_compute_tigs runs
once it has built its configuration) -- on a unitig graph with the degree structure of a compacted de Bruijn graph (1.4 node sides per
unitig, every arriving end linked with every leaving end of its node: matchtigs_amd.synth.dbg_like_links).

usage: python tools/clib_route_timing.py [--unitigs 8000000] [--reps 2] [--compute device|host|none] [--k 31]
One JSON line per repetition. The builder itself needs no GPU (--compute none); bench.py runs this as a child process for its
`clib_route` block."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    from matchtigs_amd import _lib, api
    from matchtigs_amd.api import _ptr
    from matchtigs_amd.synth import dbg_like_links

    ap = argparse.ArgumentParser()
    ap.add_argument("--unitigs", type=int, default=8_000_000)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--compute", choices=["device", "host", "none"], default="none")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    U = args.unitigs
    L = _lib.load()
    lk = dbg_like_links(U, seed=args.seed)
    # weights as a compacted de Bruijn graph has them: many short unitigs, a tail of long ones (k-mers per unitig, >= 1)
    w = np.minimum(np.random.default_rng(args.seed + 1).geometric(1.0 / 8.0, U), 10_000).astype(np.uint64)
    for rep in range(args.reps):
        t0 = time.perf_counter()
        h = L.mtg_graph_builder_new(U)
        t1 = time.perf_counter()
        L.mtg_graph_builder_merge_links(h, len(lk), _ptr(lk))
        t2 = time.perf_counter()
        L.mtg_graph_builder_build(h, _ptr(w))
        t3 = time.perf_counter()
        G = api.Bigraph(h)
        out = {"rep": rep, "unitigs": U, "links": int(len(lk)), "nodes": int(G.node_count()), "initialise_s": round(t1 - t0, 4),
               "merge_nodes_s": round(t2 - t1, 4), "merge_nodes_ns_per_link": round((t2 - t1) / max(len(lk), 1) * 1e9, 1),
               "build_graph_s": round(t3 - t2, 4), "build_graph_ns_per_link": round((t3 - t2) / max(len(lk), 1) * 1e9, 1)}
        if args.compute != "none":
            n_edges = 2 * U
            eo, io, lo = np.empty(2 * n_edges, np.int64), np.empty(2 * n_edges, np.uint64), np.empty(n_edges, np.uint64)
            mode = api.EulerMode.Device if args.compute == "device" else api.EulerMode.HostReferenceOrder
            cfg = api.GreedytigAlgorithmConfiguration(1, args.k, euler_mode=mode).to_c()
            t4 = time.perf_counter()
            n = L.mtg_compute_tigs_clib(G.handle, 5, C.byref(cfg), eo.ctypes.data, io.ctypes.data, lo.ctypes.data)
            t5 = time.perf_counter()
            out.update(euler_mode=args.compute, compute_tigs_s=round(t5 - t4, 4), tigs=int(n), tig_edges=int(lo[n - 1]) if n else 0,
                       total_s=round((t3 - t0) + (t5 - t4), 4), phases_s={kk: round(v, 4) for kk, v in api.last_phase_seconds().items()})
        else:
            out["total_s"] = round(t3 - t0, 4)
        del G
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
