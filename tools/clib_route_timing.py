"""clib_route_timing.py -- what the reference's own way in costs: matchtigs_initialise_graph + one matchtigs_merge_nodes per link
(here: mtg_graph_builder_merge_links, the same calls from one loop in C) + matchtigs_build_graph, on a unitig graph with the degree
structure of a compacted de Bruijn graph (1.4 binode sides per unitig, every arriving end linked with every leaving end of its
node). Host only; no GPU needed for the builder itself.  usage: python tools/clib_route_timing.py [unitigs=8000000] [reps=2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    from matchtigs_amd import _lib
    from matchtigs_amd.synth import dbg_like_links
    from matchtigs_amd.api import _ptr

    U = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = _lib.load()
    lk = dbg_like_links(U)
    w = np.random.default_rng(2).integers(1, 40, U).astype(np.uint64)
    print(f"{U} unitigs, {len(lk)} links")
    for _ in range(reps):
        t0 = time.time()
        h = L.mtg_graph_builder_new(U)
        t1 = time.time()
        L.mtg_graph_builder_merge_links(h, len(lk), _ptr(lk))
        t2 = time.time()
        L.mtg_graph_builder_build(h, _ptr(w))
        t3 = time.time()
        print(f"initialise {t1 - t0:.3f} s, merge_nodes x {len(lk)} {t2 - t1:.3f} s ({(t2 - t1) / len(lk) * 1e9:.1f} ns each), "
              f"build_graph {t3 - t2:.3f} s ({(t3 - t2) / U * 1e9:.0f} ns per unitig)")
        L.mtg_graph_free(h)


if __name__ == "__main__":
    main()
