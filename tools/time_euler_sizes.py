"""Host Euler walk, ns per biedge at graph sizes from cache-resident to DRAM-resident (MTG_DEBUG=1 prints the phases)."""
import sys, time; sys.path.insert(0, '.')
from matchtigs_amd import api, synth
L = api._lib.load()
for lg in (15, 17, 19, 21, 24):
    bg = synth.g_csr(int(2**lg / 1.5 / 2), seed=1, k=31)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    G.make_eulerian(0, 31)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); w = L.mtg_euler_cycles(G.handle); t1 = time.perf_counter()
        n = int(L.mtg_walks_total_edges(w)); L.mtg_walks_free(w)
        best = min(best, t1 - t0)
    print(f"log2E={lg}: {n} biedges, {best*1e3:.2f} ms, {best/n*1e9:.1f} ns per biedge (records {G.node_count()*256/1e6:.0f} MB)", file=sys.stderr)
