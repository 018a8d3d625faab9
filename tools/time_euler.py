"""Host Euler stage (Eulerisation + reference-order walk + cut) on a G-csr graph, repeated; MTG_DEBUG=1 prints the split into record
building and walking. usage: MTG_DEBUG=1 python tools/time_euler.py [log2_edges = 24] [reps = 2]"""
import sys, time; sys.path.insert(0, '.')
from matchtigs_amd import api, synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bg = synth.g_csr(int(2**lg / 1.5 / 2), seed=1, k=31)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
L = api._lib.load()
for rep in range(reps):
    t0 = time.time(); tigs = api._take_walks_np(L, L.mtg_compute_eulertigs(G.handle, 31)); t1 = time.time()
    print("eulertigs total %.3f s" % (t1 - t0), api.last_phase_seconds(), len(tigs[0]), G.edge_count(), flush=True)
    G.reset()
