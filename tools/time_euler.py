import sys, time; sys.path.insert(0,'.')
import numpy as np
from matchtigs_amd import api, synth
bg = synth.g_csr(int(2**24/1.5/2), seed=1, k=31)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
for rep in range(2):
    t0=time.time(); tigs = api._take_walks_np(api._lib.load(), api._lib.load().mtg_compute_eulertigs(G.handle, 31)); t1=time.time()
    print("eulertigs total %.3f s"%(t1-t0), api.last_phase_seconds(), len(tigs[0]), G.edge_count())
    G.reset()
