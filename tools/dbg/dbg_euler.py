import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from matchtigs_amd import api, synth
bg = synth.g_csr(300, seed=3, k=5, mean_weight=2.0, self_mirror_frac=0.05)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
dev = api.DeviceGraph(G, bg.k); dev.classify(); pairs = api.compute_pairs([dev]); del dev
print("pairs", len(pairs), flush=True)
H = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
api.finish_greedytigs_np(H, pairs, bg.k, finish_stage=api.FinishStage.Host)
print("host finish ok; wrapper path on the host-finished graph", flush=True)
lim, ed = H.euler_cycles_device_np()
print("ok", len(lim), len(ed), flush=True)
G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
print("finish path", flush=True)
lim, ed = api.finish_greedytigs_np(G2, pairs, bg.k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
print("ok", len(lim), len(ed), flush=True)
