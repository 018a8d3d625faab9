"""Times the GPU claim replay alone (candidate lists resident): python tools/replay_probe.py [log2_edges] [repeats] [extra g_csr kwargs as k=v]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from matchtigs_amd import api, synth, torch_glue

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
n_binodes = kw.pop("n_binodes", int((1 << log2) / 1.5 / 2))
k = kw.pop("k", 31)
G = synth.g_csr_device(n_binodes, seed=kw.pop("seed", 1), k=k, **kw)  # (the GPU twin of synth.g_csr: same graph)
dev = api.DeviceGraph(G, k)
st = torch_glue.current_stream_ptr()
S = dev.classify(st)
bufs = torch_glue.run_sssp(dev, 0, S)
print(f"V={G.node_count()} E={G.edge_count()} S={S} candidates={int(bufs.count[:S].sum())}")
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), st)
    dt = time.perf_counter() - t0
    print(f"replay {dt * 1e3:.3f} ms, {len(pairs)} pairs, {dev.last_replay_rounds()} rounds")
