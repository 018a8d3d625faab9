import sys, time; sys.path.insert(0, '.')
import torch
from matchtigs_amd import api, synth, torch_glue
bg = synth.g_csr(int(2**24 / 1.5 / 2), seed=1, k=31)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
dev = api.DeviceGraph(G, 31)
st = torch_glue.current_stream_ptr()
for rep in range(6):
    if rep == 3:
        time.sleep(1.5)   # idle GPU, like during the host Euler walk
    t0 = time.perf_counter(); S = dev.classify(st); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("classify %d: %.3f ms" % (rep, (t1 - t0) * 1e3))
