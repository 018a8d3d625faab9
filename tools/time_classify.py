"""Where do the milliseconds of the bench's 'classify' phase go? (kernels take 0.3 ms)"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from matchtigs_amd import api, synth, torch_glue
bg = synth.g_csr(int(2**24 / 1.5 / 2), seed=1, k=31)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
dev = api.DeviceGraph(G, 31)
st = torch_glue.current_stream_ptr()
bufs = None
for rep in range(4):
    t0 = time.perf_counter(); S = dev.classify(st); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if bufs is None:
        bufs = torch_glue.CandidateBuffers(S, max(1024, 4 * S))
    bufs = torch_glue.run_sssp(dev, 0, S, bufs)
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), st)
    t3 = time.perf_counter()
    lim, ed = api.finish_greedytigs_np(G, pairs, 31)
    t4 = time.perf_counter()
    G.reset()
    t5 = time.perf_counter()
    del lim, ed, pairs
    t6 = time.perf_counter()
    print("rep %d: classify call %.2f ms, sync %.2f ms, sssp+replay %.1f ms, finish %.0f ms, reset %.1f ms, free results %.1f ms"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
