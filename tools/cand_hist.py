"""Histogram of candidate-list lengths (what the claim replay's long-list paths see): python tools/cand_hist.py [log2_edges]"""
import sys
sys.path.insert(0, '.')
import torch
from matchtigs_amd import api, synth, torch_glue

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G = synth.g_csr_device(int((1 << log2) / 1.5 / 2), seed=1, k=31)
dev = api.DeviceGraph(G, 31)
S = dev.classify(torch_glue.current_stream_ptr())
bufs = torch_glue.run_sssp(dev, 0, S)
c = bufs.count[:S].to(torch.int64)
print(f"S={S} candidates={int(c.sum())} max={int(c.max())}")
for lo in (1, 2, 4, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049):
    m = c >= lo
    print(f"  lists with >= {lo:5d} entries: {int(m.sum()):9d}  holding {int(c[m].sum()):10d} candidates")
