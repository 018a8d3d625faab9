"""Sweep of the claim replay's admission windows (number, and from which window on they grow by which factor): rounds-kernel time by
HIP events, rounds, source visits; the pair list must not change. python tools/replay_window_sweep.py [log2_edges=27 | gseq:LENGTH] [first_n_configs]
(gseq:100000000 = a REAL compacted de Bruijn graph of that many bases, synth.g_seq_arrays through the clib.rs builder)"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
import torch
from matchtigs_amd import api, synth, torch_glue

arg = sys.argv[1] if len(sys.argv) > 1 else "27"
k = 31
if arg.startswith("gseq:"):
    ua = synth.g_seq_arrays_torch(int(arg[5:]), seed=1, k=k)
    G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
    del ua
else:
    G = synth.g_csr_device(int((1 << int(arg)) / 1.5 / 2), seed=1, k=k)
dev = api.DeviceGraph(G, k)
st = torch_glue.current_stream_ptr()
S = dev.classify(st)
bufs = torch_glue.run_sssp(dev, 0, S)
ref = None
configs = [(0, 0, 0), (1 << 16, 0, 0), (36, 8, 2), (8, 0, 0), (12, 0, 0), (16, 0, 0), (16, 8, 2), (20, 0, 0), (24, 0, 0), (48, 0, 0), (40, 0, 0), (36, 0, 0), (32, 0, 0), (40, 5, 2), (36, 5, 2), (36, 8, 2), (32, 4, 2), (32, 8, 2), (30, 5, 3), (28, 4, 3), (24, 4, 3), (24, 8, 2)]
if len(sys.argv) > 2:
    configs = configs[:int(sys.argv[2])]  # (only the first few: the default list, the same adaptive, ...)
for n, g16, mul in configs:
    enc = (1 << 32) if n == (1 << 16) else n | (g16 << 16) | (mul << 24)  # (n = 65536 stands for: the default list, adaptive)
    dev.set_replay_tuning(windows=enc)
    ms = []
    for _ in range(4):
        pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), st)
        ms.append(dev.last_replay_ms())
    crc = zlib.crc32(pairs.tobytes())
    if ref is None:
        ref = crc
    best = min(x["rounds_kernel_ms"] for x in ms[1:])
    stage = min(x["gpu_ms"] for x in ms[1:])
    print(f"windows {n:3d} grow from {g16:2d}/16 x{mul}: rounds kernel {best:.3f} ms, stage {stage:.3f} ms, {dev.last_replay_rounds()} rounds, "
          f"{dev.last_replay_visits()} visits, pairs {'same' if crc == ref else 'DIFFERENT'}", flush=True)
