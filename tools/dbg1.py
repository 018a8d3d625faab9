import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import oracle_lib
from matchtigs_amd import api, synth, torch_glue
bg = synth.g_csr(30000, seed=1, k=31)
og = oracle_lib.OracleGraph.from_arrays(bg.mirror,bg.edge_from,bg.edge_to,bg.edge_weight)
o_on, off, keys, st = og.candidate_lists(31)
want = np.diff(off).astype(np.int64)
for preset in (0,1):
  for trial in range(3):
    G = api.Bigraph.from_edges(bg.mirror,bg.edge_from,bg.edge_to,bg.edge_weight)
    dev = api.DeviceGraph(G, 31); dev.set_preset(preset)
    S = dev.classify()
    bufs = torch_glue.run_sssp(dev, 0, S)
    start,count,pool = torch_glue.candidates_to_numpy(bufs)
    bad = np.nonzero(count.astype(np.int64)!=want)[0]
    print("preset",preset,"trial",trial,"S",S,"bad",len(bad), bad[:10], count[bad[:10]], want[bad[:10]], "used",bufs.used,"cap",bufs.capacity)
    cnt = dev.sssp_count(0,S); print(cnt, st)
