import sys
sys.path.insert(0,'.')
from matchtigs_amd import api, synth, torch_glue
k=31
ua = synth.g_seq_arrays_torch(100000000, seed=1, k=k)
G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links); del ua
dev = api.DeviceGraph(G, k); st = torch_glue.current_stream_ptr(); S = dev.classify(st); bufs = torch_glue.run_sssp(dev, 0, S)
for _ in range(2):
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), st)
    print(dev.last_replay_ms(), dev.last_replay_rounds())
