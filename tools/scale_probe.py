"""Scale probe beyond the bench default (maximum sizes): G-csr graph of 2^N nominal edges through the HIP path
(device graph, classify, SSSP, claim replay) and the finish, checked by the size-independent properties of
tests/test_gpu_configs.py::test_config3. usage: python tools/scale_probe.py --log2-edges 30 [--euler device|host] [--out FILE]
The exact host Euler walk needs 256 bytes of records per node (183 GB at 2^30); --euler device keeps the host side small."""
import argparse, gc, json, os, resource, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from matchtigs_amd import api, synth, torch_glue

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=29)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--euler", choices=["host", "device"], default="device")
ap.add_argument("--host-replay-check", action="store_true", help="also run the host claim loop on the same lists and compare")
ap.add_argument("--out", default=None)
ap.add_argument("--no-finish", action="store_true", help="stop after the GPU stages (2^31: the edge count with dummies exceeds the device Euler mode's 2^31 limit and the exact host walk's records exceed the box's memory)")
ap.add_argument("--rss-limit-gb", type=float, default=260.0, help="watchdog: leave (exit code 3) before the box runs out of memory")
a = ap.parse_args()
T0 = time.time()
import threading


def _rss_watchdog():
    beat = time.time()
    while True:
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                rss = int(line.split()[1]) / 1e6
                if rss > a.rss_limit_gb:
                    print(f"RSS above {a.rss_limit_gb} GB: giving up", flush=True)
                    os._exit(3)
                if time.time() - beat > 60:  # (a silent run is taken to be hung)
                    beat = time.time()
                    print(f"[{time.time() - T0:7.1f}s] ... working, RSS {rss:.1f} GB", flush=True)
        time.sleep(0.5)


threading.Thread(target=_rss_watchdog, daemon=True).start()
k = a.k
res = {"log2_edges": a.log2_edges, "k": k, "euler_mode": a.euler}
T0 = time.time()


def lap(name, t0):
    res[name + "_s"] = round(time.time() - t0, 3)
    res["peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)
    print(f"[{time.time() - T0:7.1f}s] {name}: {res[name + '_s']} s, peak RSS {res['peak_rss_gb']} GB, "
          f"HBM in use {torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0] >> 20} MiB", flush=True)


t = time.time()
bg = synth.g_csr(int((1 << a.log2_edges) / 1.5 / 2), seed=1, k=k)
n_orig, V = bg.n_edges, bg.n_nodes
unitig_kmers = int(bg.edge_weight[0::2].sum())
res.update(V=V, E=n_orig)
lap("generate", t)
t = time.time()
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
del bg
gc.collect()
lap("graph_from_edges", t)
t = time.time()
dev = api.DeviceGraph(G, k)
res["device_graph_bytes"] = dev.graph_bytes()
lap("device_graph", t)
stream = torch_glue.current_stream_ptr()
t = time.time()
S = dev.classify(stream)
res["sources"] = S
lap("classify", t)
t = time.time()
bufs = torch_glue.run_sssp(dev, 0, S)
res["sssp_kernel_ms"] = round(dev.last_sssp_kernel_ms(), 3)
res["sssp_levels"] = dev.last_sssp_levels()
lap("sssp_incl_alloc", t)
t = time.time()
pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
res.update(pairs=len(pairs), replay_rounds=dev.last_replay_rounds())
lap("replay", t)
t = time.time()
start, count, pool = torch_glue.candidates_to_numpy(bufs)
on, mu, li = dev.classify_download()
cnt64 = count.astype(np.int64)
tot = int(cnt64.sum())
res["candidates"] = tot
seg_begin = np.cumsum(cnt64) - cnt64
idx = np.repeat(start.astype(np.int64), cnt64) + (np.arange(tot, dtype=np.int64) - np.repeat(seg_begin, cnt64))
keys = pool[idx]
del idx
seg_first = np.zeros(tot, bool)
seg_first[seg_begin[cnt64 > 0]] = True
assert (np.diff(keys.astype(np.int64))[~seg_first[1:]] > 0).all(), "candidate keys not strictly ascending per source"
nodes, dist = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64), (keys >> np.uint64(32)).astype(np.int64)
assert li[nodes].all() and dist.min() >= 1 and dist.max() <= k - 1
assert (nodes != np.repeat(on.astype(np.int64), cnt64)).all()
del keys, nodes, dist, seg_first
lap("candidate_properties", t)
if a.host_replay_check:
    t = time.time()
    host_pairs = G.replay_claims(on, mu, li, start, count, pool)
    assert len(pairs) == len(host_pairs) and all(np.array_equal(pairs[f], host_pairs[f]) for f in ("out", "in", "dist"))
    del host_pairs
    lap("host_replay_equal", t)
del start, count, pool, bufs
gc.collect()
torch.cuda.empty_cache()
if a.no_finish:
    res["total_s"] = round(time.time() - T0, 1)
    s = json.dumps(res)
    print(s)
    if a.out:
        open(a.out, "w").write(s + "\n")
    sys.exit(0)
t = time.time()
mode = api.EulerMode.Device if a.euler == "device" else api.EulerMode.HostReferenceOrder
lim, edges = api.finish_greedytigs_np(G, pairs, k, mode)
res["tigs"] = len(lim)
res["finish_phases_s"] = {n: round(v, 3) for n, v in api.last_phase_seconds().items() if v}
lap("finish", t)
t = time.time()
ex = G.export()
orig = edges[edges < n_orig]
assert len(orig) == n_orig // 2
seen = np.zeros(n_orig // 2, np.uint8)
seen[orig >> 1] = 1
assert seen.all(), "a unitig is missing from the tigs"
starts = np.r_[0, lim[:-1]].astype(np.int64)
assert (edges[starts] < n_orig).all() and (edges[lim.astype(np.int64) - 1] < n_orig).all()
w = ex["edge_weight"][edges[edges >= n_orig]]
assert (w >= 1).all() and (w <= k - 1).all()
outd = np.bincount(ex["edge_from"], minlength=V)
ind = np.bincount(ex["edge_to"], minlength=V)
sm = ex["mirror"] == np.arange(V)
assert (outd[~sm] == ind[~sm]).all() and (outd[sm] % 2 == 0).all()
cum = int(ex["edge_weight"][edges].sum()) + (k - 1) * len(lim)
assert cum == unitig_kmers + int(w.sum()) + (k - 1) * len(lim)
res["cumulative_length"] = cum
lap("tig_properties", t)
res["total_s"] = round(time.time() - T0, 1)
s = json.dumps(res)
print(s)
if a.out:
    open(a.out, "w").write(s + "\n")
