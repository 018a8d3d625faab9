"""Scale probe beyond the bench default (maximum sizes): G-csr graph of 2^N nominal edges, generated on the GPU,
through the HIP path (device graph, classify, SSSP, claim replay) and the finish, checked by the size-independent properties of
tests/gpu_props.py (the same checks tests/test_gpu_configs.py runs).
usage: python tools/scale_probe.py --log2-edges 30 [--euler device|host] [--host-replay-check] [--out FILE]
--euler host = the reference-order walk over 32-byte GPU-built records (23 GB at 2^30); --euler device = everything on the GPU."""
import argparse, gc, json, os, resource, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import gpu_props
from matchtigs_amd import api, synth, torch_glue

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=29)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--euler", choices=["host", "device"], default="device")
ap.add_argument("--host-replay-check", action="store_true", help="also run the host claim loop on the same lists and compare")
ap.add_argument("--out", default=None)
ap.add_argument("--no-finish", action="store_true", help="stop after the GPU stages")
ap.add_argument("--rss-limit-gb", type=float, default=270.0, help="watchdog: leave (exit code 3) before the box runs out of memory")
a = ap.parse_args()
T0 = time.time()


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
res = {"log2_edges": a.log2_edges, "k": k, "seed": a.seed, "euler_mode": a.euler, "generator": "synth.g_csr_device"}


def lap(name, t0):
    res[name + "_s"] = round(time.time() - t0, 3)
    res["peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)
    free, total = torch.cuda.mem_get_info()
    res["peak_hbm_gb"] = max(res.get("peak_hbm_gb", 0), round((total - free) / 1e9, 1))
    print(f"[{time.time() - T0:7.1f}s] {name}: {res[name + '_s']} s, peak RSS {res['peak_rss_gb']} GB, HBM in use {(total - free) >> 20} MiB", flush=True)


t = time.time()
G = synth.g_csr_device(int((1 << a.log2_edges) / 1.5 / 2), seed=a.seed, k=k)
n_orig, V = G.edge_count(), G.node_count()
res.update(V=V, E=n_orig)
lap("generate_and_host_graph", t)
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
on, mu, li = dev.classify_download()
res["candidates"] = gpu_props.check_candidates(torch, bufs, on, li, k)
lap("candidate_properties", t)
if a.host_replay_check:
    t = time.time()
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    host_pairs = G.replay_claims(on, mu, li, start, count, pool)
    assert len(pairs) == len(host_pairs) and all(np.array_equal(pairs[f], host_pairs[f]) for f in ("out", "in", "dist"))
    del host_pairs, start, count, pool
    lap("host_replay_equal", t)
del bufs, on, mu, li, dev
gc.collect()
torch.cuda.empty_cache()
if not a.no_finish:
    t = time.time()
    mode = api.EulerMode.Device if a.euler == "device" else api.EulerMode.HostReferenceOrder
    lim, edges = api.finish_greedytigs_np(G, pairs, k, mode)
    res["tigs"] = len(lim)
    res["E_with_dummies"] = G.edge_count()
    res["finish_device_s"] = {n: (round(v, 3) if isinstance(v, float) else v) for n, v in api.last_finish_device_times().items()}
    lap("finish", t)
    t = time.time()
    res["cumulative_length"], res["matched_dummy_kmers_in_tigs"] = gpu_props.check_tigs(torch, G, lim, edges, k)
    lap("tig_properties", t)
res["total_s"] = round(time.time() - T0, 1)
s = json.dumps(res)
print(s)
if a.out:
    open(a.out, "w").write(s + "\n")
