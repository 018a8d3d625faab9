"""BASELINE configs[2] shape: Eulertigs only (Eulerisation + Euler bicycles + cut, no SSSP / matching) on one MI355X.

usage: python tools/bench_eulertigs.py [--log2-edges 24] [--steps 3]
Prints one JSON line per Euler mode: host = reference-order walk (bit-exact tigs), device = euler_device.hip (same number of
tigs and cumulative length, different order).
"""
import argparse, json, sys, time
sys.path.insert(0, '.')
import numpy as np
from matchtigs_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=24)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
bg = synth.g_csr(int((1 << args.log2_edges) / 1.5 / 2), seed=1, k=args.k)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
L = api._lib.load()
results = {}
for mode, name in ((api.EulerMode.HostReferenceOrder, "host"), (api.EulerMode.Device, "device")):
    times, phases = [], []
    for it in range(args.steps + 1):
        t0 = time.perf_counter()
        lim, ed = api.EulertigAlgorithm.compute_tigs_np(G, api.EulertigAlgorithmConfiguration(args.k, euler_mode=mode))
        t1 = time.perf_counter()
        if it:  # first iteration = warm-up
            times.append(t1 - t0)
            phases.append(api.last_phase_seconds())
        w = G.export()["edge_weight"] if it == args.steps else None
        n_tigs, n_edges = len(lim), len(ed)
        cum = int(w[ed].sum()) + (args.k - 1) * n_tigs if w is not None else None
        G.reset()
    results[name] = (n_tigs, cum)
    print(json.dumps({"workload": f"eulertigs, G-csr |V|={bg.n_nodes} |E|={bg.n_edges} k={args.k}", "euler_mode": name,
                      "seconds_per_graph": round(float(np.mean(times)), 4), "tigs": n_tigs, "tig_edges": n_edges,
                      "cumulative_length": cum,
                      "phases_s": {k: round(float(np.mean([p[k] for p in phases])), 4) for k in ("eulerise", "euler", "cut")},
                      "device_kernels_ms": round(api.last_euler_kernel_ms(), 3) if mode else None}), flush=True)
assert results["host"] == results["device"], results   # same number of tigs and cumulative length in both modes
