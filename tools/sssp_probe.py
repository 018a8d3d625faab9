"""Kernel-iteration probe: one G-csr (or G-seq) graph, the SSSP stage run REPS times, per-level kernel times (HIP events recorded by
the engine on the launch stream), plus the claim-replay stage once. Development tool: `--lib PATH` loads another build of the
library (an A/B variant built with different flags) instead of matchtigs_amd/libmatchtigs.so.
usage: python tools/sssp_probe.py --log2-edges 24 27 [--reps 5] [--lib matchtigs_amd/libmatchtigs_B.so] [--out FILE]"""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, nargs="+", default=[24])
ap.add_argument("--gseq", type=int, default=0, help="also probe a real de Bruijn graph of this genome length")
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--lib", default=None)
ap.add_argument("--plan", type=int, default=0)
ap.add_argument("--replay", action="store_true")
ap.add_argument("--gap", type=float, default=-1.0, help=">= 0: like bench.py, run the claim replay after every SSSP pass and idle this many seconds")
ap.add_argument("--between", choices=["both", "classify", "replay"], default="both", help="with --gap: what runs between two SSSP passes")
ap.add_argument("--out", default=None)
a = ap.parse_args()
from matchtigs_amd import _lib

if a.lib:
    _lib.LIB_PATH = type(_lib.LIB_PATH)(os.path.abspath(a.lib))
import numpy as np
import torch
from matchtigs_amd import api, synth, torch_glue


def probe(tag, G):
    t = time.time()
    dev = api.DeviceGraph(G, a.k)
    t_dev = time.time() - t
    dev.set_plan(a.plan)
    stream = torch_glue.current_stream_ptr()
    S = dev.classify(stream)
    bufs = None
    runs = []
    for _ in range(a.reps):
        if a.gap >= 0 and bufs is not None and a.between in ("both", "classify"):
            dev.classify(stream)
        bufs = torch_glue.run_sssp(dev, 0, S, bufs)
        runs.append(dev.last_sssp_levels())
        if a.gap >= 0:
            if a.between in ("both", "replay"):
                dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
            time.sleep(a.gap)
    best = min(runs, key=lambda lv: sum(x["ms"] for x in lv))
    res = {"workload": tag, "lib": a.lib or "default", "plan": a.plan, "V": G.node_count(), "E": G.edge_count(), "sources": S, "device_graph_s": round(t_dev, 3),
           "stage_ms_best": round(sum(x["ms"] for x in best), 4), "stage_ms_all": [round(sum(x["ms"] for x in lv), 4) for lv in runs],
           "levels": best, "candidates": int(bufs.count.to(torch.int64).clamp(max=1 << 20).sum().item())}
    cnt = bufs.count[:S].to(torch.int64).clamp(max=40)
    res["count_histogram"] = torch.bincount(cnt, minlength=41).tolist()
    if a.replay:
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.time()
            pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
            times.append(round((time.time() - t) * 1e3, 3))
        res.update(replay_ms=times, pairs=len(pairs), replay_rounds=dev.last_replay_rounds(), replay_visits=dev.last_replay_visits())
    print(json.dumps(res), flush=True)
    if a.out:
        open(a.out, "a").write(json.dumps(res) + "\n")


for lg in a.log2_edges:
    probe(f"g_csr 2^{lg}", synth.g_csr_device(int((1 << lg) / 1.5 / 2), seed=1, k=a.k))
if a.gseq:
    ua = synth.g_seq_arrays_torch(a.gseq, seed=1, k=a.k, haplotypes=4, sub_rate=0.02)
    probe(f"g_seq L={a.gseq}", api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links))
