#!/usr/bin/env python3
"""Times the finishing stages (insertion + Euleriser + Euler bicycles + cut) of one G-csr graph generated on the GPU,
host stages vs device finish, both Euler modes.  python tools/finish_probe.py --log2-edges 27 [--modes host,device] [--reps 2]"""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None, help="another build of the library (A/B of a development variant)")
ap.add_argument("--log2-edges", type=int, default=24)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--stages", default="device,host")
ap.add_argument("--eulers", default="device,exact")
ap.add_argument("--gseq", type=int, default=0, help="a compacted de Bruijn graph of a random genome of this length (G-seq, 4 haplotypes) instead of G-csr")
ap.add_argument("--no-cut-first", action="store_true", help="device Euler mode through the closed walks (A/B of cut_first_device.hip)")
args = ap.parse_args()
if args.lib:
    from matchtigs_amd import _lib
    _lib.LIB_PATH = type(_lib.LIB_PATH)(str(Path(args.lib).resolve()))
from matchtigs_amd import api, synth
k = args.k
nb = int(round((1 << args.log2_edges) / 3.0))
t0 = time.perf_counter()
if args.gseq:
    ua = synth.g_seq_arrays_torch(args.gseq, seed=args.seed, k=k)
    G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
    del ua
else:
    G = synth.g_csr_device(nb, seed=args.seed, k=k)
if args.no_cut_first:
    api.set_finish_tuning(no_cut_first=True)
t1 = time.perf_counter()
out = {"log2_edges": args.log2_edges, "V": G.node_count(), "E": G.edge_count(), "generate_s": round(t1 - t0, 3)}
dev = api.DeviceGraph(G, k)
S = dev.classify()
t2 = time.perf_counter()
pairs = api.compute_pairs([dev])
t3 = time.perf_counter()
del dev
out.update(sources=int(S), pairs=len(pairs), device_graph_s=round(t2 - t1, 3), pairs_s=round(t3 - t2, 3))
print(json.dumps(out), flush=True)
for stage in args.stages.split(","):
    for eu in args.eulers.split(","):
        fs = api.FinishStage.Device if stage == "device" else api.FinishStage.Host
        em = api.EulerMode.Device if eu == "device" else api.EulerMode.HostReferenceOrder
        for r in range(args.reps):
            t = time.perf_counter()
            lim, ed = api.finish_greedytigs_np(G, pairs, k, euler_mode=em, finish_stage=fs)
            tf = time.perf_counter() - t
            n_tigs, n_edges, Etot = len(lim), len(ed), G.edge_count()
            t = time.perf_counter()
            G.reset()
            tr = time.perf_counter() - t
            rec = {"stage": stage, "euler": eu, "rep": r, "finish_s": round(tf, 4), "reset_s": round(tr, 4), "tigs": n_tigs, "tig_edges": n_edges, "E_total": Etot}
            if stage == "device":
                rec.update({kk: round(v, 4) if isinstance(v, float) else v for kk, v in api.last_finish_device_times().items()})
            else:
                rec.update({kk: round(v, 4) for kk, v in api.last_phase_seconds().items()} if hasattr(api, "last_phase_seconds") else {})
            print(json.dumps(rec), flush=True)
            del lim, ed
