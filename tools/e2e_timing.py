"""End-to-end wall clock of the drop-in path on ONE GPU, from host arrays to tig arrays (what a clib.rs caller waits for):
graph construction (mtg_graph_from_edges) + mtg_compute_tigs_cfg (device graph build, classification, SSSP, claim replay,
dummy insertion, Eulerisation, Euler walk in the reference's order, cut) + flattening into the clib.rs output arrays.
The synthetic generator itself is NOT part of it. Optionally (--g-seq L) a REAL de Bruijn graph incl. BCALM2 parsing and
FASTA spelling on the GPU.

usage: python tools/e2e_timing.py [--log2-edges 27] [--euler host|device] [--g-seq LENGTH]"""
import argparse, json, sys, time, tempfile, os
sys.path.insert(0, '.')
import numpy as np
from matchtigs_amd import api, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=27)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--euler", choices=["host", "device"], default="host")
ap.add_argument("--g-seq", type=int, default=0)
args = ap.parse_args()
k = args.k
mode = api.EulerMode.Device if args.euler == "device" else api.EulerMode.HostReferenceOrder
L = _lib.load()
out = {"k": k, "euler_mode": args.euler}
if args.g_seq:
    ua = synth.g_seq_arrays_torch(args.g_seq, seed=1, k=k)
    d = tempfile.mkdtemp()
    inp, fa = os.path.join(d, "unitigs.fa"), os.path.join(d, "tigs.fa")
    open(inp, "wb").write(ua.bcalm2_text())
    t0 = time.perf_counter()
    G, store = api.read_bcalm2(inp, k)
    t1 = time.perf_counter()
    res = api.compute_tigs_to_fasta_file(G, store, 5, k, fa, configuration=api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode))
    t2 = time.perf_counter()
    out.update(workload=f"G-seq L={args.g_seq}: {ua.n_unitigs} unitigs, {len(ua.kmers)} k-mers", read_bcalm2_s=round(t1 - t0, 3),
               compute_s=round(res["compute_s"], 3), spell_and_write_s=round(res["write_s"], 3), total_s=round(t2 - t0, 3),
               tigs=res["tigs"], fasta_bytes=res["fasta_bytes"], phases=api.last_phase_seconds(), spell_kernel=api.last_spell_kernel())
else:
    bg = synth.g_csr(int((1 << args.log2_edges) / 1.5 / 2), seed=1, k=k)
    t0 = time.perf_counter()
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    t1 = time.perf_counter()
    cfg = api.GreedytigAlgorithmConfiguration(1, k, euler_mode=mode).to_c()
    import ctypes as C
    w = L.mtg_compute_tigs_cfg(G.handle, 5, C.byref(cfg))
    t2 = time.perf_counter()
    ec = bg.n_edges
    eo, io, lo = np.zeros(2 * ec, np.int64), np.zeros(2 * ec, np.uint64), np.zeros(ec, np.uint64)
    n = L.mtg_flatten_clib(G.handle, w, eo.ctypes.data, io.ctypes.data, lo.ctypes.data)
    t3 = time.perf_counter()
    out.update(workload=f"G-csr 2^{args.log2_edges}: V={bg.n_nodes} E={bg.n_edges}", graph_from_edges_s=round(t1 - t0, 3),
               compute_tigs_s=round(t2 - t1, 3), flatten_clib_s=round(t3 - t2, 3), total_s=round(t3 - t0, 3), tigs=int(n),
               phases={kk: round(v, 3) for kk, v in api.last_phase_seconds().items()})
print(json.dumps(out))
