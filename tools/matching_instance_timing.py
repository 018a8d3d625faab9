"""Times the optimal-matchtigs front end (tig algorithm 4 up to the external matcher) on the bench graph:
GPU all-targets searches + candidate download, host collapse into the matching instance, writing the instance file.
usage: python tools/matching_instance_timing.py [--log2-edges 24] [--k 31] [--out FILE]"""
import argparse, json, os, sys, tempfile, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchtigs_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=24)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--out", default=None)
a = ap.parse_args()
bg = synth.g_csr(int((1 << a.log2_edges) / 1.5 / 2), seed=1, k=a.k)
G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
res = {"workload": f"G-csr 2^{a.log2_edges}: V={bg.n_nodes} E={bg.n_edges}", "k": a.k}
for rep in range(2):
    t0 = time.time()
    m = api.MatchingInstance(G, a.k)
    t1 = time.time()
    with tempfile.TemporaryDirectory() as td:
        n = m.write(os.path.join(td, "x.minimalperfectmatching"))
        t2 = time.time()
    res[f"run{rep}"] = {"instance_s": round(t1 - t0, 3), "write_s": round(t2 - t1, 3), "file_bytes": n}
    res["stats"] = m.stats()
    m.close()
s = json.dumps(res)
print(s)
if a.out:
    open(a.out, "w").write(s + "\n")
