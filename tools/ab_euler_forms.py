"""A/B of the device Euler decomposition's forms in one process (mtg_set_euler_device_tuning flags): kernel time of the whole
decomposition per form, same graph, interleaved. usage: python tools/ab_euler_forms.py [--log2-edges 27] [--flags 0 4] [--reps 4]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from matchtigs_amd import _lib, api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--log2-edges", type=int, default=27)
ap.add_argument("--flags", type=int, nargs="+", default=[0, 4])
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
k = 31
L = _lib.load()
G = synth.g_csr_device(int(round((1 << a.log2_edges) / 3.0)), seed=1, k=k)
dev = api.DeviceGraph(G, k)
dev.classify()
pairs = api.compute_pairs([dev])
del dev
res = {f: [] for f in a.flags}
for r in range(a.reps):
    for f in a.flags:
        L.mtg_set_euler_device_tuning(f)
        lim, ed = api.finish_greedytigs_np(G, pairs, k, euler_mode=api.EulerMode.Device, finish_stage=api.FinishStage.Device)
        res[f].append(round(api.last_euler_kernel_ms(), 3))
        G.reset()
L.mtg_set_euler_device_tuning(0)
print(json.dumps({"log2_edges": a.log2_edges, "decomposition_kernel_ms_by_flags": res}))
