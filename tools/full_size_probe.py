"""bench.py's full_size block alone (one cold device-mode step at G-csr 2^N and the same step once more), with the library's
MTG_DEBUG laps on stderr: where a cold step's time goes.   usage: MTG_DEBUG=1 python tools/full_size_probe.py [log2_edges]"""
import json, os, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.cuda.set_device(0)
args = types.SimpleNamespace(full_size_log2=lg, seed=1, plan=0)
print(json.dumps(bench.full_size_step(args, 31, 0)))
