"""SURVEY 8e inside the library (mtg_compute_pairs / mtg_config.device_ids): sources block-partitioned by work over several
resident copies of the graph, peer-copy gather on the first, claim replay there. The copies go to DISTINCT GPUs when the box has
them (then the gather really runs hipMemcpyPeerAsync over xGMI); a 1-GPU box can only check the FUNCTION (every copy on GPU 0).
The pair list and the tigs must be identical to the single-device path and to the oracle. Also here: bench.py's one-process-per-
GPU path with two ranks (gloo, both on GPU 0 of this box)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_ids(lib, n):
    """n device ids: distinct GPUs as far as the box has them, GPU 0 again beyond."""
    have = lib.mtg_device_count()
    return tuple(i if i < have else 0 for i in range(n))


def _pairs_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("out", "in", "dist"))


@pytest.mark.parametrize("n_dev", [2, 3])
def test_compute_pairs_over_several_device_copies(n_dev, oracle, product_lib):
    from matchtigs_amd import api, synth

    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    bg = synth.g_csr(40000, seed=5, k=31, mean_out_degree=1.7)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    devs = [api.DeviceGraph(G, bg.k, i) for i in _device_ids(product_lib, n_dev)]
    S = [d.classify() for d in devs]
    assert len(set(S)) == 1
    cuts = api.partition_sources(devs[0], n_dev)
    assert cuts[0] == 0 and cuts[-1] == S[0] and all(a <= b for a, b in zip(cuts, cuts[1:]))
    # blocks of near-equal estimated work (1 + out-degree of the source)
    on, _, _ = devs[0].classify_download()
    work = 1 + np.bincount(bg.edge_from, minlength=bg.n_nodes)[on]
    per_block = [int(work[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
    assert max(per_block) - min(per_block) <= 8
    got = api.compute_pairs(devs)
    one = api.compute_pairs(devs[:1])
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).greedy_pairs_np(bg.k)
    assert _pairs_equal(got, one) and _pairs_equal(got, want)


def test_compute_tigs_cfg_with_device_ids(oracle, product_lib):
    """mtg_compute_tigs_cfg with n_devices = 2 (distinct GPUs if the box has two, else both ids 0): tigs equal the oracle's."""
    from matchtigs_amd import api, synth

    bg = synth.g_csr(20000, seed=9, k=31)
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration(1, bg.k, device_ids=_device_ids(product_lib, 2)))
    want, _ = oracle.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight).compute_greedytigs(bg.k)
    assert tigs == want


def test_bench_two_ranks_one_process_per_gpu(product_lib):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process): candidate exchange at exact
    sizes, claim replay + finish on rank 0, the non-zero rank without a host graph. On a 1-GPU box both ranks share GPU 0 (gloo)."""
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    root = Path(__file__).resolve().parents[1]
    two = product_lib.mtg_device_count() >= 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-edges", "19",
           "--no-cpu-baseline"] + ([] if two else ["--backend", "gloo", "--single-device"])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(root), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    one = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--log2-edges", "19",
                          "--no-cpu-baseline", "--extra-seeds", ""], capture_output=True, text=True, cwd=str(root), timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "sources/2"
    for key in ("pairs", "tigs", "sources", "candidates", "V", "E"):  # the same graph, the same result for every N
        assert d["config"][key] == d1["config"][key], key
    assert d["units_per_step"]["relaxed_edges"] == d1["units_per_step"]["relaxed_edges"]
    assert d["scaling_stages_ms"]["sssp_stage"] > 0 and d["scaling_stages_ms"]["allgather"] > 0
    assert d["device_mode"]["tigs"] == d["config"]["tigs"]
