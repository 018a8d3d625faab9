"""The library's device arena (csrc/hip_util.hpp: DeviceArena) under a random allocation / free load, compiled from
tests/tools/arena_test.hip with hipcc on the GPU box: ranges never overlap, keep their content, coalesce back into whole chunks."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_arena_random_load(product_lib, tmp_path, seed):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "arena_test"
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", str(ROOT / "matchtigs_amd" / "csrc"),
                        str(ROOT / "tests" / "tools" / "arena_test.hip"), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), str(seed), "3000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "arena_test ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_reserved_step_makes_no_driver_allocation(product_lib):
    """mtg_device_create_opts(MTG_DEVICE_RESERVE_WORK): the stages of a step -- classification, search, claim replay with the pairs
    resident, the finish on the GPU -- then find their arrays in the arena; the driver is not asked for memory inside the step, the
    first time or the second (a cold staged step must not pay for allocations a warm one does not make)."""
    import torch

    from matchtigs_amd import api, synth, torch_glue

    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    k = 31
    api.release_device_memory(0)
    G = synth.g_csr_device(int((1 << 23) / 1.5 / 2), seed=5, k=k)
    dev = api.DeviceGraph(G, k, 0, reserve_work=True)
    stream = torch_glue.current_stream_ptr()
    before = api.device_arena_stats(0, reset_peak=True)
    counts = []
    for _ in range(2):
        S = dev.classify(stream)
        bufs = torch_glue.run_sssp(dev, 0, S)  # (candidate buffers are the caller's: torch tensors, not the arena's)
        n_pairs = dev.replay_claims_resident(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
        tigs = api.finish_greedytigs_resident(G, dev, k, api.EulerMode.Device, 0, api.FinishStage.Auto)
        counts.append((S, n_pairs, tigs.count(), tigs.total_edges()))
        del tigs, bufs
        torch.cuda.synchronize()
        G.reset()
    after = api.device_arena_stats(0)
    assert counts[0] == counts[1] and counts[0][2] > 0
    assert after["driver_allocations"] == before["driver_allocations"], (before, after)
    assert after["peak_bytes"] - before["live_bytes"] <= product_lib_step_estimate(G)


def product_lib_step_estimate(G):
    V, E = G.node_count(), G.original_edge_count()
    return (V * 64 + E * 62) // 100 * 108 + (64 << 20)


@pytest.mark.gpu
def test_reserve_ahead_can_be_turned_off(product_lib):
    """mtg_set_reserve_ahead(0): a host-only graph constructor starts no helper thread and takes no GPU memory (round-5 advice: a pure host
    API must not have GPU side effects its caller did not opt into); with the default, the constructor's helper thread reserves the
    call's chunk. A process of its own: the arena starts empty."""
    import subprocess
    import sys

    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "from matchtigs_amd import api, synth\n"
        "bg = synth.g_csr(2_000_000, seed=3, k=31)\n"
        "api.set_reserve_ahead(False)\n"
        "G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)\n"
        "time.sleep(1.0)\n"
        "off = api.device_arena_stats(0)\n"
        "api.set_reserve_ahead(True)\n"
        "G2 = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)\n"
        "for _ in range(100):\n"
        "    on = api.device_arena_stats(0)\n"
        "    if on['chunk_bytes']: break\n"
        "    time.sleep(0.1)\n"
        "tigs = api.GreedytigAlgorithm.compute_tigs_np(G2, api.GreedytigAlgorithmConfiguration.new(1, 31))\n"
        "print('RESULT', off['chunk_bytes'], off['driver_allocations'], on['chunk_bytes'], len(tigs[0]))\n") % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    off_bytes, off_allocs, on_bytes, n_tigs = (int(x) for x in [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:])
    assert off_bytes == 0 and off_allocs == 0
    assert on_bytes >= (256 << 20) and n_tigs > 0
