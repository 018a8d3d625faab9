"""The N>1 path on CPU: world_size 2 and 3 over gloo. Sources are block-partitioned, each rank contributes the
candidate lists of its block (here taken from the oracle, since there is no GPU), ONE all-gather, then the
product's claim replay on the concatenation must equal the oracle's single-process pair list."""
import os
import socket
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    from matchtigs_amd import distributed as mdist
    from matchtigs_amd import synth

    k = 31
    bg = synth.g_csr(3000, seed=5, k=k, mean_out_degree=1.7, mean_weight=5.0)
    arrs = (bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    og = helpers.oracle_graph(*arrs)
    on, off, keys, _ = og.candidate_lists(k)
    S = len(on)
    ranges = mdist.partition_sources(S, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == S and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    lo, hi = ranges[rank]
    # this rank's block, with its pool in a *different* local layout (reversed source order) to prove rebasing works
    cnt = np.diff(off)[lo:hi].astype(np.int32)
    order = np.arange(hi - lo)[::-1]
    local_start = np.zeros(hi - lo, np.int64)
    chunks, pos = [], 0
    for i in order:
        local_start[i] = pos
        chunks.append(keys[int(off[lo + i]):int(off[lo + i + 1])])
        pos += int(cnt[i])
    pool = np.concatenate(chunks) if chunks else np.zeros(0, np.uint64)
    pad = np.full(17 + rank * 5, 0xDEAD, np.uint64)  # capacity larger than used
    pool_t = torch.from_numpy(np.concatenate([pool, pad]).view(np.int64).copy())
    start_all, count_all, pool_all = mdist.allgather_candidates(torch.from_numpy(local_start), torch.from_numpy(cnt), pool_t,
                                                                len(pool), ranges)
    cs, cc, po = mdist.to_numpy_u(start_all, count_all, pool_all)
    assert len(cs) == S and len(cc) == S and int(cc.sum()) == len(po) == len(keys)
    got = np.concatenate([po[int(s):int(s) + int(c)] for s, c in zip(cs, cc)]) if S else np.zeros(0, np.uint64)
    assert np.array_equal(got, keys)
    if rank == 0:
        _, live, mult, _, _ = og.classify()
        G = helpers.product_graph(*arrs)
        pr = G.replay_claims(on, mult.astype(np.int32), live, cs, cc, po)
        want, _ = helpers.oracle_graph(*arrs).greedy_pairs_np(k)
        ok = len(pr) == len(want) and all(np.array_equal(pr[f], want[f]) for f in ("out", "in", "dist"))
        Path(out_dir, "result.txt").write_text("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_then_replay_matches_single_process(world, oracle, product_lib):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        assert Path(d, "result.txt").read_text() == "ok"


def test_partition_sources_properties():
    from matchtigs_amd import distributed as mdist

    for S in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = mdist.partition_sources(S, w)
            assert len(r) == w and r[0][0] == 0 and r[-1][1] == S
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1 and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_partition_sources_by_work_properties():
    from matchtigs_amd import distributed as mdist

    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 1000, 12345):
        work = rng.integers(1, 6, size=n)
        for w in (1, 2, 3, 8):
            r = mdist.partition_sources_by_work(work, w)
            assert len(r) == w and r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1)) and all(lo <= hi for lo, hi in r)
            if n >= 1000:
                loads = [int(work[lo:hi].sum()) for lo, hi in r]
                assert max(loads) - min(loads) <= 10        # equal work up to one source
    skew = np.array([1] * 900 + [100] * 100)                 # the heavy tail goes to its own ranks
    r = mdist.partition_sources_by_work(skew, 4)
    loads = [int(skew[lo:hi].sum()) for lo, hi in r]
    assert max(loads) <= 1.1 * sum(loads) / 4
