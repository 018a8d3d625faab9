"""BASELINE.json configs at (or as near as one test run allows to) their stated sizes, on the GPU.

configs[0]  E. coli K-12 BCALM2 unitigs k=31, --greedytigs-fa-out : SURVEY 8d stand-in G-seq(L = 4.6e6, H = 4, p = 0.02, k = 31),
            end to end through the BCALM2 file route; FASTA bytes == oracle, k-mer set preserved.
configs[3]  human whole genome k=31 : G-csr stand-in at |E| = 2^27 (the bench.py default; 2^30 does not fit a test run's time
            budget), through the size-independent properties of test_full_bench_size_properties + GPU/host claim-loop parity.
configs[4]  661k-bacteria pangenome (|E| = 2^31, node ids at the u32 edge): not run -- the device graph alone is 128-256 GB.
configs[1] / configs[2] are covered at their sizes by test_gpu_parity.py::test_full_bench_size_properties,
test_gpu_replay.py::test_gpu_replay_full_bench_size and test_gpu_euler.py::test_device_euler_full_bench_size."""
import gc
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def gpu(product_lib):
    import torch

    if product_lib.mtg_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("these tests need a GPU: the matchtigs_amd hot path has no CPU fallback")
    return torch


def _fasta_records(text: bytes):
    """(concatenated sequence bytes as uint8, offsets) of a FASTA whose records are one header + one sequence line."""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 2 == 0 and all(l.startswith(b">") for l in lines[0::2][:1000])
    seqs = lines[1::2]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8), off


def test_config0_ecoli_like_bcalm2_to_greedytigs_fasta(gpu, oracle, tmp_path):
    from matchtigs_amd import synth

    k = 31
    ua = synth.g_seq_arrays(4_600_000, seed=1, k=k, haplotypes=4, sub_rate=0.02)
    assert ua.n_unitigs > 400_000
    inp, out = tmp_path / "unitigs.fa", tmp_path / "greedy.fa"
    inp.write_bytes(ua.bcalm2_text())
    r = subprocess.run([sys.executable, "-m", "matchtigs_amd", "--bcalm-in", str(inp), "-k", str(k), "--greedytigs-fa-out", str(out)],
                       capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    fa = out.read_bytes()
    # T4 at this size: FASTA bytes equal the oracle's for the same unitig links
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    tigs, st = og.compute_greedytigs(k)
    want = og.fasta(tigs, ua.unitig_list(), k).encode()
    assert len(fa) == len(want) and fa == want
    # semantic: the spelled tigs contain exactly the input k-mer set; greedy matchtigs are shorter than the unitigs
    seq, off = _fasta_records(fa)
    assert np.array_equal(synth.kmer_codes_of_sequences(seq, off, k), ua.kmers)
    assert len(off) - 1 == len(tigs) < ua.n_unitigs
    assert int(off[-1]) < int(ua.off[-1])
    print(f"config0: {ua.n_unitigs} unitigs, {len(ua.kmers)} k-mers -> {len(tigs)} greedy matchtigs, "
          f"{int(ua.off[-1])} -> {int(off[-1])} characters; oracle queries {st['queries']}")


def test_config3_human_like_2pow27_properties(gpu):
    from matchtigs_amd import api, synth, torch_glue

    k = 31
    bg = synth.g_csr(int((1 << 27) / 1.5 / 2), seed=1, k=k)
    n_orig, V = bg.n_edges, bg.n_nodes
    unitig_kmers = int(bg.edge_weight[0::2].sum())
    G = api.Bigraph.from_edges(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    del bg
    gc.collect()
    dev = api.DeviceGraph(G, k)
    stream = torch_glue.current_stream_ptr()
    S = dev.classify(stream)
    bufs = torch_glue.run_sssp(dev, 0, S)
    gpu_pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
    start, count, pool = torch_glue.candidates_to_numpy(bufs)
    on, mu, li = dev.classify_download()
    # candidate lists: strictly ascending keys per source, targets only, bound respected, source excluded
    cnt64 = count.astype(np.int64)
    tot = int(cnt64.sum())
    seg_begin = np.cumsum(cnt64) - cnt64
    idx = np.repeat(start.astype(np.int64), cnt64) + (np.arange(tot, dtype=np.int64) - np.repeat(seg_begin, cnt64))
    keys = pool[idx]
    del idx
    seg_first = np.zeros(tot, bool)
    seg_first[seg_begin[cnt64 > 0]] = True
    assert (np.diff(keys.astype(np.int64))[~seg_first[1:]] > 0).all()
    nodes, dist = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64), (keys >> np.uint64(32)).astype(np.int64)
    assert li[nodes].all() and dist.min() >= 1 and dist.max() <= k - 1
    assert (nodes != np.repeat(on.astype(np.int64), cnt64)).all()
    del keys, nodes, dist, seg_first
    # T2 at this size: the GPU claim loop equals the host claim loop on the same lists
    host_pairs = G.replay_claims(on, mu, li, start, count, pool)
    assert len(gpu_pairs) == len(host_pairs) and all(np.array_equal(gpu_pairs[f], host_pairs[f]) for f in ("out", "in", "dist"))
    del host_pairs, start, count, pool, bufs
    gc.collect()
    lim, edges = api.finish_greedytigs_np(G, gpu_pairs, k)
    ex = G.export()
    orig = edges[edges < n_orig]
    assert len(orig) == n_orig // 2
    seen = np.zeros(n_orig // 2, np.uint8)
    seen[orig >> 1] = 1
    assert seen.all()                                                    # every unitig exactly once, in one orientation
    starts = np.r_[0, lim[:-1]].astype(np.int64)
    assert (edges[starts] < n_orig).all() and (edges[lim.astype(np.int64) - 1] < n_orig).all()
    w = ex["edge_weight"][edges[edges >= n_orig]]
    assert (w >= 1).all() and (w <= k - 1).all()                         # only matched dummies survive inside tigs
    outd = np.bincount(ex["edge_from"], minlength=V)
    ind = np.bincount(ex["edge_to"], minlength=V)
    sm = ex["mirror"] == np.arange(V)
    assert (outd[~sm] == ind[~sm]).all() and (outd[sm] % 2 == 0).all()   # Eulerian after Eulerisation
    cum = int(ex["edge_weight"][edges].sum()) + (k - 1) * len(lim)
    assert cum == unitig_kmers + int(w.sum()) + (k - 1) * len(lim)       # cumulative-length identity
    print(f"config3 stand-in: V={V} E={n_orig} S={S} pairs={len(gpu_pairs)} tigs={len(lim)} replay rounds={dev.last_replay_rounds()}")
