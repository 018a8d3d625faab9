"""BASELINE.json configs at (or as near as one test run allows to) their stated sizes, on the GPU.

configs[0]  E. coli K-12 BCALM2 unitigs k=31, --greedytigs-fa-out : SURVEY 8d stand-in G-seq(L = 4.6e6, H = 4, p = 0.02, k = 31),
            end to end through the BCALM2 file route; FASTA bytes == oracle, k-mer set preserved.
configs[3]  human whole genome k=31 : G-csr stand-in at |E| = 2^27 (the bench.py default, reference-order finish) AND at SURVEY 8d's
            nominal 2^30, through the size-independent properties of tests/gpu_props.py + GPU/host claim-loop parity + the exact
            sampled comparison with the CPU oracle of tests/sampled_parity.py (classification in full, candidate lists of 10^5
            sources, pair-list prefix).
configs[4]  661k-bacteria pangenome : G-csr stand-in at |E| = 2^31 (node ids at the u32 edge, 2.98 G darts after the finish), GPU
            stages + device finish + properties + the same exact sampled comparison and GPU/host claim-loop parity. The 2^30 / 2^31 cases skip on a box without the HBM / host memory they need.
configs[1]  C. elegans k=31 greedy matchtigs : REAL de Bruijn topology at SURVEY 8d's size, G-seq(L = 10^8) generated on the GPU: pairs equal
            the oracle's exactly, tig properties, exact k-mer set.
configs[2]  human chr1 k=31 Eulertigs : G-seq(L = 2.5 * 10^8), device order: tig properties, exact k-mer set, no repeated k-mer.
(configs[1] / configs[2] on the G-csr stand-in at 2^24: test_gpu_parity.py::test_full_bench_size_properties,
test_gpu_replay.py::test_gpu_replay_full_bench_size and test_gpu_euler.py::test_device_euler_full_bench_size.)"""
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
    # the generator of the larger real-topology cases below (torch on the GPU) builds the same graph
    ut = synth.g_seq_arrays_torch(4_600_000, seed=1, k=k, haplotypes=4, sub_rate=0.02)
    assert np.array_equal(ua.seq, ut.seq) and np.array_equal(ua.off, ut.off) and np.array_equal(ua.links, ut.links) and np.array_equal(ua.kmers, ut.kmers)
    del ut
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


def _fasta_sequences(fa: bytes):
    """(sequence bytes as uint8, offsets) of a FASTA of `>header` + one sequence line per record, at sizes where a Python list of
    records is too slow: line ends by numpy."""
    a = np.frombuffer(fa, np.uint8)
    nl = np.nonzero(a == 10)[0]
    assert len(nl) % 2 == 0 and a[0] == ord(">")
    starts = nl[0::2] + 1  # a sequence line begins after its header's line end
    ends = nl[1::2]
    assert (a[np.r_[0, ends[:-1] + 1]] == ord(">")).all()
    lens = ends - starts
    off = np.zeros(len(lens) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    keep = np.ones(len(a), bool)
    keep[nl] = False
    hdr_len = nl[0::2] - np.r_[0, ends[:-1] + 1]
    # drop the header characters: mark them through a difference array
    d = np.zeros(len(a) + 1, np.int64)
    np.add.at(d, np.r_[0, ends[:-1] + 1], 1)
    np.add.at(d, nl[0::2], -1)
    keep &= np.cumsum(d[:-1]) == 0
    del hdr_len
    return a[keep], off


def test_config1_celegans_like_real_dbg_greedy(gpu, oracle):
    """BASELINE configs[1] (C. elegans k = 31 greedy matchtigs, one GPU) on REAL de Bruijn topology at SURVEY 8d's size: G-seq(L = 10^8,
    H = 4, p = 0.02), 9.6 M unitigs, generated with torch on the GPU (held equal to the numpy generator at 4.6 Mbp above). T1 / T2 exact:
    the GPU's pair list equals the CPU oracle's sequential claim loop with its own truncated Dijkstras; the finish in the reference's
    order passes the tig properties; the spelled tigs hold exactly the input k-mer set, in fewer characters than the unitigs."""
    import gpu_props
    from matchtigs_amd import api, synth, torch_glue

    torch = gpu
    free_hbm, free_host = _free_memory_gb(torch)
    if free_hbm < 60 or free_host < 60:
        pytest.skip(f"needs 60 GB of HBM and of host memory; this box has {free_hbm:.0f} / {free_host:.0f}")
    k = 31
    ua = synth.g_seq_arrays_torch(100_000_000, seed=1, k=k)
    torch.cuda.empty_cache()
    assert ua.n_unitigs > 9_000_000
    G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
    dev = api.DeviceGraph(G, k)
    stream = torch_glue.current_stream_ptr()
    S = dev.classify(stream)
    bufs = torch_glue.run_sssp(dev, 0, S)
    pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
    on, mu, li = dev.classify_download()
    n_cand = gpu_props.check_candidates(torch, bufs, on, li, k)
    del bufs, dev
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    want, st = og.greedy_pairs_np(k)
    del og
    assert len(pairs) == len(want) and all(np.array_equal(pairs[f], want[f]) for f in ("out", "in", "dist"))
    lim, edges = api.finish_greedytigs_np(G, pairs, k, api.EulerMode.HostReferenceOrder)
    cum, dummy_kmers = gpu_props.check_tigs(torch, G, lim, edges, k)
    fa = api.write_walks_text_device(G, (lim, edges), (ua.seq, ua.off), k)
    seq, off = _fasta_sequences(fa)
    del fa
    codes, n_occ = synth.kmer_codes_of_sequences_torch(seq, off, k)
    assert np.array_equal(codes, ua.kmers)
    assert len(off) - 1 == len(lim) < ua.n_unitigs and int(off[-1]) < int(ua.off[-1])
    print(f"config1 (real dBG, 10^8 bp): {ua.n_unitigs} unitigs, {len(ua.kmers)} k-mers, S={S}, candidates={n_cand}, pairs={len(pairs)} "
          f"(oracle queries {st['queries']}), tigs={len(lim)}, {int(ua.off[-1])} -> {int(off[-1])} characters, k-mer occurrences {n_occ}")


def test_config2_chr1_like_real_dbg_eulertigs(gpu):
    """BASELINE configs[2] (human chr1 k = 31 Eulertigs: the Euler-walk kernels only, no SSSP / matching) on REAL de Bruijn topology at
    SURVEY 8d's size: G-seq(L = 2.5 * 10^8), device order (the tigs cut straight from the pairing). Tig properties, the exact input
    k-mer set, and no k-mer spelled twice (an Euler tiling repeats nothing)."""
    import gpu_props
    from matchtigs_amd import api, synth

    torch = gpu
    free_hbm, free_host = _free_memory_gb(torch)
    if free_hbm < 120 or free_host < 100:
        pytest.skip(f"needs 120 GB of HBM and 100 GB of host memory; this box has {free_hbm:.0f} / {free_host:.0f}")
    k = 31
    ua = synth.g_seq_arrays_torch(250_000_000, seed=1, k=k)
    torch.cuda.empty_cache()
    G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
    lim, edges = api.EulertigAlgorithm.compute_tigs_np(G, api.EulertigAlgorithmConfiguration(k, euler_mode=api.EulerMode.Device))
    cum, dummy_kmers = gpu_props.check_tigs(torch, G, lim, edges, k)
    assert dummy_kmers == 0
    fa = api.write_walks_text_device(G, (lim, edges), (ua.seq, ua.off), k)
    seq, off = _fasta_sequences(fa)
    del fa
    codes, n_occ = synth.kmer_codes_of_sequences_torch(seq, off, k)
    assert np.array_equal(codes, ua.kmers) and n_occ == len(ua.kmers)
    assert len(off) - 1 == len(lim) < ua.n_unitigs
    print(f"config2 (real dBG, 2.5 * 10^8 bp): {ua.n_unitigs} unitigs, {len(ua.kmers)} k-mers -> {len(lim)} eulertigs, {int(ua.off[-1])} -> {int(off[-1])} characters")


def _free_memory_gb(torch):
    """(free HBM, host memory this process may still use) in GB; the host side honours a cgroup limit if there is one."""
    gc.collect()
    torch.cuda.empty_cache()
    from matchtigs_amd import api

    api.release_device_memory(0)  # (what the library kept of an earlier test's finish: it would come back on demand, but it is not "free")
    free_hbm = torch.cuda.mem_get_info()[0] / 1e9
    avail = None
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            avail = int(line.split()[1]) * 1024 / 1e9
    try:
        limit = Path("/sys/fs/cgroup/memory.max").read_text().strip()
        stat = dict(line.split() for line in Path("/sys/fs/cgroup/memory.stat").read_text().splitlines())
        used = int(stat.get("anon", 0)) + int(stat.get("shmem", 0))  # (page cache is reclaimable and does not count)
        if limit != "max":
            avail = min(avail, (int(limit) - used) / 1e9)
    except (OSError, ValueError):
        pass
    return free_hbm, avail


def _stand_in(gpu, oracle, log2_edges, hbm_gb, host_gb, host_replay_check, euler_mode):
    """One G-csr stand-in through the whole HIP path: generated on the GPU, device graph, classification, SSSP, GPU claim replay,
    finish (insertion + Euleriser + Euler bicycles + cut on the GPU), every stage checked by tests/gpu_props.py -- and, exactly,
    by tests/sampled_parity.py: the classification in full, the candidate lists of 10^5 sampled sources (the first 20 000, the last
    2 000 -- the highest node ids and block offsets of the graph -- and 80 000 drawn in between) and the pair list's prefix of the
    first 20 000 sources against the CPU oracle's own searches and claim loop on the subgraph those sources can reach."""
    import gpu_props
    import sampled_parity
    from matchtigs_amd import api, synth, torch_glue

    torch = gpu
    free_hbm, free_host = _free_memory_gb(torch)
    if free_hbm < hbm_gb or free_host < host_gb:
        pytest.skip(f"2^{log2_edges} stand-in needs {hbm_gb} GB of free HBM and {host_gb} GB of host memory; this box has {free_hbm:.0f} / {free_host:.0f}")
    k = 31
    G = synth.g_csr_device(int((1 << log2_edges) / 1.5 / 2), seed=1, k=k)
    n_orig, V = G.edge_count(), G.node_count()
    dev = api.DeviceGraph(G, k)
    stream = torch_glue.current_stream_ptr()
    S = dev.classify(stream)
    bufs = torch_glue.run_sssp(dev, 0, S)
    levels = dev.last_sssp_levels()
    gpu_pairs = dev.replay_claims_device(bufs.start.data_ptr(), bufs.count.data_ptr(), bufs.pool.data_ptr(), stream)
    rounds = dev.last_replay_rounds()
    on, mu, li = dev.classify_download()
    n_cand = gpu_props.check_candidates(torch, bufs, on, li, k)
    if host_replay_check:  # T2 at this size: the GPU claim loop equals the host claim loop on the same lists
        start, count, pool = torch_glue.candidates_to_numpy(bufs)
        host_pairs = G.replay_claims(on, mu, li, start, count, pool)
        assert len(gpu_pairs) == len(host_pairs) and all(np.array_equal(gpu_pairs[f], host_pairs[f]) for f in ("out", "in", "dist"))
        del host_pairs, start, count, pool
    del dev
    gc.collect()
    api.release_device_memory(0)  # (the device graph's memory goes back to the driver: the check below works with torch tensors)
    sp = sampled_parity.check_sampled_lists_and_prefix(torch, oracle, G, bufs, on, mu, li, gpu_pairs, k)
    assert sp["sampled"] >= min(S, 100000) and sp["prefix_pairs"] > 0
    del bufs, on, mu, li
    gc.collect()
    torch.cuda.empty_cache()
    lim, edges = api.finish_greedytigs_np(G, gpu_pairs, k, euler_mode)
    E_total = G.edge_count()
    cum, dummy_kmers = gpu_props.check_tigs(torch, G, lim, edges, k)
    print(f"2^{log2_edges} stand-in: V={V} E={n_orig} S={S} candidates={n_cand} pairs={len(gpu_pairs)} replay rounds={rounds} "
          f"darts after the finish={E_total} tigs={len(lim)} cumulative length={cum}; SSSP levels {[(l['sources'], round(l['ms'], 2)) for l in levels]}")
    return dict(V=V, E=n_orig, tigs=len(lim), cum=cum, E_total=E_total, pairs=len(gpu_pairs))


def test_config3_human_like_2pow27_properties(gpu, oracle):
    """configs[3] at the bench's size, finish in the reference's walk order (the default)."""
    from matchtigs_amd import api

    r = _stand_in(gpu, oracle, 27, 40, 40, True, api.EulerMode.HostReferenceOrder)
    assert r["E"] == 130045206 and r["tigs"] == 23687715 and r["cum"] == 1294322592


def test_config3_human_like_2pow30_full_size(gpu, oracle):
    """BASELINE configs[3] (human whole genome, k = 31) at SURVEY 8d's nominal size: |V| = 716 M, |E| = 1.04 G, 270 M sources on
    ONE GPU; GPU-vs-host claim parity at that size; the finish in device Euler mode (the reference-order walk at this size takes
    150 s: tools/scale_probe.py --euler host, profiles/r03_scale_probe_2p30_exact.json -- same tig count and cumulative length)."""
    from matchtigs_amd import api

    r = _stand_in(gpu, oracle, 30, 170, 140, True, api.EulerMode.Device)
    assert r["E"] == 1040327706 and r["tigs"] == 189468154 and r["cum"] == 10353599420


def test_config4_pangenome_like_2pow31(gpu, oracle):
    """BASELINE configs[4] (661k-bacteria pangenome, ~10^9 unitigs) as its G-csr stand-in: |V| = 1.43 G, |E| = 2.08 G, 539 M
    sources, node ids up to 1.43e9 and 2.98 G darts after the finish (beyond 2^31: the dart ids use all 32 bits) on ONE GPU."""
    from matchtigs_amd import api

    r = _stand_in(gpu, oracle, 31, 262, 190, True, api.EulerMode.Device)
    assert r["E"] == 2080660578 and r["E_total"] == 2984426052 and r["tigs"] == 378964208


