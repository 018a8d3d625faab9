"""Semantic end-to-end on tiny real de Bruijn graphs (CPU): the tigs spell exactly the input k-mer set,
FASTA spelling follows bin.rs:466-606, and cumulative lengths order as unitigs >= eulertigs >= greedy matchtigs."""
import pytest

import pyref
from matchtigs_amd import synth


@pytest.mark.parametrize("seed,k,length", [(1, 11, 1200), (2, 15, 3000), (3, 21, 2500), (4, 31, 4000)])
def test_kmer_set_preserved_and_fasta_matches_pyref(seed, k, length, oracle):
    ug = synth.g_seq(length, seed=seed, k=k, haplotypes=3, sub_rate=0.03)
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    tigs, _ = og.compute_greedytigs(k)
    fa = og.fasta(tigs, ug.unitigs, k)
    seqs = [l for l in fa.split("\n") if l and not l.startswith(">")]
    assert fa.startswith(">1\n") and fa.endswith("\n") and len(seqs) == len(tigs)
    assert synth.kmer_set_of_tigs(seqs, k) == ug.kmers
    # every tig k-mer count adds up: no k-mer is lost, repeats only inside kept dummy overlaps
    pg = pyref.from_unitig_links([int(x) for x in ug.weights], ug.links)
    ptigs, _, _ = pyref.compute_greedytigs(pg, k)
    assert ptigs == tigs and pyref.fasta(pg, ptigs, ug.unitigs, k) == fa
    og2 = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    et = og2.compute_eulertigs(k)
    fe = og2.fasta(et, ug.unitigs, k)
    eseqs = [l for l in fe.split("\n") if l and not l.startswith(">")]
    assert synth.kmer_set_of_tigs(eseqs, k) == ug.kmers
    unitig_len = sum(map(len, ug.unitigs))
    assert unitig_len >= sum(map(len, eseqs)) >= sum(map(len, seqs))
    # an Euler tiling repeats nothing: total k-mers in eulertigs == distinct k-mers
    assert sum(len(s) - k + 1 for s in eseqs) == len(ug.kmers)


def test_g_seq_arrays_equals_g_seq():
    """The vectorised generator (used for the E. coli-sized GPU test and the real-dBG bench workload) builds exactly
    g_seq's unitigs, unitig order, orientations, links and link order."""
    from matchtigs_amd import synth

    for (L, seed, k, H, p) in [(3000, 7, 15, 4, 0.02), (6000, 3, 15, 4, 0.03), (8000, 5, 31, 4, 0.02), (6000, 9, 21, 3, 0.03)]:
        a = synth.g_seq(L, seed=seed, k=k, haplotypes=H, sub_rate=p)
        b = synth.g_seq_arrays(L, seed=seed, k=k, haplotypes=H, sub_rate=p)
        assert b.unitig_list() == a.unitigs
        assert [(int(x[0]), bool(x[1]), int(x[2]), bool(x[3])) for x in b.links] == a.links
        assert synth.unitig_graph_of_arrays(b).kmers == a.kmers
        assert (b.weights == a.weights).all()


def test_g_seq_arrays_torch_equals_g_seq_arrays():
    """The torch form of the generator (on a GPU: the C. elegans-like and chr1-like sizes of SURVEY 8d in seconds) builds exactly what
    the numpy form builds -- here with torch on the CPU; tests/test_gpu_configs.py holds the two equal at 4.6 Mbp on the GPU."""
    import numpy as np
    from matchtigs_amd import synth

    for (L, seed, k, H, p) in [(30000, 1, 31, 4, 0.02), (120000, 5, 21, 3, 0.05), (5000, 2, 15, 2, 0.1), (60000, 9, 31, 1, 0.0)]:
        a = synth.g_seq_arrays(L, seed=seed, k=k, haplotypes=H, sub_rate=p)
        b = synth.g_seq_arrays_torch(L, seed=seed, k=k, haplotypes=H, sub_rate=p, device="cpu")
        assert np.array_equal(a.seq, b.seq) and np.array_equal(a.off, b.off) and np.array_equal(a.links, b.links) and np.array_equal(a.kmers, b.kmers)
        codes, n = synth.kmer_codes_of_sequences_torch(a.seq, a.off, k, "cpu")
        assert np.array_equal(codes, a.kmers) and n == len(a.kmers)  # (unitigs repeat no k-mer)


def test_real_dbg_eulertigs_at_scale_cpu(tmp_path, oracle, product_lib):
    """BCALM2 file route + eulertigs (host-only path) on a 10^5-bp genome: FASTA bytes equal the oracle's, k-mer set preserved."""
    import numpy as np
    from matchtigs_amd import api, synth

    k = 31
    ua = synth.g_seq_arrays(100_000, seed=5, k=k, haplotypes=4, sub_rate=0.02)
    inp, out = tmp_path / "u.fa", tmp_path / "e.fa"
    inp.write_bytes(ua.bcalm2_text())
    G, store = api.read_bcalm2(str(inp), k)
    res = api.compute_tigs_to_fasta_file(G, store, 3, k, str(out))
    og = oracle.OracleGraph.from_unitig_links_arrays(ua.weights, ua.links)
    want = og.fasta(og.compute_eulertigs(k), ua.unitig_list(), k).encode()
    fa = out.read_bytes()
    assert fa == want and res["tigs"] > 0
    lines = fa.split(b"\n")[1::2]
    off = np.zeros(len(lines) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in lines])
    assert np.array_equal(synth.kmer_codes_of_sequences(np.frombuffer(b"".join(lines), np.uint8), off, k), ua.kmers)
