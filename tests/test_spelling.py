"""Tig spelling (bin.rs:466-606): the product's mtg_write_walks_fasta against the oracle's C restatement and
the independent Python restatement, byte for byte; plus the k-mer-set semantic check. The eulertig path and the
spelling itself need no GPU; the greedy end-to-end FASTA check is marked gpu."""
import numpy as np
import pytest

import pyref
from matchtigs_amd import api, synth


def _fasta_seqs(fa: str):
    return [l for l in fa.split("\n") if l and not l.startswith(">")]


@pytest.mark.parametrize("seed,k,length", [(1, 11, 1500), (2, 15, 3000), (5, 31, 5000)])
def test_eulertig_fasta_product_equals_oracle_and_pyref(seed, k, length, oracle, product_lib):
    ug = synth.g_seq(length, seed=seed, k=k, haplotypes=3, sub_rate=0.03)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    tigs = api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k))
    fa = api.write_walks_fasta(G, tigs, ug.unitigs, k).decode()
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    otigs = og.compute_eulertigs(k)
    assert otigs == tigs
    assert og.fasta(otigs, ug.unitigs, k) == fa
    pg = pyref.from_unitig_links([int(x) for x in ug.weights], ug.links)
    assert pyref.fasta(pg, pyref.compute_eulertigs(pg, k), ug.unitigs, k) == fa
    assert synth.kmer_set_of_tigs(_fasta_seqs(fa), k) == ug.kmers
    # numpy (limits, edges) form gives the same bytes
    lim = np.cumsum([len(t) for t in tigs]).astype(np.uint64)
    ed = np.array([e for t in tigs for e in t], dtype=np.uint32)
    assert api.write_walks_fasta(G, (lim, ed), ug.unitigs, k).decode() == fa


def test_greedy_fasta_from_oracle_walks_spelled_by_product(oracle, product_lib):
    """Greedy tigs contain matched dummy edges (shortened overlaps k-1-weight): spell the ORACLE's walks with the PRODUCT's
    speller on the product's identically mutated graph (host stages only, no GPU)."""
    import helpers

    k = 15
    ug = synth.g_seq(4000, seed=3, k=k, haplotypes=4, sub_rate=0.03)
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    pr = helpers.product_pairs_from_oracle_lists(G, oracle.OracleGraph.from_unitig_links(ug.weights, ug.links), k)
    tigs = G.finish_greedytigs(pr, k)
    otigs, _ = og.compute_greedytigs(k)
    assert tigs == otigs
    assert any(e >= 2 * len(ug.unitigs) for t in tigs for e in t), "expected at least one kept dummy edge"
    fa = api.write_walks_fasta(G, tigs, ug.unitigs, k).decode()
    assert fa == og.fasta(otigs, ug.unitigs, k)
    assert synth.kmer_set_of_tigs(_fasta_seqs(fa), k) == ug.kmers
    assert fa.count(">") == len(tigs) and fa.startswith(">1\n")


@pytest.mark.gpu
def test_greedy_fasta_end_to_end_on_gpu(oracle, product_lib):
    """Unitig links in -> GPU greedy matchtigs -> FASTA out, bit-identical to the oracle's FASTA (T4 + spelling)."""
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    k = 21
    ug = synth.g_seq(6000, seed=9, k=k, haplotypes=4, sub_rate=0.02)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration.new(1, k))
    fa = api.write_walks_fasta(G, tigs, ug.unitigs, k).decode()
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    otigs, _ = og.compute_greedytigs(k)
    assert fa == og.fasta(otigs, ug.unitigs, k)
    assert synth.kmer_set_of_tigs(_fasta_seqs(fa), k) == ug.kmers
    # duplication bitvector (implementation/mod.rs:668-702): one character per spelled k-mer, '0' exactly where a matched
    # dummy edge repeats k-mers; the '1's add up to the number of distinct k-mers of the input
    bits = api.write_duplication_bitvector(G, tigs).decode().splitlines()
    assert [len(b) for b in bits] == [len(s) - k + 1 for s in _fasta_seqs(fa)]
    assert sum(b.count("1") for b in bits) == len(ug.kmers)
    ex = G.export()
    kept_dummy_weight = sum(int(ex["edge_weight"][e]) for t in tigs for e in t if ex["edge_dummy_id"][e])
    assert sum(b.count("0") for b in bits) == kept_dummy_weight and kept_dummy_weight > 0


def test_gfa_records_are_the_fasta_records(product_lib):
    """bin.rs:667-818: GFA output = header line + one S record per tig with the same spelling as the FASTA writer."""
    k = 15
    ug = synth.g_seq(3000, seed=4, k=k, haplotypes=3, sub_rate=0.03)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    tigs = api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k))
    fa = api.write_walks_fasta(G, tigs, ug.unitigs, k).decode()
    seqs = _fasta_seqs(fa)
    want = "H\tKL:Z:15\n" + "".join(f"S\t{i + 1}\t{s}\n" for i, s in enumerate(seqs))
    assert api.write_walks_gfa(G, tigs, ug.unitigs, k).decode() == want
    custom = api.write_walks_gfa(G, tigs, ug.unitigs, k, header="H\tVN:Z:1.0\tKL:Z:15").decode()
    assert custom == "H\tVN:Z:1.0\tKL:Z:15\n" + want.split("\n", 1)[1]


def test_duplication_bitvector(product_lib):
    """implementation/mod.rs:668-702: per tig, weight x '1' for an original edge and weight x '0' for a dummy edge."""
    k = 15
    ug = synth.g_seq(3000, seed=4, k=k, haplotypes=3, sub_rate=0.03)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    tigs = api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k))
    ex = G.export()
    want = "".join("".join(("0" if ex["edge_dummy_id"][e] else "1") * int(ex["edge_weight"][e]) for e in t) + "\n" for t in tigs)
    got = api.write_duplication_bitvector(G, tigs).decode()
    assert got == want
    # eulertigs contain no dummy edges (every dummy is a breaking edge and is cut): all ones, one per spelled k-mer
    assert set(got) <= {"1", "\n"}
    fa = api.write_walks_fasta(G, tigs, ug.unitigs, k).decode()
    assert [len(l) for l in got.splitlines()] == [len(s) - k + 1 for s in _fasta_seqs(fa)]


def test_duplication_bitvector_of_greedy_tigs_against_the_oracle(oracle, product_lib):
    """The '0' branch by bytes, against data the product did not make: greedy matchtigs keep matched dummy edges, whose k-mers are
    duplicates. Expected text = implementation/mod.rs:668-702 evaluated over the ORACLE's graph (its edge weights and dummy ids after its
    own insertion + Eulerisation) and the oracle's tigs; the product writes the same bytes from its own graph and tigs."""
    import helpers

    k = 15
    ug = synth.g_seq(4000, seed=3, k=k, haplotypes=4, sub_rate=0.03)
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    pr = helpers.product_pairs_from_oracle_lists(G, oracle.OracleGraph.from_unitig_links(ug.weights, ug.links), k)
    tigs = G.finish_greedytigs(pr, k)
    otigs, _ = og.compute_greedytigs(k)
    oe = og.edges()  # (from, to, weight, dummy id, handle, forwards) of the oracle's graph after its own finish
    want = "".join("".join(("1" if oe[e][3] == 0 else "0") * int(oe[e][2]) for e in t) + "\n" for t in otigs)
    got = api.write_duplication_bitvector(G, tigs).decode()
    assert got == want
    assert got.count("0") > 0 and got.count("1") == len(ug.kmers)


@pytest.mark.gpu
def test_device_spelling_is_byte_identical(oracle, product_lib):
    """SURVEY f-1 on the device (spell_device.hip): FASTA and GFA bytes equal the host speller's and the oracle's, for greedy
    matchtigs (dummy overlaps), eulertigs and plain unitigs, on real tiny dBGs incl. backwards edges."""
    from matchtigs_amd import api, synth

    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    for seed, k in ((3, 11), (5, 21), (8, 31)):
        ug = synth.g_seq(6000, seed=seed, k=k, haplotypes=4, sub_rate=0.03)
        for alg in ("greedy", "euler", "unitigs"):
            G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
            if alg == "greedy":
                tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration.new(1, k))
            elif alg == "euler":
                tigs = api.EulertigAlgorithm.compute_tigs(G, api.EulertigAlgorithmConfiguration(k))
            else:
                tigs = [[2 * u] for u in range(len(ug.unitigs))] + [[2 * u + 1] for u in range(0, len(ug.unitigs), 7)]
            host = api.write_walks_fasta(G, tigs, ug.unitigs, k)
            dev = api.write_walks_text_device(G, tigs, ug.unitigs, k)
            assert dev == host, (seed, k, alg)
            assert api.write_walks_text_device(G, tigs, ug.unitigs, k, gfa=True) == api.write_walks_gfa(G, tigs, ug.unitigs, k)
            assert api.write_walks_text_device(G, tigs, ug.unitigs, k, gfa=True, header="H\tVN:Z:1.0") == \
                api.write_walks_gfa(G, tigs, ug.unitigs, k, header="H\tVN:Z:1.0")
    # empty input and lower-case input
    G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
    assert api.write_walks_text_device(G, [], ug.unitigs, k) == b""
    assert api.write_walks_text_device(G, [[0]], [u.lower() for u in ug.unitigs], k) == api.write_walks_fasta(G, [[0]], ug.unitigs, k)


@pytest.mark.gpu
def test_device_spelling_at_scale(product_lib):
    """E. coli-sized real dBG (config[0] stand-in): greedy matchtigs spelled on the GPU == host speller; prints the kernel's HBM rate."""
    import numpy as np
    from matchtigs_amd import api, synth

    k = 31
    ua = synth.g_seq_arrays(1_000_000, seed=3, k=k, haplotypes=4, sub_rate=0.02)
    G = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
    lim, ed = api._take_walks_np(api._lib.load(), api._lib.load().mtg_compute_tigs(G.handle, 5, k, 0))
    units = (ua.seq, ua.off)
    dev = api.write_walks_text_device(G, (lim, ed), units, k)
    info = api.last_spell_kernel()
    host = api.write_walks_fasta(G, (lim, ed), ua.unitig_list(), k)
    assert dev == host
    print(f"device spelling: {len(dev)} bytes, kernel {info['ms']:.3f} ms, {info['bytes'] / max(info['ms'], 1e-9) / 1e6:.1f} GB/s")
