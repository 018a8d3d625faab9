"""BCALM2 reader + the CLI-mirroring driver (`python -m matchtigs_amd --bcalm-in ... -k ...`).

The reader must build exactly the graph the clib.rs construction gives for the same links (the in-tree specification of
that construction); the written FASTA must equal the oracle's for the same graph and contain exactly the input k-mer set.
Eulertigs need no GPU; the greedy route is marked gpu."""
import gzip
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from matchtigs_amd import api, synth

ROOT = Path(__file__).resolve().parent.parent


def write_bcalm2(path, ug: synth.UnitigGraph, gz=False, shuffle_ids=False):
    by_src = {}
    for (a, sa, b, sb) in ug.links:
        by_src.setdefault(a, []).append((sa, b, sb))
    lines = []
    for i, u in enumerate(ug.unitigs):
        links = " ".join(f"L:{'+' if sa else '-'}:{b}:{'+' if sb else '-'}" for (sa, b, sb) in by_src.get(i, []))
        lines.append(f">{i} LN:i:{len(u)} KC:i:{len(u) - ug.k + 1} km:f:1.0 {links}".rstrip())
        lines.append(u if i % 3 else u.lower())  # readers must accept lower case
    text = "\n".join(lines) + "\n"
    if gz:
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        Path(path).write_text(text)


def _fasta_seqs(text):
    return [l for l in text.split("\n") if l and not l.startswith(">")]


@pytest.mark.parametrize("gz", [False, True])
def test_reader_builds_the_clib_graph(tmp_path, gz, oracle, product_lib):
    k = 15
    ug = synth.g_seq(3000, seed=4, k=k, haplotypes=3, sub_rate=0.03)
    p = tmp_path / ("u.fa.gz" if gz else "u.fa")
    write_bcalm2(p, ug, gz=gz)
    G, store = api.read_bcalm2(str(p), k)
    assert len(store) == len(ug.unitigs) and store.sequences() == ug.unitigs
    ref = api.Bigraph.from_unitig_links(ug.weights, ug.links).export()
    got = G.export()
    for key in ("mirror", "edge_from", "edge_to", "edge_weight", "edge_unitig", "edge_forwards"):
        assert np.array_equal(ref[key], got[key]), key


def test_cli_eulertigs_fasta(tmp_path, oracle, product_lib):
    k = 21
    ug = synth.g_seq(4000, seed=6, k=k, haplotypes=3, sub_rate=0.02)
    inp, out = tmp_path / "unitigs.fa", tmp_path / "euler.fa.gz"
    write_bcalm2(inp, ug)
    r = subprocess.run([sys.executable, "-m", "matchtigs_amd", "--bcalm-in", str(inp), "-k", str(k), "--eulertigs-fa-out", str(out)],
                       capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr
    assert "Computing eulertigs took" in r.stderr
    fa = gzip.open(out, "rt").read()
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    assert fa == og.fasta(og.compute_eulertigs(k), ug.unitigs, k)
    assert synth.kmer_set_of_tigs(_fasta_seqs(fa), k) == ug.kmers


def test_cli_eulertigs_gfa_out(tmp_path, oracle, product_lib):
    """--eulertigs-gfa-out (bin.rs:97-99, 667-818): same tigs as the fasta output, as GFA S records, optionally gzipped."""
    k = 15
    ug = synth.g_seq(2500, seed=12, k=k, haplotypes=3, sub_rate=0.03)
    inp = tmp_path / "unitigs.fa"
    write_bcalm2(inp, ug, gz=False)
    r = subprocess.run([sys.executable, "-m", "matchtigs_amd", "--bcalm-in", str(inp), "-k", str(k),
                        "--eulertigs-gfa-out", str(tmp_path / "e.gfa.gz"), "--eulertigs-fa-out", str(tmp_path / "e.fa")],
                       capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr
    seqs = _fasta_seqs((tmp_path / "e.fa").read_text())
    gfa = gzip.open(tmp_path / "e.gfa.gz", "rt").read()
    assert gfa == f"H\tKL:Z:{k}\n" + "".join(f"S\t{i + 1}\t{s}\n" for i, s in enumerate(seqs))
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    assert (tmp_path / "e.fa").read_text() == og.fasta(og.compute_eulertigs(k), ug.unitigs, k)


def test_cli_flag_rules(tmp_path, product_lib):
    def run(*a):
        return subprocess.run([sys.executable, "-m", "matchtigs_amd", *a], capture_output=True, text=True, cwd=str(ROOT))

    assert run().returncode != 0 and "Missing input argument" in run().stderr                      # bin.rs:855-858
    r = run("--bcalm-in", "a", "--fa-in", "b", "-k", "5", "--eulertigs-fa-out", "o")
    assert r.returncode != 0 and "Too many input arguments" in r.stderr                            # bin.rs:860-862
    assert run("--bcalm-in", "a", "--eulertigs-fa-out", "o").returncode != 0                       # -k required
    (tmp_path / "bad.fa").write_text(">0 LN:i:3\nACN\n")
    r = run("--bcalm-in", str(tmp_path / "bad.fa"), "-k", "3", "--eulertigs-fa-out", str(tmp_path / "o.fa"))
    assert r.returncode != 0 and "not in the DNA alphabet" in r.stderr
    (tmp_path / "short.fa").write_text(">0 LN:i:3\nACG\n")
    r = run("--bcalm-in", str(tmp_path / "short.fa"), "-k", "5", "--eulertigs-fa-out", str(tmp_path / "o.fa"))
    assert r.returncode != 0 and "< k" in r.stderr


@pytest.mark.gpu
def test_cli_greedytigs_fasta_on_gpu(tmp_path, oracle, product_lib):
    """BASELINE configs[0] shape (`--bcalm-in X -k 31 --greedytigs-fa-out Y`) on a tiny real dBG."""
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    k = 31
    ug = synth.g_seq(8000, seed=8, k=k, haplotypes=4, sub_rate=0.02)
    inp, out = tmp_path / "unitigs.fa.gz", tmp_path / "greedy.fa"
    write_bcalm2(inp, ug, gz=True)
    r = subprocess.run([sys.executable, "-m", "matchtigs_amd", "--bcalm-in", str(inp), "-k", str(k), "-t", "4",
                        "--greedytigs-fa-out", str(out), "--eulertigs-fa-out", str(tmp_path / "euler.fa")],
                       capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr
    fa = out.read_text()
    og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    tigs, _ = og.compute_greedytigs(k)
    assert fa == og.fasta(tigs, ug.unitigs, k)
    assert synth.kmer_set_of_tigs(_fasta_seqs(fa), k) == ug.kmers
    og2 = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
    assert (tmp_path / "euler.fa").read_text() == og2.fasta(og2.compute_eulertigs(k), ug.unitigs, k)
