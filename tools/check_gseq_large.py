import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib
from matchtigs_amd import api, synth
t0 = time.time()
ug = synth.g_seq(int(sys.argv[1]) if len(sys.argv) > 1 else 150000, seed=77, k=31, haplotypes=4, sub_rate=0.01)
print("unitigs", len(ug.unitigs), "kmers", len(ug.kmers), "gen %.1fs" % (time.time() - t0))
G = api.Bigraph.from_unitig_links(ug.weights, ug.links)
tigs = api.GreedytigAlgorithm.compute_tigs(G, api.GreedytigAlgorithmConfiguration.new(1, 31))
fa = api.write_walks_fasta(G, tigs, ug.unitigs, 31).decode()
og = oracle_lib.OracleGraph.from_unitig_links(ug.weights, ug.links)
otigs, _ = og.compute_greedytigs(31)
print("tigs", len(tigs), len(otigs), "fasta identical:", fa == og.fasta(otigs, ug.unitigs, 31))
seqs = [l for l in fa.splitlines() if l and not l.startswith(">")]
print("kmer set preserved:", synth.kmer_set_of_tigs(seqs, 31) == ug.kmers, "cum len", sum(len(s) for s in seqs), "vs unitigs", sum(len(u) for u in ug.unitigs))
G2 = api.Bigraph.from_unitig_links(ug.weights, ug.links)
t2 = api.GreedytigAlgorithm.compute_tigs(G2, api.GreedytigAlgorithmConfiguration(1, 31, euler_mode=api.EulerMode.Device))
fa2 = api.write_walks_fasta(G2, t2, ug.unitigs, 31).decode()
s2 = [l for l in fa2.splitlines() if l and not l.startswith(">")]
print("device euler: tigs", len(t2), "kmer set preserved:", synth.kmer_set_of_tigs(s2, 31) == ug.kmers, "cum len", sum(len(s) for s in s2))
