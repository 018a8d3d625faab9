"""Writes tests/golden/ref_inputs.txt: the inputs of tools/ref_fixtures/dump_fixtures.rs (see README.md beside this file), in the
input form of the reference's C-ABI (clib.rs:124-259: unitig weights + links in call order). Deterministic; rerun = same bytes.
    python tools/ref_fixtures/make_inputs.py"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import fuzz_small  # noqa: E402
import helpers  # noqa: E402
import pyref  # noqa: E402
from matchtigs_amd import synth  # noqa: E402


def panics(weights, links, k) -> bool:
    """The reference aborts on this input (e.g. an odd self-mirror node with nothing to pair it with): not a fixture."""
    try:
        for algo in (pyref.compute_greedytigs, pyref.compute_eulertigs):
            algo(pyref.from_unitig_links(list(weights), list(links)), k)
    except (AssertionError, ValueError, IndexError, KeyError):
        return True
    return False


def cases():
    for kat in json.loads((ROOT / "tests" / "golden" / "kats.json").read_text()):
        if "unitigs" in kat:
            links = helpers.links_of_bigraph(kat["mirror"], kat["unitigs"])
            yield "kat:" + kat["name"].split()[0], kat["k"], [w for (_, _, w) in kat["unitigs"]], links
    n = 0
    seed = 700000
    while n < 200:
        k, mirror, unitigs = fuzz_small.tiny_bigraph(seed)
        links = helpers.links_of_bigraph(mirror, unitigs)
        weights = [w for (_, _, w) in unitigs]
        if not panics(weights, links, k):
            yield f"tiny:{seed}", k, weights, links
            n += 1
        seed += 1
    n = 0
    seed = 9000
    while n < 8:
        bg = fuzz_small.medium_bigraph(seed)
        unitigs = [(int(bg.edge_from[2 * u]), int(bg.edge_to[2 * u]), int(bg.edge_weight[2 * u])) for u in range(bg.n_edges // 2)]
        links = helpers.links_of_bigraph([int(x) for x in bg.mirror], unitigs)
        weights = [w for (_, _, w) in unitigs]
        if len(links) < 12000 and not panics(weights, links, bg.k):
            yield f"medium:{seed}", bg.k, weights, links
            n += 1
        seed += 1
    ua = synth.g_seq_arrays(20000, seed=1, k=31, haplotypes=4, sub_rate=0.02)
    yield "gseq:20000", 31, [int(x) for x in ua.weights], [tuple(int(y) for y in l) for l in ua.links]


def main():
    out = ["# inputs of tools/ref_fixtures/dump_fixtures.rs (clib.rs:124-259 form); written by tools/ref_fixtures/make_inputs.py"]
    count = 0
    for name, k, weights, links in cases():
        assert " " not in name
        out.append(f"case {name} k {k} unitigs {len(weights)} links {len(links)}")
        out.append("w " + " ".join(str(int(w)) for w in weights))
        out.extend(f"l {int(a)} {1 if sa else 0} {int(b)} {1 if sb else 0}" for (a, sa, b, sb) in links)
        count += 1
    path = ROOT / "tests" / "golden" / "ref_inputs.txt"
    path.write_text("\n".join(out) + "\n")
    print(f"{count} cases, {len(out)} lines -> {path}")


if __name__ == "__main__":
    main()
