#!/usr/bin/env python3
"""Writes tests/golden/kats.json: HAND-DERIVED known-answer cases for the greedy-matchtigs path.

The reference has no golden vectors and cannot be run here, so these expectations were derived by hand
from the cited reference lines (+ the SURVEY Appendix A policies for the absent third-party crates);
the `expect` values below are literals typed in from those derivations, NOT outputs of any code in this
repository. Each case says which rule it pins. Derivations: see the `why` strings and SURVEY.md App. C.

Graph encoding: `mirror[n]`, and `unitigs` = [from, to, weight]; unitig u becomes edge 2u (from -> to,
forwards) and edge 2u+1 (mirror(to) -> mirror(from), backwards), as /root/reference/src/clib.rs:239-248.
`raw_edges` (KAT-E only) gives explicit edges (from, to, weight, dummy_id, handle, forwards) instead.
"""
import json
from pathlib import Path


def pairs_mirror(n_pairs, self_mirrors=()):
    m = []
    for i in range(n_pairs):
        m += [2 * i + 1, 2 * i]
    base = len(m)
    for j in range(len(self_mirrors)):
        m.append(base + j)
    return m


KATS = []

# --- KAT-E: the reference's own (assertion-free) test graph, implementation/mod.rs:764-783 -------------
KATS.append({
    "name": "KAT-E euleriser on the reference's test graph",
    "k": 4,
    "mirror": [1, 0, 2, 3, 5, 4, 6, 7],
    "raw_edges": [[0, 3, 0, 1, 1, True], [3, 1, 0, 1, 1, False], [2, 0, 0, 2, 2, True], [1, 2, 0, 2, 2, False],
                  [6, 4, 0, 3, 3, True], [5, 6, 0, 3, 3, False], [7, 4, 0, 4, 4, True], [5, 7, 0, 4, 4, False]],
    "euleriser_start_dummy_id": 5,
    "expect": {
        # self-mirrors 2,3,6,7 have odd degree -> paired two by two (mod.rs:481-493): ids 6 and 7;
        # node 4 has difference -2, node 5 +2; 5 == mirror(4) but OUT[4] = -2 is not > -2 (mod.rs:263),
        # so 4->5 is added once and both counters drop by 2 (the mirror updates at mod.rs:609-644).
        "breaking_edges": [[2, 3, 4, 6, True], [3, 2, 4, 6, False], [6, 7, 4, 7, True], [7, 6, 4, 7, False],
                           [4, 5, 4, 8, True], [4, 5, 4, 8, False]],
        "final_dummy_id": 8,
    },
    "why": "SURVEY App. C KAT-E",
})

# --- KAT-1: one greedy match, SURVEY App. C -------------------------------------------------------------
KATS.append({
    "name": "KAT-1 one greedy match (k=5)",
    "k": 5,
    "mirror": pairs_mirror(6),
    "unitigs": [[0, 4, 9], [2, 4, 9], [4, 6, 2], [6, 8, 9], [6, 10, 9]],
    "expect": {
        "out_nodes": [1, 3, 4, 7, 8, 10],
        "multiplicity": [1, -1, 1, -1, -1, 1, 1, -1, -1, 1, -1, 1],
        "pairs": [[4, 6, 2]],
        "greedy_tig_count": 2, "greedy_cumulative_length": 48,
        "euler_tig_count": 3, "euler_cumulative_length": 50,
        "greedy_breaking_edges": [[10, 0], [8, 2]],
    },
    "why": "source 4 (demand mult[5]=1) settles 4 then 6 at distance 2 (live) -> claims it; 8,10,1,3 have no "
           "out-edges; source 7's demand mult[6] is 0 by then. Euleriser: OUT order 10,8,3,1 / IN order 0,2,9,11.",
})

# --- KAT-2: equal distance, lower node index wins (heap order (distance, node), App. A.1) ----------------
KATS.append({
    "name": "KAT-2 tie at equal distance broken by node index",
    "k": 5,
    "mirror": pairs_mirror(10),
    "unitigs": [[2, 0, 9], [4, 0, 9], [6, 0, 9], [0, 8, 2], [0, 10, 2], [8, 12, 9], [8, 14, 9], [10, 16, 9], [10, 18, 9]],
    "expect": {"pairs": [[0, 8, 2]]},
    "why": "node 0: in 3 / out 2 -> demand 1, T = 2; targets 8 and 10 (each out 2 / in 1) both at distance 2; "
           "D = [(8,2),(10,2)]; 8 is claimed, demand drops to 0 and the loop breaks at (10,2) (greedytigs/mod.rs:412-414). "
           "Source 9 = mirror(8) is skipped later (mult[8] == 0), source 11 only reaches the dead node 1.",
})

# --- KAT-3: bound is inclusive: distance k-1 accepted, k rejected ---------------------------------------
KATS.append({
    "name": "KAT-3 distance k-1 accepted, distance k rejected",
    "k": 5,
    "mirror": pairs_mirror(10),
    "unitigs": [[2, 0, 9], [4, 0, 9], [6, 0, 9], [0, 8, 4], [0, 10, 5], [8, 12, 9], [8, 14, 9], [10, 16, 9], [10, 18, 9]],
    "expect": {"pairs": [[0, 8, 4]]},
    "why": "max_weight = k-1 = 4 (greedytigs/mod.rs:329); pop (4, 8) is kept (4 > 4 is false), pop (5, 10) breaks the search.",
})

# --- KAT-4a/4b: the candidate that is the mirror of the source ------------------------------------------
KATS.append({
    "name": "KAT-4a mirror-of-self candidate skipped when demand < 2",
    "k": 5,
    "mirror": pairs_mirror(4),
    "unitigs": [[0, 1, 2], [2, 0, 9], [4, 0, 9], [6, 0, 9]],
    "expect": {"pairs": []},
    "why": "unitig 0->1 and its mirror are both arcs 0->1; node 0: in 3 / out 2 -> demand mult[1] = 1; the only "
           "candidate is node 1 == mirror(0) and out_node_multiplicity < 2 -> continue (greedytigs/mod.rs:352-354).",
})
KATS.append({
    "name": "KAT-4b mirror-of-self candidate taken as a 2-cost self-mirror edge when demand >= 2",
    "k": 5,
    "mirror": pairs_mirror(5),
    "unitigs": [[0, 1, 2], [2, 0, 9], [4, 0, 9], [6, 0, 9], [8, 0, 9]],
    "expect": {"pairs": [[0, 1, 2]]},
    "why": "node 0: in 4 / out 2 -> demand 2; candidate node 1 == mirror(0) with demand >= 2 -> is_self_mirror_edge, "
           "multiplicity_reduction 2 (greedytigs/mod.rs:355-357, 399, 468-469).",
})

# --- KAT-5: a target claimed by an earlier source is skipped by a later one -----------------------------
KATS.append({
    "name": "KAT-5 dead target skipped by a later source",
    "k": 5,
    "mirror": pairs_mirror(15),
    "unitigs": [[2, 0, 9], [4, 0, 9], [0, 8, 1], [10, 8, 1], [10, 12, 3], [14, 10, 9], [16, 10, 9], [18, 10, 9],
                [8, 20, 9], [8, 22, 9], [8, 24, 9], [12, 26, 9], [12, 28, 9]],
    "expect": {"pairs": [[0, 8, 1], [10, 12, 3]]},
    "why": "sources 0 and 10 both have target 8 at distance 1 (8: in 2 / out 3 -> multiplicity 1). Source 0 claims it "
           "and clears its live bit (greedytigs/mod.rs:497-501); source 10's query then returns (12,3) only.",
})

out = Path(__file__).resolve().parent / "kats.json"
out.write_text(json.dumps(KATS, indent=1))
print(f"wrote {out} ({len(KATS)} cases)")
