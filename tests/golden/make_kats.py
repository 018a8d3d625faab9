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
        # T4 (edge ids: unitig u = edges 2u / 2u+1; dummies follow from id 10):
        # greedy: e10/e11 = matched 4->6 / 7->5 (w 2), e12/e13 = breaking 10->0 / 1->11, e14/e15 = breaking 8->2 / 3->9 (w 5).
        "greedy_euler_cycles": [[10, 8, 12, 0, 4, 6, 14, 2]],
        "greedy_tigs": [[0, 4, 6], [2, 10, 8]],
        # eulertigs: e10/e11 = 10->0 / 1->11, e12/e13 = 8->2 / 3->9, e14/e15 = 7->5 / 4->6 (all breaking, w 5).
        "euler_euler_cycles": [[15, 8, 10, 0, 4, 6, 12, 2]],
        "euler_tigs": [[8], [0, 4, 6], [2]],
    },
    "why": "source 4 (demand mult[5]=1) settles 4 then 6 at distance 2 (live) -> claims it; 8,10,1,3 have no "
           "out-edges; source 7's demand mult[6] is 0 by then. Euleriser: OUT order 10,8,3,1 / IN order 0,2,9,11. "
           "T4 greedy: the walk starts at the lowest unused edge e0 (0->4); at node 4 the NEWEST out-edge is the matched dummy "
           "e10 (petgraph newest-first, App. A.3), so 4->6 is first walked over the dummy; 6: e8 (newer than e6) -> 10: e12 -> 0: "
           "stuck at the start: [0,10,8,12]. The scan from index 0 finds node 4 (index 1) with e4 unused: rotate_left(1) and "
           "continue: e4, 6: e6 (e8 used), 8: e14, 2: e2, 4: stuck -> [10,8,12,0,4,6,14,2]. Cutter: dummies e10 (2), e12 (5), e14 (5): the "
           "first STRICTLY longest is e12 at index 2 (e14's 5 is not > 5): rotate_left(2) = [12,0,4,6,14,2,10,8]; cut at e12 (index 0) "
           "and at e14 (weight >= k): tigs [0,4,6] and the tail [2,10,8] (e10 has weight 2 < k and stays inside). "
           "Eulertigs: node 4's newest out-edge is the breaking edge e15 (4->6): [0,15,8,10], splice at index 1: "
           "[15,8,10,0,4,6,12,2]; all dummies weigh 5: the first one (index 0) is the rotation point; cuts at e15, e10, e12.",
})

# --- KAT-2: equal distance, lower node index wins (heap order (distance, node), App. A.1) ----------------
KATS.append({
    "name": "KAT-2 tie at equal distance broken by node index",
    "k": 5,
    "mirror": pairs_mirror(10),
    "unitigs": [[2, 0, 9], [4, 0, 9], [6, 0, 9], [0, 8, 2], [0, 10, 2], [8, 12, 9], [8, 14, 9], [10, 16, 9], [10, 18, 9]],
    "expect": {
        "pairs": [[0, 8, 2]],
        # optimal matchtigs' matching instance (matchtigs/mod.rs:150-719), derived by hand below
        "matching_stats": {"transformed_node_count": 3, "edge_count": 2, "wcc_amount": 2, "matching_node_count": 14,
                           "matching_edge_count": 19, "mirror_biedges": 0, "mirror_expanded_biedges": 0},
        "matching_instance": "14 19\n0 1 2\n0 2 2\n0 3 4\n0 10 0\n0 11 0\n1 4 4\n1 10 0\n1 11 0\n2 5 4\n2 10 0\n2 11 0\n"
                             "3 4 2\n3 5 2\n3 12 0\n3 13 0\n4 12 0\n4 13 0\n5 12 0\n5 13 0\n",
    },
    "why": "node 0: in 3 / out 2 -> demand 1, T = 2; targets 8 and 10 (each out 2 / in 1) both at distance 2; "
           "D = [(8,2),(10,2)]; 8 is claimed, demand drops to 0 and the loop breaks at (10,2) (greedytigs/mod.rs:412-414). "
           "Source 9 = mirror(8) is skipped later (mult[8] == 0), source 11 only reaches the dead node 1. "
           "Matching instance: the all-targets searches give (0,8,2), (0,10,2), (9,1,2), (11,1,2) (out-nodes 3,5,7,12,.. have no "
           "out-edges). Matching ids in first-touch order: binode {0,1} -> 0, {8,9} -> 1, {10,11} -> 2; edge map (0,1) -> 2 and "
           "(0,2) -> 2 (each inserted twice, once per strand). The two strands are separate WCCs; in node order node 0 puts the forward "
           "one at index 0, node 1 the mirror one at index 1; the extra offset of an id is written per input node in ascending order "
           "(matchtigs/mod.rs:569-587), so the odd (larger) node of every binode wins: 2*3 + 4*1 = 10 for all three ids. File: header "
           "2*3 + 4*2 = 14 nodes, 2*2 + 3 + 4*3 = 19 edges; first copy = edges of id 0, then per id its copy edge (weight k-1 = 4) and "
           "its two extra edges; second copy shifted by 3 with extras +2/+3.",
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
    "expect": {
        "pairs": [[0, 1, 2]],
        "matching_stats": {"transformed_node_count": 2, "edge_count": 1, "wcc_amount": 1, "matching_node_count": 8,
                           "matching_edge_count": 12, "mirror_biedges": 1, "mirror_expanded_biedges": 1},
        "matching_instance": "8 12\n0 1 2\n0 2 4\n0 4 0\n0 5 0\n1 3 4\n1 4 0\n1 5 0\n2 3 2\n2 6 0\n2 7 0\n3 6 0\n3 7 0\n",
    },
    "why": "node 0: in 4 / out 2 -> demand 2; candidate node 1 == mirror(0) with demand >= 2 -> is_self_mirror_edge, "
           "multiplicity_reduction 2 (greedytigs/mod.rs:355-357, 399, 468-469). "
           "Matching instance: the only result is (0,1,2), a mirror biedge; binode {0,1} gets the two ids 0,1 (|diff| = 2) and the "
           "target shares them, so of the four id combinations (0,0) and (1,1) are skipped as self-loops and (0,1), (1,0) hit the "
           "same key (0,1): one edge, one expanded mirror biedge. One WCC (the arc 0->1 joins the strands): all extra offsets 2*2 = 4.",
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

# --- KAT-6: Hierholzer splice at a non-zero, non-adjacent index (App. A.2) --------------------------------
KATS.append({
    "name": "KAT-6 Euler walk stuck early, splice at index 2",
    "k": 5,
    "mirror": pairs_mirror(4),
    # A: 0->2, B: 2->4, D: 4->6, E: 6->4, C: 4->0 (C is the NEWEST out-edge of node 4, so the walk goes home first)
    "unitigs": [[0, 2, 3], [2, 4, 3], [4, 6, 3], [6, 4, 3], [4, 0, 3]],
    "expect": {
        "pairs": [],
        "greedy_euler_cycles": [[8, 0, 2, 4, 6]], "greedy_tigs": [[8, 0, 2, 4, 6]],
        "euler_euler_cycles": [[8, 0, 2, 4, 6]], "euler_tigs": [[8, 0, 2, 4, 6]],
        "clib": {"tigs_edge_out": [4, 0, 1, 2, 3], "tigs_insert_out": [0, 0, 0, 0, 0], "tigs_out_limits": [5]},
    },
    "why": "every node is balanced: no sources, no dummies. Edges: e0 0->2, e2 2->4, e4 4->6, e6 6->4, e8 4->0 (+ mirrors). "
           "Walk from e0: 2: e2, 4: newest first = e8 (4->0), 0: stuck at the start with [0,2,8]. Scan from index 0: node 0 "
           "(from of e0) and node 2 (from of e2) have nothing left, node 4 (from of e8, index 2) has e4: rotate_left(2) = "
           "[8,0,2], continue e4, 6: e6, 4: stuck -> [8,0,2,4,6]. No dummy: no rotation, one tig. clib: unitig ids with "
           "sign (all forwards; clib.rs:397-398).",
})

# --- KAT-7: two components; a unitig walked backwards ------------------------------------------------------
KATS.append({
    "name": "KAT-7 two bicycles, lowest unused edge starts, backward unitigs",
    "k": 5,
    "mirror": pairs_mirror(5),
    # P: 0->2, Q: 2->0 | Y: 8->7 (c -> mirror(b)), X: 4->6 (a -> b), Z: 9->4 (mirror(c) -> a)
    "unitigs": [[0, 2, 3], [2, 0, 3], [8, 7, 3], [4, 6, 3], [9, 4, 3]],
    "expect": {
        "pairs": [],
        "greedy_euler_cycles": [[0, 2], [4, 7, 9]], "greedy_tigs": [[0, 2], [4, 7, 9]],
        "euler_euler_cycles": [[0, 2], [4, 7, 9]], "euler_tigs": [[0, 2], [4, 7, 9]],
        "clib": {"tigs_edge_out": [0, 1, 2, -3, -4], "tigs_insert_out": [0, 0, 0, 0, 0], "tigs_out_limits": [2, 5]},
    },
    "why": "component 1: e0 0->2, e2 2->0 -> cycle [0,2]. The outer loop then starts at the lowest unused edge e4 (8->7): "
           "node 7's only out-edge is e7 (the MIRROR edge of X: 7->5), node 5's is e9 (mirror of Z: 5->8), 8: stuck at the start: "
           "[4,7,9], i.e. unitig 2 forwards, unitigs 3 and 4 backwards (clib: +2, -3, -4; unitig 0 loses its sign).",
})

# --- KAT-8: only a matched dummy in the cycle: rotation to it and the cut at index 0 -------------------------
KATS.append({
    "name": "KAT-8 cycle with one matched dummy and no breaking edge",
    "k": 5,
    "mirror": pairs_mirror(4),
    # A: 0->4, B: 2->4, U: 4->6 (2 k-mers), C: 6->0, D: 6->2
    "unitigs": [[0, 4, 9], [2, 4, 9], [4, 6, 2], [6, 0, 9], [6, 2, 9]],
    "expect": {
        "out_nodes": [4, 7],
        "multiplicity": [0, 0, 0, 0, -1, 1, 1, -1],
        "pairs": [[4, 6, 2]],
        "greedy_euler_cycles": [[0, 10, 8, 2, 4, 6]],
        "greedy_tigs": [[8, 2, 4, 6, 0]],
        "greedy_tig_count": 1, "greedy_cumulative_length": 42,
        "clib": {"tigs_edge_out": [4, 1, 2, 3, 0], "tigs_insert_out": [0, 0, 0, 0, 0], "tigs_out_limits": [5]},
    },
    "why": "node 4: in 2 / out 1 -> source with demand 1; node 6: in 1 / out 2 -> target at distance 2: pair (4,6,2), dummy "
           "e10 4->6 / e11 7->5; source 7 = mirror(6) then has demand 0. Everything is balanced: no breaking edge. Walk from e0 "
           "(0->4): 4: e10 (newest), 6: e8 (6->2, newer than e6), 2: e2, 4: e4, 6: e6 (6->0), 0: stuck = start, all six biedges "
           "used: [0,10,8,2,4,6]. Cutter: the only dummy (weight 2 > 0) is at index 1: rotate_left(1) = [10,8,2,4,6,0]; a dummy at "
           "index 0 is cut although its weight is < k (greedytigs/mod.rs:768); the rest is one tig [8,2,4,6,0] "
           "(cumulative length 9+9+2+9+9 + (k-1) = 42; the dummy itself is dropped, so tigs_insert_out is all 0).",
})

# --- KAT-C: the cutter alone on hand-made cycles (greedytigs/mod.rs:726-789) --------------------------------
KATS.append({
    "name": "KAT-C cutter: strictly-longest rotation, consecutive breaking edges, weight-0 dummies, trailing dummy",
    "k": 5,
    "mirror": pairs_mirror(4),
    "unitigs": [[0, 2, 3], [2, 4, 3], [4, 6, 3], [6, 0, 3]],
    # dummy biedges in insertion order (out, in, weight): edge ids 8/9, 10/11, 12/13, 14/15, 16/17, 18/19, 20/21
    "dummy_pairs": [[2, 4, 5], [4, 6, 5], [6, 0, 3], [0, 2, 7], [0, 4, 0], [4, 0, 0], [6, 4, 0]],
    "cycles": [[0, 8, 10, 2, 12, 4, 14, 6, 16], [1, 3, 18], [19, 5], [20]],
    "expect": {"cut_tigs": [[6, 16, 0], [2, 12, 4], [1, 3], [5]]},
    "why": "cycle 1: dummy weights by index: 1:5, 2:5, 4:3, 6:7, 8:0; the running strict maximum ends at index 6 (7): "
           "rotate_left(6) = [14,6,16,0,8,10,2,12,4]; index 0 (e14) is cut; e16 (weight 0 < k, not index 0) stays inside; "
           "e8 (5 >= k) at index 4 emits [6,16,0]; e10 (5 >= k) at index 5 follows immediately: offset == index, nothing is "
           "emitted ('Found consecutive breaking edges', :772-774); e12 (3 < k) stays; the last edge e4 is original: tail "
           "[2,12,4]. Cycle 2: only a weight-0 dummy -> longest weight 0 -> NO rotation (:746); no cut; the last edge is a dummy, "
           "so the tail is everything but the last edge: [1,3] (:784-786). Cycle 3: weight-0 dummy at index 0 is cut (:768), "
           "tail [5]. Cycle 4: a lone dummy: cut at index 0, nothing left.",
})

out = Path(__file__).resolve().parent / "kats.json"
out.write_text(json.dumps(KATS, indent=1))
print(f"wrote {out} ({len(KATS)} cases)")
