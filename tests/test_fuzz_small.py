"""10^4 tiny seeded bigraphs through the oracle, the independent Python restatement and the product's host stages on the CPU, and
3 000 of them through the HIP path and the clib.rs C-ABI on the GPU (tests/fuzz_small.py says what is compared and why the work
runs in child processes). SURVEY.md 8c(3): "differential: oracle vs an independent restatement on >= 10^4 seeded random bigraphs;
GPU vs oracle on the same"."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CORNER_RULES = ("demand_3_or_4", "self_mirror_source", "self_mirror_target", "own_mirror_skipped", "self_mirror_edge",
                "equal_distance_tie", "distance_k_minus_1")


FLIPPED_LIB = ROOT / "matchtigs_amd" / "libmatchtigs_flipped.so"


def _flipped_env():
    """The other setting of the five out-of-tree policies (include/mtg_policy.h: heap tie-break, inclusive bound, adjacency order,
    union-find tie, Hierholzer's splice rule; mask 31 = all five flipped): the product built by `make flipped`, the oracle's flipped build and the Python
    restatement under MTG_POLICY. Built here when missing (14 s)."""
    import os

    if not FLIPPED_LIB.exists():
        subprocess.run(["make", "-C", str(ROOT / "matchtigs_amd" / "csrc"), "ARCH=gfx950", "-j8", "flipped"], check=True, capture_output=True)
    env = dict(os.environ)
    env["MTG_POLICY"] = "31"
    env["MATCHTIGS_LIBRARY"] = str(FLIPPED_LIB)
    return env


def _run(mode, first, n, timeout, flipped=False):
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_small.py"), mode, str(first), str(n)], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=timeout, env=_flipped_env() if flipped else None)
    lines = r.stdout.splitlines()
    assert r.returncode == 0, f"{mode} fuzz, seeds {first}..{first + n - 1}: rc {r.returncode}\n" + "\n".join(lines[-3:]) + "\n" + r.stderr[-1500:]
    tally = dict(x.split("=") for x in next(l for l in lines if l.startswith("TALLY ")).split()[1:])
    ev_line = next((l for l in lines if l.startswith("EVENTS ")), "EVENTS")
    events = dict(x.split("=") for x in ev_line.split()[1:])
    return {k: int(v) for k, v in tally.items()}, {k: int(v) for k, v in events.items()}


def test_fuzz_10000_tiny_bigraphs_oracle_vs_restatement_vs_host_stages(oracle, product_lib):
    chunks = [(i * 2500, 2500) for i in range(4)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda c: _run("cpu", c[0], c[1], 600), chunks))
    tally, events = {}, {}
    for t, e in results:
        for k, v in t.items():
            tally[k] = tally.get(k, 0) + v
        for k, v in e.items():
            events[k] = events.get(k, 0) + v
    assert sum(tally.values()) == 10000
    assert tally.get("ok:pairs", 0) >= 3000 and tally.get("panic", 0) <= 500, tally  # most graphs are legal inputs, a third match pairs
    for rule in CORNER_RULES:  # every corner rule of the claim loop fired hundreds of times
        assert events.get(rule, 0) >= 300, (rule, events)


@pytest.mark.gpu
def test_fuzz_3000_tiny_bigraphs_through_the_hip_path(oracle, product_lib):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    tally, _ = _run("gpu", 20000, 3000, 800)  # (one child process on the GPU; other seeds than the CPU test's)
    assert sum(tally.values()) == 3000 and tally.get("ok:pairs", 0) >= 900 and tally.get("panic", 0) <= 150, tally


def test_fuzz_200_medium_bigraphs_oracle_vs_restatement_vs_host_stages(oracle, product_lib):
    """G-csr graphs of 50-3000 binodes with random parameters (k 3..300 ...): oracle == Python restatement == the product's host stages
    on greedy tigs and eulertigs."""
    chunks = [(i * 50, 50) for i in range(4)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda c: _run("cpu_medium", c[0], c[1], 900), chunks))
    tally = {}
    for t, _ in results:
        for k, v in t.items():
            tally[k] = tally.get(k, 0) + v
    assert sum(tally.values()) == 200 and tally.get("ok:pairs", 0) >= 100 and tally.get("panic", 0) <= 20, tally


@pytest.mark.gpu
def test_fuzz_300_medium_bigraphs_through_the_hip_path(oracle, product_lib):
    """The same family through the HIP path (plans 0-3 in turn): candidate lists, full-ball counters, device finish in reference
    order == oracle; device Euler mode: equal tig count and cumulative length; eulertigs == oracle."""
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    tally, _ = _run("gpu_medium", 1000, 300, 800)
    assert sum(tally.values()) == 300 and tally.get("ok:pairs", 0) >= 150 and tally.get("panic", 0) <= 30, tally


# ---- the same fuzz under the OTHER setting of the five policies the reference inherits from crates outside its tree --------------
# Pop order among equal distances, inclusive search bound, adjacency iteration order, union-find tie (include/mtg_policy.h) were
# restated from recollection: parity with the real binary is unpinned exactly there. Each is one named switch shared
# by the product's kernels and host stages, the oracle and the Python restatement; these tests hold the three parties to each other
# with all five switches flipped, so that if one recollection proves wrong the fix is one definition -- not a hunt through kernels.
def test_flipped_policies_are_a_different_behaviour(oracle, product_lib):
    """The flipped build really behaves differently: on a graph with an equal-distance tie the two settings claim different pairs."""
    import json

    code = (
        "import sys, json; sys.path.insert(0, 'tests'); import helpers, fuzz_small\n"
        "out = []\n"
        "for seed in range(0, 400):\n"
        "    k, mirror, unitigs = fuzz_small.tiny_bigraph(seed)\n"
        "    arrs = helpers.unitigs_to_arrays(mirror, unitigs)\n"
        "    if fuzz_small.reference_panics(arrs, k): out.append(None); continue\n"
        "    out.append([helpers.oracle_graph(*arrs).greedy_pairs(k)[0], helpers.oracle_graph(*arrs).compute_greedytigs(k)[0]])\n"
        "print(json.dumps(out))\n")
    res = []
    for env in (None, _flipped_env()):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        res.append(json.loads(r.stdout.splitlines()[-1]))
    both = [(a, b) for a, b in zip(*res) if a is not None and b is not None]
    assert len(both) >= 300
    assert sum(1 for a, b in both if a[0] != b[0]) >= 20, "pairs never differ between the two policy settings"
    assert sum(1 for a, b in both if a[1] != b[1]) >= 50, "tigs never differ between the two policy settings"


def test_policy_p5_alone_changes_walk_order_but_neither_tig_count_nor_cumulative_length(oracle):
    """P5 (Hierholzer's splice rule: resume at the FIRST or at the LAST position of the cycle with an unused out-edge) isolated in the
    Python restatement (MTG_POLICY = 16 flips nothing else): the closed walks cover the same biedges per bicycle in another order on
    part of the graphs, and -- SURVEY 8a's invariance note -- tig count and cumulative length never move."""
    import json

    code = (
        "import sys, json; sys.path.insert(0, 'tests'); import helpers, fuzz_small, pyref\n"
        "out = []\n"
        "for seed in range(0, 300):\n"
        "    k, mirror, unitigs = fuzz_small.tiny_bigraph(seed)\n"
        "    arrs = helpers.unitigs_to_arrays(mirror, unitigs)\n"
        "    if fuzz_small.reference_panics(arrs, k): out.append(None); continue\n"
        "    g = helpers.py_graph(*arrs)\n"
        "    tigs, _, _ = pyref.compute_greedytigs(g, k)\n"
        "    cum = sum(sum(g.edges[e].weight for e in t) + k - 1 for t in tigs)\n"
        "    g2 = helpers.py_graph(*arrs); pairs, _ = pyref.greedy_pairs(g2, k); did = pyref.insert_pair_edges(g2, pairs); pyref.make_eulerian(g2, did, k)\n"
        "    cyc = pyref.euler_cycles(g2)\n"
        "    out.append([cyc, sorted(sorted(min(e, g2.mirror_edge(e)) for e in c) for c in cyc), len(tigs), cum])\n"
        "print(json.dumps(out))\n")
    res = []
    for mask in ("0", "16"):
        env = dict(os.environ)
        env["MTG_POLICY"] = mask
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        res.append(json.loads(r.stdout.splitlines()[-1]))
    both = [(a, b) for a, b in zip(*res) if a is not None and b is not None]
    assert len(both) >= 200
    assert all(a[1] == b[1] for a, b in both), "the two settings cover different biedges per bicycle"
    assert all(a[2] == b[2] and a[3] == b[3] for a, b in both), "tig count or cumulative length depends on P5"
    assert sum(1 for a, b in both if a[0] != b[0]) >= 10, "the walk order never differs between the two settings of P5"


def test_fuzz_flipped_policies_2000_tiny_and_100_medium_bigraphs_cpu(oracle, product_lib):
    with ThreadPoolExecutor(max_workers=4) as ex:
        tiny = list(ex.map(lambda c: _run("cpu", c[0], c[1], 600, flipped=True), [(40000 + i * 500, 500) for i in range(4)]))
        medium = list(ex.map(lambda c: _run("cpu_medium", c[0], c[1], 900, flipped=True), [(3000 + i * 25, 25) for i in range(4)]))
    for results, total, pairs in ((tiny, 2000, 600), (medium, 100, 50)):
        tally = {}
        for t, _ in results:
            for k, v in t.items():
                tally[k] = tally.get(k, 0) + v
        assert sum(tally.values()) == total and tally.get("ok:pairs", 0) >= pairs and tally.get("panic", 0) <= total // 10, tally


@pytest.mark.gpu
def test_fuzz_flipped_policies_1000_tiny_and_100_medium_bigraphs_through_the_hip_path(oracle, product_lib):
    """libmatchtigs_flipped.so on the GPU against the flipped oracle: classification, candidate lists (plans 0 / 1 / 2), GPU claim
    replay, device finish in the (flipped) reference order, the one-shot operator, eulertigs and the clib.rs C-ABI (whose node
    numbering follows the flipped union-find tie)."""
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    tally, _ = _run("gpu", 50000, 1000, 800, flipped=True)
    assert sum(tally.values()) == 1000 and tally.get("ok:pairs", 0) >= 300 and tally.get("panic", 0) <= 60, tally
    tally, _ = _run("gpu_medium", 4000, 100, 800, flipped=True)
    assert sum(tally.values()) == 100 and tally.get("ok:pairs", 0) >= 50 and tally.get("panic", 0) <= 10, tally
