"""10^4 tiny seeded bigraphs through the oracle, the independent Python restatement and the product's host stages on the CPU, and
3 000 of them through the HIP path and the clib.rs C-ABI on the GPU (tests/fuzz_small.py says what is compared and why the work
runs in child processes). SURVEY.md 8c(3): "differential: oracle vs an independent restatement on >= 10^4 seeded random bigraphs;
GPU vs oracle on the same"."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CORNER_RULES = ("demand_3_or_4", "self_mirror_source", "self_mirror_target", "own_mirror_skipped", "self_mirror_edge",
                "equal_distance_tie", "distance_k_minus_1")


def _run(mode, first, n, timeout):
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_small.py"), mode, str(first), str(n)], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=timeout)
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
