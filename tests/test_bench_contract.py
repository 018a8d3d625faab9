"""bench.py's output contract (one JSON line with the driver's keys plus roofline and cpu_baseline), on a small graph."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_bench_json_line(product_lib):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log2-edges", "19",
                        "--cpu-baseline-seconds", "2"], capture_output=True, text=True, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "device_mode", "scaling_stages_ms", "seeds",
                "roofline_stages", "cold_step_ms", "visited_per_step", "full_size", "one_shot", "clib_route"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["unit"] == "edges/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["units_per_step"]["relaxed_edges"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6 and rf["kernels"] and "sssp_enum_kernel" in rf["kernels"][0]["kernel"]
    assert "traffic" in rf and "traffic_note" in rf
    # the pruned search visits no more than the full balls and emits the same candidates; the roofline prices the visited bytes
    vis, full = d["visited_per_step"], d["units_per_step"]
    assert vis["emitted"] == full["emitted"] and vis["settled_nodes"] <= full["settled_nodes"] and vis["relaxed_edges"] <= full["relaxed_edges"]
    assert rf["algorithmic_bytes_per_launch"] == 5 * vis["relaxed_edges"] + 12 * vis["settled_nodes"] + 12 * vis["emitted"]
    assert rf["full_ball_equivalent"]["algorithmic_bytes_per_launch"] == 5 * full["relaxed_edges"] + 12 * full["settled_nodes"] + 12 * full["emitted"]
    # every other GPU stage of the step has its own roofline entry (HIP-event time, byte model, frac), in both Euler modes
    stages = {(x["stage"], x["euler_mode"]) for x in d["roofline_stages"]}
    assert {("replay", "device"), ("insert_eulerise", "device"), ("replay", "host"), ("insert_eulerise", "host"), ("records", "host"), ("cut", "host")} <= stages
    # device order: the tigs straight from the pairing (one stage, cut_first_device.hip) -- or, where the graph sent the step through the
    # closed walks, decomposition + cut
    assert ("tigs_from_pairing", "device") in stages or {("decomposition", "device"), ("cut", "device")} <= stages
    for x in d["roofline_stages"]:
        assert x["avg_launch_ms"] > 0 and x["algorithmic_bytes"] > 0 and abs(x["frac"] - x["achieved"] / 8000.0) < 1e-4
        # the pass-independent floor (inputs once + outputs once) lies below what the implementation's passes move
        assert 0 < x["minimum_bytes"] <= x["algorithmic_bytes"] and 0 < x["frac_of_minimum"] <= x["frac"] + 1e-6
    cs = d["cold_step_ms"]
    assert cs["device"]["step_ms"] > 0 and cs["host"]["step_ms"] > 0 and cs["device"]["device_graph_build_ms"] > 0
    assert d["full_size"] is None  # (only the headline configuration runs the nominal-size step)
    # the consuming one-shot call (host arrays in -> clib.rs arrays out), each Euler mode in a process of its own; same tigs as the steps
    for mode in ("device", "host"):
        os_ = d["one_shot"][mode]
        assert "error" not in os_, os_
        assert os_["total_s"] > 0 and os_["same_result_both_calls"] and os_["later_call"]["tigs"] == d["config"]["tigs"]
        assert set(os_["later_call"]["phases_s"]) >= {"device_build", "sssp", "eulerise", "euler", "cut"}
    # the reference's own way in (initialise / merge_nodes per link / build_graph / compute): same number of tigs in both Euler modes (T3)
    cr = d["clib_route"]
    for mode in ("device", "host"):
        assert "error" not in cr[mode], cr[mode]
        assert cr[mode]["links"] > cr[mode]["unitigs"] > 0 and cr[mode]["build_graph_s"] > 0 and cr[mode]["compute_tigs_s"] > 0
    assert cr["device"]["tigs"] == cr["host"]["tigs"] > 0 and cr["device"]["tig_edges"] == cr["host"]["tig_edges"]
    dm = d["device_mode"]
    assert dm["ms_per_step"] > 0 and dm["tigs"] == d["config"]["tigs"] and "finish" in dm["phases_ms"]   # T3 across the two modes
    assert set(d["seeds"]) == {"2", "3"} and all(v["sssp_stage_ms"] > 0 for v in d["seeds"].values())
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert cb["pairs"] == d["config"]["pairs"] and cb["tigs"] == d["config"]["tigs"]   # the CPU port and the GPU path agree
