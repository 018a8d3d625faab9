"""Per-stage HBM traffic of one bench step from rocprofv3 --pmc passes (raw *_counter_collection.csv under the given directories):
every dispatch is assigned to a stage of the step by walking the dispatches in order -- a stage begins at the first launch of one
of its marker kernels and lasts until another stage's marker -- and FETCH_SIZE / WRITE_SIZE (KB) are summed per stage and divided by
the number of steps the profiled command ran (= launches of classify_kernel).

usage: python tools/stage_traffic.py OUT.json KEY DIR [DIR ...]
Adds {KEY: {stage: {fetch_bytes, write_bytes, traffic_bytes, dispatches_per_step, kernels}}} to OUT.json (created if missing).
bench.py reports these as roofline_stages[*].traffic when its workload key and mode match (KEY = "<workload key>:<euler mode>").
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

# first kernel of a stage -> stage (kernel names as rocprofv3 prints them, matched by substring)
MARKERS = [
    ("classify_kernel", "classify"),
    ("active_range_kernel", "sssp"),
    ("sssp_enum_kernel", "sssp"),
    ("replay_state_init_kernel", "replay"),
    ("replay_record_init_kernel", "replay"),
    ("row_degree_kernel", "insert_eulerise"),
    ("pair_degree_kernel", "insert_eulerise"),
    ("degree_rank_offset_kernel", "buckets"),   # re-labelled below: decomposition (device mode) or records (reference order)
    ("rotation_kernel", "cut"),
]
# once per graph, inside its first finish (before the degrees are read): the kept buckets of the original darts -- not a stage of a step
ONE_OFF = ("degree_rank_kernel", "fill_kernel")
# kernels whose reads are dependent random accesses of 4 to 64 bytes (one request each, counted as issued); the others stream
# coalesced 4-byte words, where gfx950's FETCH_SIZE reports half the bytes (MI355X_MICROARCH.md, HBM)
GATHER = ("sssp_enum_kernel", "sssp_kernel", "replay_rounds_kernel", "walk_measure_kernel", "walk_write_kernel", "label_hook_kernel",
          "propose_kernel", "flatten_kernel", "rotate_kernel", "wyllie_kernel", "wide_build_kernel", "mid_build_slice_kernel", "lean_build_kernel",
          "zip_check_kernel", "sort_lists_kernel", "replay_compact_kernel", "replay_dense_fill_kernel", "root_len_kernel", "succ_node_kernel",
          "stretch_measure_kernel", "stretch_write_kernel")
ORDER = ["classify", "sssp", "replay", "insert_eulerise", "buckets", "cut"]
# kernels that only occur in one kind of step tell which stage the buckets belong to
DEVICE_ONLY = ("succ_node_kernel", "label_hook_kernel", "walk_measure_kernel")
CUT_FIRST_ONLY = ("stretch_measure_kernel",)  # device order without closed walks (cut_first_device.hip): decomposition + cut in one stage
HOST_ONLY = ("lean_ext_kernel", "lean_build_kernel", "wide_build_kernel", "mid_build_slice_kernel")


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")  # (before the argument list is cut at its first parenthesis)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("mtg::", "").replace("(anonymous namespace)::", "").replace("hu::", "")


def main():
    out, key, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per_stage = defaultdict(lambda: defaultdict(float))
    kernels = defaultdict(set)
    n_dispatch = defaultdict(int)
    steps = 0
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = defaultdict(dict)  # dispatch id -> {counter: value}; name
            names = {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    did = int(row.get("Dispatch_Id") or row.get("Dispatch Id"))
                    names[did] = short(row.get("Kernel_Name") or row.get("Kernel Name") or "")
                    c = row.get("Counter_Name") or row.get("Counter Name")
                    rows[did][c] = rows[did].get(c, 0.0) + float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            order = sorted(names)
            # pass 1: stage of every dispatch. Inside a step the stages follow each other in ORDER; a marker that would go back
            # (other than classify_kernel, which opens the next step) belongs to what the command does after its steps (unit
            # counting, other seeds): "post", not counted. The counting instantiations of sssp_kernel are instrumentation as well.
            stage_of = {}
            cur = "setup"
            one_off = False
            for did in order:
                nm = names[did]
                if nm.split("<")[0] in ONE_OFF:
                    one_off = True  # (lasts until the step's next marker: the scans and the bucket sort in between belong to it)
                elif one_off and any(sub in nm for sub, _ in MARKERS):
                    one_off = False
                if one_off:
                    stage_of[did] = "one_off_kept_buckets"
                    continue
                if "sssp_kernel<" in nm and "true" in [x.strip() for x in nm[nm.index("<") + 1:].rstrip(">").split(",")][-2:-1]:
                    cur = "post"
                for sub, st in MARKERS:
                    if sub in nm:
                        if st == "classify":
                            cur = st
                        elif cur != "post" and cur != "setup" and ORDER.index(st) >= ORDER.index(cur if cur in ORDER else "buckets"):
                            cur = st
                        elif cur != "setup":
                            cur = "post"
                        break
                stage_of[did] = cur
            # pass 2: a run of "buckets" belongs to what follows it
            i = 0
            while i < len(order):
                if stage_of[order[i]] != "buckets":
                    i += 1
                    continue
                j = i
                while j < len(order) and stage_of[order[j]] == "buckets":
                    j += 1
                label = "tigs_from_pairing" if any(any(k in names[x] for k in CUT_FIRST_ONLY) for x in order[i:j]) else "decomposition" if any(any(k in names[x] for k in DEVICE_ONLY) for x in order[i:j]) else (
                    "records" if any(any(k in names[x] for k in HOST_ONLY) for x in order[i:j]) else "buckets")
                for x in order[i:j]:
                    stage_of[x] = label
                i = j
            has_fetch = any("FETCH_SIZE" in rows[did] for did in order)
            if has_fetch:
                steps = max(steps, sum(1 for did in order if "classify_kernel" in names[did]))
            for did in order:
                st = stage_of[did]
                if st in ("setup", "post", "one_off_kept_buckets"):
                    continue
                if "__amd_rocclr_copyBuffer" in names[did]:  # the runtime's copy kernels (staged downloads on the side stream run beside
                    st = "copies"                             # any stage): their own bucket
                gather = names[did].split("<")[0] in GATHER
                for c in ("FETCH_SIZE", "WRITE_SIZE"):
                    if c in rows[did]:
                        per_stage[st][c] += rows[did][c] * 1024.0
                        if c == "FETCH_SIZE":
                            per_stage[st]["FETCH_CORRECTED"] += rows[did][c] * 1024.0 * (1.0 if gather else 2.0)
                if has_fetch:
                    n_dispatch[st] += 1
                    kernels[st].add(names[did].split("<")[0])
    steps = max(steps, 1)
    res = {}
    for st in per_stage:
        f, w = per_stage[st].get("FETCH_SIZE", 0.0) / steps, per_stage[st].get("WRITE_SIZE", 0.0) / steps
        # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of coalesced streaming reads (128-B requests
        # tallied at 64 B; confirmed here on classify_kernel / replay_state_init_kernel / build_ext_need_kernel, whose reads are
        # known: 0.50 of them) -> doubled for the streaming kernels; the gather kernels (GATHER above: dependent random accesses,
        # one 64-byte request each) keep the raw figure, as calibrated in round 1 (profiles/r01_final.md)
        fc = per_stage[st].get("FETCH_CORRECTED", 0.0) / steps
        res[st] = {"fetch_bytes_raw": int(f), "fetch_bytes": int(fc), "write_bytes": int(w), "traffic_bytes": int(fc + w),
                   "dispatches_per_step": round(n_dispatch[st] / steps, 1), "kernels": sorted(kernels[st])}
    try:
        allj = json.loads(open(out).read())
    except (OSError, ValueError):
        allj = {"comment": "HBM bytes per stage of one bench step from separate rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, KB -> bytes), "
                           "dispatches assigned to stages in launch order (tools/stage_traffic.py); steps = launches of classify_kernel"}
    allj[key] = {"steps_profiled": steps, "stages": res}
    open(out, "w").write(json.dumps(allj, indent=1) + "\n")
    print(key, json.dumps({k: v["traffic_bytes"] for k, v in res.items()}))


if __name__ == "__main__":
    main()
