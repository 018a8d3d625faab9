"""Adds the HBM traffic of the SSSP stage (FETCH_SIZE + WRITE_SIZE of its kernels, KB -> bytes, per launch) from a
tools/pmc_summary.py CSV to profiles/traffic.json under bench.py's workload key.
usage: python tools/make_traffic_json.py PMC_SUMMARY.csv KEY BENCH_LINE.json [SOURCE-NOTE]
BENCH_LINE.json = the JSON line of the profiled bench command: its roofline.kernels names are stored as `levels`, and bench.py
only reports the traffic while the same level kernels run (a changed kernel makes the entry stale instead of silently wrong)."""
import csv, json, sys
from pathlib import Path

pmc, key, bench_line = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else pmc
levels = [kk["kernel"] for kk in json.loads([l for l in open(bench_line) if l.startswith("{")][-1])["roofline"]["kernels"]]
fetch = write = 0.0
kernels = []
for row in csv.DictReader(open(pmc)):
    k = row["kernel"]
    if not k.startswith(("sssp_enum_kernel", "sort_lists_kernel", "fix_compact_kernel", "sssp_kernel", "active_range_kernel")):
        continue
    if row["counter"] == "FETCH_SIZE":
        fetch += float(row["mean_per_launch"]) * 1024
        kernels.append(k)
    elif row["counter"] == "WRITE_SIZE":
        write += float(row["mean_per_launch"]) * 1024
p = Path(__file__).resolve().parent.parent / "profiles" / "traffic.json"
d = json.loads(p.read_text())
d[key] = {"traffic_bytes": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write), "levels": levels,
          "source": f"{note} ({' + '.join(kernels)}; separate rocprofv3 --pmc passes of the bench command, tools/profile_round.sh)"}
p.write_text(json.dumps(d, indent=1) + "\n")
print(key, d[key])
