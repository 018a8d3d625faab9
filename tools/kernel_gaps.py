"""GPU idle gaps inside the last bench step of a rocprofv3 --kernel-trace CSV: intervals longer than MIN us in which no kernel runs,
between the last classify_kernel and the last kernel of the trace, with the kernels on either side.
usage: python tools/kernel_gaps.py KERNEL_TRACE.csv [MIN_US=100]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "classify_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
short = lambda n: n.replace("mtg::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
busy_end = int(rows[idx]["End_Timestamp"])
prev = rows[idx]
total_gap = 0.0
for r in rows[idx + 1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gap = (s - busy_end) / 1e3
        if gap >= min_us:
            print(f"{(busy_end - t0) / 1e3:9.1f} us: idle {gap:8.1f} us   after {short(prev['Kernel_Name'])}  before {short(r['Kernel_Name'])}")
        total_gap += gap
    if e > busy_end:
        busy_end, prev = e, r
print(f"step span {(busy_end - t0) / 1e3:.1f} us, idle {total_gap:.1f} us")
