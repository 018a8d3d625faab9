"""Kernels around the last classify_kernel of a rocprofv3 --kernel-trace CSV: B before, A after (start relative to it, duration, name).
usage: python tools/kernel_window.py KERNEL_TRACE.csv [B=6] [A=12]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
A = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "classify_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[max(0, idx - B):idx + A]:
    name = r["Kernel_Name"].replace("mtg::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:12.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f}  {name}")
