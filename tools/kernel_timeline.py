"""Timeline of the kernels of the last finish in a rocprofv3 --kernel-trace CSV: start (us after the last need_emit_kernel), duration, name.
usage: python tools/kernel_timeline.py KERNEL_TRACE.csv [N=60]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "need_emit_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + n]:
    name = r["Kernel_Name"].replace("mtg::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {name}")
