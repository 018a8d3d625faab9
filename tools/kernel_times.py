"""Per-kernel durations (us) of the SSSP stage from a rocprofv3 --kernel-trace CSV. usage: python tools/kernel_times.py TRACE.csv"""
import collections, csv, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sssp" in n or "sort_" in n or "fix_" in n:
        d[n.replace("void mtg::", "")[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{k:46s}", [round(x) for x in v])
