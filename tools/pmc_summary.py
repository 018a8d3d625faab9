"""Aggregates rocprofv3 --pmc output (…_counter_collection.csv under the given directories) into
kernel,counter,mean_per_launch,launches rows -- the format of profiles/*_pmc.csv.

usage: python tools/pmc_summary.py OUT.csv DIR [DIR ...]
Counting instantiations of the SSSP kernels (template argument COUNT = true) are untimed instrumentation and skipped.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")  # (before the argument list is cut at its first parenthesis)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("mtg::", "").replace("(anonymous namespace)::", "")


def is_count_instantiation(name: str) -> bool:
    m = re.search(r"<(.*)>", name)
    return bool(m and "true" in [x.strip() for x in m.group(1).split(",")][-2:] and "sssp_kernel" in name)


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    launches = defaultdict(lambda: defaultdict(set))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row.get("Kernel_Name") or row.get("Kernel Name") or "")
                    if not k or is_count_instantiation(k):
                        continue
                    c = row.get("Counter_Name") or row.get("Counter Name")
                    v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                    disp = (path, row.get("Dispatch_Id") or row.get("Dispatch Id"))
                    acc[k][c] += v
                    launches[k][c].add(disp)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
        for k in sorted(acc):
            for c in sorted(acc[k]):
                n = max(len(launches[k][c]), 1)
                w.writerow([k, c, round(acc[k][c] / n, 1), n])
    print(f"wrote {out}")


if __name__ == "__main__":
    main()
