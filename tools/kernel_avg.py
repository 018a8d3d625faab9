"""Average duration per kernel from a rocprofv3 --stats directory: python tools/kernel_avg.py DIR PATTERN [PATTERN ...]"""
import csv, glob, os, sys
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if any(p in r["Name"] for p in sys.argv[2:]):
            print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
