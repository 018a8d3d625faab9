fmt='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(sys.argv[1], d["workload"], d["stage_ms_best"], [round(x["ms"],3) for x in d["levels"]], [x["sources"] for x in d["levels"]])'
for i in 1 2; do
timeout -k 10 200 python tools/sssp_probe.py --log2-edges 24 27 --reps 7 2>/dev/null | python -c "$fmt" A
timeout -k 10 200 python tools/sssp_probe.py --log2-edges 24 27 --reps 7 --lib matchtigs_amd/libmatchtigs_B.so 2>/dev/null | python -c "$fmt" B
done
