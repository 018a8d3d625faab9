import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_parity as T
from matchtigs_amd import synth
import oracle_lib
bg = synth.g_csr(2000, seed=4, k=15, mean_out_degree=5.0, mean_weight=4.0, max_degree=9)
G, dev, S, start, count, pool = T._gpu_candidates(bg, 0)
o_on, off, keys, st = T._oracle(oracle_lib, bg).candidate_lists(bg.k)
print("levels", dev.last_sssp_levels())
bad = 0
for i in range(S):
    g = pool[int(start[i]):int(start[i]) + int(count[i])]
    e = keys[int(off[i]):int(off[i + 1])]
    if not np.array_equal(g, e):
        bad += 1
        if bad <= 6:
            print("src", i, "count", count[i], "exp count", len(e), "start", start[i], "start%2048", int(start[i]) % 2048)
            print("   got", [hex(int(x)) for x in g[:12]])
            print("   exp", [hex(int(x)) for x in e[:12]])
print("bad sources", bad, "of", S, "count mismatches", int((count.astype(np.int64) != np.diff(off).astype(np.int64)).sum()))
for i in (1238, 1630):
    s0 = int(start[i]); e = keys[int(off[i]):int(off[i + 1])]
    print("src", i, "pool around start:", [hex(int(x)) for x in pool[s0:s0 + 14]])
    w = np.nonzero(pool == e[0])[0]
    print("   first expected key found at pool positions", w[:10], " (start", s0, ")")
    for p in w[:4]:
        print("      pool[p:p+12] =", [hex(int(x)) for x in pool[int(p):int(p) + 12]])
# who else starts near
order = np.argsort(start)
for i in (1238, 1630):
    s0 = int(start[i])
    near = [(int(j), int(start[j]), int(count[j])) for j in order if count[j] > 0 and s0 - 40 <= int(start[j]) <= s0 + 40]
    print("lists near", s0, near)
