#!/usr/bin/python3
"""Stand-in for the external blossom5 matcher in tests: `tiny_matcher.py -e <instance> -w <solution>`.

Reads a minimum-cost perfect matching instance in the format optimal matchtigs writes (first line "<nodes> <edges>", then
"<n1> <n2> <weight>" per edge) and writes "<nodes> <nodes/2>" followed by one "<n1> <n2>" line per matched edge.
Up to 22 nodes it solves the instance exactly (subset dynamic programme); beyond that it builds a valid, not necessarily
optimal perfect matching from the instance's known shape (two copies of the matching graph joined by copy edges, four extra
nodes per component) -- enough to exercise everything after the matcher. Exit code 1 if no perfect matching is found.
"""
import sys


def exact(n, edges):
    INF = float("inf")
    w = {}
    for a, b, c in edges:
        if a != b and c < w.get((min(a, b), max(a, b)), INF):
            w[(min(a, b), max(a, b))] = c
    adj = [[] for _ in range(n)]
    for (a, b), c in sorted(w.items()):
        adj[a].append((b, c))
    full = (1 << n) - 1
    best = {0: (0, None)}
    order = [0]
    for mask in order:  # breadth first by matched-node count: a mask's cost is final before it is expanded
        cost = best[mask][0]
        i = 0
        while mask >> i & 1:
            i += 1
        if i >= n:
            continue
        for j, c in adj[i]:
            if mask >> j & 1:
                continue
            nm = mask | 1 << i | 1 << j
            if nm not in best:
                best[nm] = (cost + c, (mask, i, j))
                order.append(nm)
            elif cost + c < best[nm][0]:
                best[nm] = (cost + c, (mask, i, j))
    if full not in best:
        return None
    out, mask = [], full
    while mask:
        _, (pm, i, j) = best[mask]
        out.append((i, j))
        mask = pm
    return sorted(out)


def structured(n, edges):
    zero_second = [b for a, b, c in edges if c == 0]
    if not zero_second:
        return [] if n == 0 else None
    T = min(zero_second) // 2
    members = {}  # first extra node of a component -> first-copy nodes attached to it
    for a, b, c in edges:
        if c == 0 and a < T and (b - 2 * T) % 4 == 0:
            members.setdefault(b, []).append(a)
    matched = [False] * n
    out = []
    for e0 in sorted(members):
        ms = members[e0]
        if len(ms) < 2:
            return None
        a, b = ms[0], ms[1]
        out += [(a, e0), (b, e0 + 1), (a + T, e0 + 2), (b + T, e0 + 3)]
        for x in (a, b, a + T, b + T, e0, e0 + 1, e0 + 2, e0 + 3):
            matched[x] = True
    for a, b, c in edges:  # first-copy edges in file order, mirrored into the second copy
        if c != 0 and a < T and b < T and not matched[a] and not matched[b]:
            out += [(a, b), (a + T, b + T)]
            matched[a] = matched[b] = matched[a + T] = matched[b + T] = True
    for x in range(T):
        if not matched[x]:
            out.append((x, x + T))
            matched[x] = matched[x + T] = True
    return out if all(matched) else None


def main(argv):
    src = argv[argv.index("-e") + 1]
    dst = argv[argv.index("-w") + 1]
    with open(src) as f:
        n, m = (int(x) for x in f.readline().split())
        edges = [tuple(int(x) for x in line.split()) for line in f if line.strip()]
    sol = exact(n, edges) if n <= 22 else structured(n, edges)
    if sol is None:
        sys.stderr.write("tiny_matcher: no perfect matching found\n")
        return 1
    with open(dst, "w") as f:
        f.write(f"{n} {n // 2}\n")
        for a, b in sol:
            f.write(f"{a} {b}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
