#!/usr/bin/env python3
"""Summarise a tune_gemm --csv sweep: what the dispatcher picked against the best candidate per shape."""
import collections
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
by = collections.defaultdict(list)
for row in rows:
    dt, ta, tb, m, k, n, bs, kern, tile, depth, ms = row[:11]
    fused = int(row[11]) if len(row) > 11 else 0  # both K halves inside one workgroup
    by[(int(m), int(n), int(k))].append((float(ms), int(kern), int(tile), int(depth), fused))
names = {0: "auto", 1: "large", 2: "small", 3: "splitk", 4: "naive", 5: "huge", 7: "tiny", 8: "stream", 9: "medium"}
print("shape (m,n,k)          auto_us  best_us  best             auto/best")
worst = []
for key, c in by.items():
    auto = [x for x in c if x[1] == 0][0][0]
    best = min(x for x in c if x[1] != 0)
    worst.append(auto / best[0])
    print(f"{str(key):22s} {auto*1e3:8.1f} {best[0]*1e3:8.1f}  {names[best[1]]:7s} t{best[2]} d{best[3]:<4d}{'h' if best[4] else ' '} {auto/best[0]:6.2f}")
worst.sort()
print(f"auto/best: median {worst[len(worst)//2]:.2f}, 90th pct {worst[int(len(worst)*0.9)]:.2f}, max {worst[-1]:.2f}")
