#!/usr/bin/env python3
"""How far the planner's choice is from the best measured candidate, per shape of a tune_gemm --csv sweep.

Every candidate line carries what the library reports it launched (tile rows, tile columns, K slabs, fused halves).
The planner's choice is looked up AMONG the candidates by that identity, so its time and the best time come from the
same table (the `auto` line's own timing differs by a few percent with its place in the run).  The choice is either
the sweep's `auto` line (the library as it was when the sweep ran) or, with --probe, what tools/bin/plan_probe says
now (the planner of the sources as they are: constants can be changed and judged without another sweep).
usage: tools/plan_regret.py gpurun_out/tune_sweep_float.csv [--probe] [--cus N] [--list 1.05]"""
import collections
import csv
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bk_of(tile_m, dtype):
    if dtype == "double":
        return 16
    return 16 if tile_m == 256 else 32


def main():
    path = sys.argv[1]
    probe = "--probe" in sys.argv
    cus = int(sys.argv[sys.argv.index("--cus") + 1]) if "--cus" in sys.argv else 256
    show = float(sys.argv[sys.argv.index("--list") + 1]) if "--list" in sys.argv else 1.05
    cands = collections.defaultdict(dict)   # shape -> identity -> best ms
    auto = {}                               # shape -> (identity, ms)
    dtype = None
    for row in csv.reader(open(path)):
        dtype = row[0]
        m, k, n, kern = int(row[3]), int(row[4]), int(row[5]), int(row[7])
        ms = float(row[10])
        ident = (int(row[13]), int(row[14]), int(row[15]), int(row[16]))
        shape = (m, n, k)
        if kern == 0:
            auto[shape] = (ident, ms)
        elif kern != 4:  # the one-element-per-lane kernel is a checker, not a candidate
            cands[shape][ident] = min(ms, cands[shape].get(ident, 1e300))
    picks = {}
    if probe:
        shapes = sorted(cands)
        out = subprocess.run([os.path.join(ROOT, "tools", "bin", "plan_probe"), dtype, str(cus)],
                             input="".join(f"{m} {n} {k}\n" for m, n, k in shapes), capture_output=True, text=True, check=True).stdout
        for line in out.splitlines():
            m, n, k, tm, tn, depth, fused = (int(x) for x in line.split())
            if depth > 1 and not fused:  # whole K steps per slab, as the launch code cuts them
                bk = bk_of(tm, dtype)
                chunk = max(bk, bk * math.ceil(k / (bk * depth)))
                depth = max(1, math.ceil(k / chunk))
            picks[(m, n, k)] = (tm, tn, depth, fused)
    else:
        picks = {s: auto[s][0] for s in auto}
    rows = []
    missing = 0
    for shape, table in cands.items():
        if shape not in picks:
            continue
        best_id = min(table, key=table.get)
        pick = picks[shape]
        if pick in table:
            t = table[pick]
        elif not probe:
            t, missing = auto[shape][1], missing + 1
        else:
            missing += 1
            continue
        rows.append((t / table[best_id], shape, pick, t, best_id, table[best_id]))
    rows.sort()
    r = [x[0] for x in rows]
    print(f"{os.path.basename(path)}: {len(rows)} shapes, planner = {'sources as they are (plan_probe)' if probe else 'library at sweep time (auto line)'}"
          f"{', ' + str(cus) + ' compute units' if probe else ''}; {missing} choices not among the candidates")
    print(f"  time of the chosen candidate / best candidate: median {r[len(r) // 2]:.3f}, 90th pct {r[int(len(r) * 0.9)]:.3f}, max {r[-1]:.3f}")
    for floor in (20.0, 50.0):
        big = [x[0] for x in rows if x[5] * 1e3 >= floor]
        if big:
            print(f"  shapes whose best candidate takes at least {floor:.0f} us ({len(big)}): median {big[len(big) // 2]:.3f}, 90th pct {big[int(len(big) * 0.9)]:.3f}, max {big[-1]:.3f}")
    lost = sorted((x[3] - x[5]) * 1e3 for x in rows)
    print(f"  microseconds lost to the best candidate: median {lost[len(lost) // 2]:.2f}, 90th pct {lost[int(len(lost) * 0.9)]:.2f}, max {lost[-1]:.1f}")
    for ratio, shape, pick, t, best_id, tb in rows:
        if ratio >= show:
            print(f"    {shape[0]:5d} x {shape[1]:5d} x {shape[2]:5d}: chose {pick[0]}x{pick[1]} depth {pick[2]}{' fused' if pick[3] else ''} {t * 1e3:8.1f} us, "
                  f"best {best_id[0]}x{best_id[1]} depth {best_id[2]}{' fused' if best_id[3] else ''} {tb * 1e3:8.1f} us  ({ratio:.3f})")


if __name__ == "__main__":
    main()
