#!/usr/bin/env python3
"""Fit the dispatcher's cost model to a tune_gemm --csv sweep and report the regret of its choices.

The model (gemm_dispatch.hpp, estimate_us): for a tile with `occ` resident workgroups per compute unit,
  us = k_chunk * (full * per_k[occ-1] + per_k[rem-1]) + (depth > 1 ? t_split + cube_bytes*2/bw + c_depth*depth : t_single)
Prints the fitted constants as C++ initialisers and, using the measured time of every candidate,
how far the plan the fitted model would pick is from the best measured candidate.
"""
import collections
import csv
import os
import math
import sys

import numpy as np
from scipy.optimize import least_squares

CUS = 256
# The workgroups of a last, partly filled round do not spread evenly over the units (they go where slots free up
# first): `rem` leftover workgroups per unit cost what TAIL * rem resident ones do.
TAIL = int(os.environ.get("TAIL", "1"))
TILES = {1: (64, 64, 4), 2: (128, 128, 2), 3: (256, 128, 1), 6: (96, 96, 2), 7: (192, 192, 1)}  # the fp32 "huge" tile is 256 wide, see features()
KERN_TILE = {2: 1, 1: 2, 5: 3}  # single-launch families -> tile (family 9: the 96 / 192 tiles by parameter bit 0)


def bk_of(tile, dtype):
    return {1: 32, 2: 32, 3: 16, 6: 32, 7: 32}[tile] if dtype == "float" else 16


def features(dtype, tile, depth, m, n, k):
    bm, bn, occ = TILES[tile]
    if tile == 3 and dtype == "float":
        bn = 256
    wgs = math.ceil(m / bm) * math.ceil(n / bn) * depth
    busiest = math.ceil(wgs / CUS)
    full, rem = divmod(busiest, occ)
    bk = bk_of(tile, dtype)
    k_chunk = bk * math.ceil(k / (bk * depth))
    return occ, full, rem, k_chunk


def unpack(x):
    return {"per_k": {1: x[0:4], 2: x[4:6], 3: x[6:7], 6: x[11:13], 7: x[13:14]}, "t_single": x[7], "t_split": x[8], "bw": x[9], "c_depth": x[10]}


def model(p, dtype, tile, depth, m, n, k):
    occ, full, rem, k_chunk = features(dtype, tile, depth, m, n, k)
    per_k = p["per_k"][tile]
    us = k_chunk * (full * per_k[occ - 1] + (per_k[(min(occ, TAIL * rem) if full else rem) - 1] if rem else 0.0))
    size = 4 if dtype == "float" else 8
    if depth > 1:
        us += p["t_split"] + 2.0 * m * n * depth * size / (p["bw"] * 1e6) + p["c_depth"] * depth
    else:
        us += p["t_single"]
    return us


def main(path):
    data = []
    fused_rows = []  # both K halves in one workgroup: fitted afterwards as a factor on the model of two resident workgroups
    auto = {}
    for row in csv.reader(open(path)):
        dt, ta, tb, m, k, n, bs, kern, tile, depth, ms = row[:11]
        fused = len(row) > 11 and row[11] == "1"
        m, k, n, kern, tile, depth = int(m), int(k), int(n), int(kern), int(tile), int(depth)
        us = float(ms) * 1e3
        if fused:
            if tile in TILES:  # (the four-quarters form on the 32 x 32 tile has constants of its own)
                fused_rows.append((dt, tile, m, n, k, us))
            continue
        if kern == 0:
            auto[(m, n, k)] = us
            continue
        if kern in KERN_TILE:
            tile, depth = KERN_TILE[kern], 1
        elif kern == 9:
            tile, depth = (7 if depth & 1 else 6), 1
        elif kern != 3:
            continue
        data.append((dt, tile, depth, m, n, k, us))
    dtype = data[0][0]
    x0 = np.array([0.035, 0.041, 0.057, 0.074, 0.078, 0.146, 0.12, 1.5, 5.0, 3.0, 0.03, 0.04, 0.075, 0.135])
    if dtype == "double":
        x0[:7] *= 1.8
        x0[11:] *= 1.8

    def resid(x):
        p = unpack(x)
        return [math.log(model(p, *row[:6]) / row[6]) for row in data]

    fit = least_squares(resid, x0, bounds=(1e-4, np.inf), loss="soft_l1", f_scale=0.2)
    p = unpack(fit.x)
    err = np.abs(np.array(resid(fit.x)))
    print(f"{len(data)} candidates, median |log error| {np.median(err):.3f}, 90th pct {np.percentile(err, 90):.3f}")
    print("small_k = {" + ", ".join(f"{v:.4f}" for v in p["per_k"][1]) + "}")
    print("large_k = {" + ", ".join(f"{v:.4f}" for v in p["per_k"][2]) + "}")
    print("huge_k  = {" + ", ".join(f"{v:.4f}" for v in p["per_k"][3]) + "}")
    print("medium_k = {" + ", ".join(f"{v:.4f}" for v in p["per_k"][6]) + "}")
    print("wide_k  = {" + ", ".join(f"{v:.4f}" for v in p["per_k"][7]) + "}")
    print(f"t_single = {p['t_single']:.2f} us, t_split = {p['t_split']:.2f} us, bw = {p['bw']:.2f} TB/s, "
          f"c_depth = {p['c_depth']:.4f} us")
    by = collections.defaultdict(list)
    for row in data:
        by[(row[3], row[4], row[5])].append(row)
    for margin in (1.0, 0.9, 0.85, 0.8):
        regrets = []
        for shape, cands in by.items():
            best = min(c[6] for c in cands)
            single = min((model(p, *c[:6]), c[6]) for c in cands if c[2] == 1)
            split = min(((model(p, *c[:6]), c[6]) for c in cands if c[2] > 1), default=(1e300, 0))
            pick = split if split[0] < margin * single[0] else single
            regrets.append(pick[1] / best)
        regrets.sort()
        print(f"margin {margin}: regret median {regrets[len(regrets)//2]:.3f}, 90th pct "
              f"{regrets[int(len(regrets)*0.9)]:.3f}, max {regrets[-1]:.3f}")
    # halves_factor of gemm_dispatch.hpp (estimate_us with fused = true): measured time of the fused launch
    # over the model's time for the two sets as two resident workgroups per tile, where the planner may use it
    for tile in (1, 2, 6):
        ratios = []
        for dt, t, m, n, k, us in fused_rows:
            bm, bn, occ = TILES[t]
            tiles = math.ceil(m / bm) * math.ceil(n / bn)
            if t != tile or k < 512:
                continue
            busiest = math.ceil(tiles / CUS) * 2
            full, rem = divmod(busiest, occ)
            bk = bk_of(t, dt)
            k_chunk = bk * math.ceil(k / (bk * 2))
            per_k = p["per_k"][t]
            base = k_chunk * (full * per_k[occ - 1] + (per_k[(min(occ, TAIL * rem) if full else rem) - 1] if rem else 0.0))
            ratios.append((us - p["t_single"]) / base)
        if ratios:
            ratios.sort()
            print(f"halves_factor tile {tile}: median {ratios[len(ratios)//2]:.3f} (min {ratios[0]:.3f}, max {ratios[-1]:.3f}, n={len(ratios)})")
    now = sorted(auto[s] / min(c[6] for c in by[s]) for s in by if s in auto)
    print(f"library as measured: regret median {now[len(now)//2]:.3f}, 90th pct {now[int(len(now)*0.9)]:.3f}, max {now[-1]:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
