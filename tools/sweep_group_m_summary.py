#!/usr/bin/env python3
"""Table from tools/sweep_group_m.sh: per dtype and band height, TFLOP/s, fetched bytes per launch
(FETCH_SIZE KiB x 1024 x 2, the gfx950 correction) and L2 hit rate of the headline kernel."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
print("dtype  group_m   ms/launch  TFLOP/s   fetched GB/launch   L2 hit")
for dt in ("f32", "f64"):
    for gm in (2, 4, 8, 16):
        try:
            line = [l for l in open(os.path.join(root, f"{dt}_gm{gm}.json")) if l.startswith("{")][-1]
        except (OSError, IndexError):
            continue
        j = json.loads(line)

        def mean(sub, counter):
            vals = {}
            for path in glob.glob(os.path.join(root, f"{dt}_gm{gm}_{sub}", "*counter_collection.csv")):
                for row in csv.DictReader(open(path)):
                    if "gemm_mfma_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals[row["Dispatch_Id"]] = vals.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            return sum(vals.values()) / len(vals) if vals else float("nan")

        hit, miss = mean("tcc", "TCC_HIT_sum"), mean("tcc", "TCC_MISS_sum")
        print(f"{dt:5s} {gm:8d} {j['ms_per_step']:11.4f} {j['value']:8.2f} {mean('fetch', 'FETCH_SIZE') * 2048 / 1e9:19.2f} {hit / (hit + miss):8.3f}")
