#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/*_counter_collection.csv):
per kernel and counter, the mean value per dispatch.  FETCH_SIZE is doubled to correct
for gfx950 counting 128-byte requests as 64 bytes on wide coalesced reads; FETCH_SIZE and
WRITE_SIZE are reported in KiB by the tool and converted to bytes here.
usage: tools/pmc_summary.py gpurun_out/pmc_f32nn [--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "gemm" not in k and "reduce" not in k and "trsm" not in k:
                continue
            acc[k.split("(")[0][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    print(k)
    out[k] = {}
    for c, vals in sorted(cs.items()):
        # one row per dispatch (and per dimension instance for some counters): average per dispatch
        mean = sum(vals) / len(vals)
        note = ""
        if c == "FETCH_SIZE":
            mean = mean * 1024 * 2
            note = " bytes (KiB x1024, x2 gfx950 correction)"
        elif c == "WRITE_SIZE":
            mean = mean * 1024
            note = " bytes"
        out[k][c] = mean
        print(f"   {c:32s} {mean:18.1f}{note}  (n={len(vals)})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        pass
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
