#!/usr/bin/env python3
"""Table from tools/pmc_tiles.sh output (gpurun_out/pmc_tiles_<dtype>/): per tile kernel, the counters of its
single-launch dispatches (the smallest grid of that kernel name; bigger grids are the tuner's K-slab launches).
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); FETCH_SIZE x 1024 x 2 (gfx950).
usage: tools/pmc_tiles_summary.py gpurun_out/pmc_tiles_float gpurun_out/pmc_tiles_double"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> grid -> counter -> [sum over dispatches]
    count = defaultdict(lambda: defaultdict(lambda: defaultdict(int)))
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "gemm_mfma_kernel" not in k:
                    continue
                name = re.sub(r"\(.*", "", k.replace("void sbamd::", ""))
                g = int(row["Grid_Size"])
                per[name][g][row["Counter_Name"]] += float(row["Counter_Value"])
                count[name][g][row["Counter_Name"]] += 1
    for name in sorted(per):
        g = min(per[name])
        c = {k: per[name][g][k] / count[name][g][k] for k in per[name][g]}
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cycles) if cycles else float("nan")
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        ratio = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else float("nan")
        hit = c.get("TCC_HIT_sum", 0.0) / (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)) if c.get("TCC_HIT_sum") else float("nan")
        fetch = c.get("FETCH_SIZE", 0.0) * 2048.0 / 1e9
        print(f"{name:62s} mfma_busy={busy * 100:5.1f}%  lds_conflict/active={lds * 100:5.1f}%  valu/mfma insts={ratio:5.2f}  "
              f"L2 hit={hit * 100:5.1f}%  fetch={fetch:6.2f} GB  ({cycles / 1e6:6.2f} Mcycles)")
