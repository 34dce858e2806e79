#!/usr/bin/env python3
"""Table from tools/pmc_families.sh output (gpurun_out/pmc_families/<dtype>_<label>/): per case, every library kernel
the 12 calls launched, with per-launch averages.  MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8 XCDs); FETCH_SIZE x 1024 x 2 (KiB, and gfx950 counts a 128-byte request as 64 bytes).
Under counter collection GRBM_GUI_ACTIVE holds some 35 kcycles (15 us) of dispatch per launch, so the MFMA-busy share of
launches shorter than about 300 kcycles is understated (SGEMM 1024^3 runs 18.9 us = 45 kcycles without counters).
usage: tools/pmc_families_summary.py gpurun_out/pmc_families > profiles/r03/r03_tiles_pmc.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
print(__doc__.split("usage:")[0].strip())
print()
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    per = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    grid = {}
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "sbamd::" not in k or "fill_uniform" in k:
                    continue
                name = re.sub(r"\(.*", "", k.replace("void sbamd::", "")).replace("sbamd::", "")
                per[name][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[name][row["Counter_Name"]] += 1
                grid[name] = (int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])), int(row["Workgroup_Size"]),
                              row.get("LDS_Block_Size", "?"), row.get("VGPR_Count", "?"), row.get("Accum_VGPR_Count", "?"), row.get("Scratch_Size", "?"))
    log = open(os.path.join(d, "sq.log")).read().strip().splitlines()
    print(f"== {os.path.basename(d)}   ({log[-1] if log else ''})")
    for name in sorted(per, key=lambda n: -per[n].get("GRBM_GUI_ACTIVE", 0.0)):
        c = {k: per[name][k] / cnt[name][k] for k in per[name]}
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cycles) if cycles else float("nan")
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        ratio = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else float("nan")
        hm = c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)
        hit = c.get("TCC_HIT_sum", 0.0) / hm if hm else float("nan")
        fetch = c.get("FETCH_SIZE", 0.0) * 2048.0 / 1e9
        g = grid[name]
        print(f"   {name}\n      grid {g[0]} x {g[1]} lanes, VGPR {g[3]}+{g[4]}, scratch {g[5]} B, {cnt[name].get('GRBM_GUI_ACTIVE', 0)} launches: "
              f"mfma_busy={busy * 100:5.1f}%  lds_conflict/active={lds * 100:5.1f}%  valu/mfma={ratio:6.2f}  L2 hit={hit * 100:5.1f}%  "
              f"fetch={fetch:7.3f} GB  ({cycles / 1e3:9.1f} kcycles per launch)")
