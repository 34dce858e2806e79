#!/usr/bin/env python3
"""Summarise tools/pmc_bf16.sh output: per layout, the ring kernel's mean counters per dispatch,
the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-busy share, LDS conflict share,
L2 hit rate and fetched bytes (FETCH_SIZE in KiB, doubled for gfx950's 128-byte requests).
usage: tools/pmc_bf16_summary.py gpurun_out/pmc_bf16 [kernel substring=ring_kernel] [n=8192]"""
import csv
import os
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "ring_kernel"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
SIMDS = 1024
flops = 2.0 * n ** 3
print(f"bf16 -> f32 {n}^3, kernel *{needle}*, uniform [-1, 1) operands (tools/bf16_lab), one MI355X; means over the dispatches of each pass")
print("layout   ms(sq1)  clock GHz  MFMA-busy  TFLOP/s  at 2.4 GHz  LDS conflict  wait-any  wait-inst  active   L2 hit  fetched MB (alg MB)  VALU:MFMA  LDS:MFMA  coexec")
for tr in ("nn", "nt", "tn", "tt"):
    vals = defaultdict(list)
    dur = defaultdict(list)
    for name in ("sq1", "sq2", "fetch"):
        path = os.path.join(root, f"{tr}_{name}", "out_counter_collection.csv")
        if not os.path.exists(path):
            continue
        seen = set()
        with open(path) as f:
            for row in csv.DictReader(f):
                if needle not in row["Kernel_Name"]:
                    continue
                vals[(name, row["Counter_Name"])].append((row["Dispatch_Id"], float(row["Counter_Value"])))
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    if not dur:
        continue

    def mean(name, counter):
        per = defaultdict(float)
        for d, v in vals.get((name, counter), []):
            per[d] += v  # some counters come as one row per instance
        return sum(per.values()) / len(per) if per else float("nan")

    ms = sum(dur["sq1"]) / len(dur["sq1"])
    gui = mean("sq1", "GRBM_GUI_ACTIVE")
    clock = gui / 8 / (ms * 1e-3) / 1e9
    busy = mean("sq1", "SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * gui / 8)
    tf = flops / (ms * 1e-3) / 1e12
    conflict = mean("sq1", "SQ_LDS_BANK_CONFLICT") / mean("sq1", "SQ_LDS_IDX_ACTIVE")
    wc = mean("sq1", "SQ_WAVE_CYCLES")
    hit = mean("sq2", "TCC_HIT_sum")
    miss = mean("sq2", "TCC_MISS_sum")
    fetched = mean("fetch", "FETCH_SIZE") * 1024 * 2 / 1e6
    alg = 2 * n * n * 2 / 1e6
    mf = mean("sq2", "SQ_INSTS_MFMA")
    print(f"{tr.upper():6s} {ms:8.3f} {clock:10.3f} {busy:10.3f} {tf:8.1f} {tf * 2.4 / clock:11.1f} {conflict:13.3f} "
          f"{mean('sq1', 'SQ_WAIT_ANY') / wc:9.3f} {mean('sq1', 'SQ_WAIT_INST_ANY') / wc:10.3f} {mean('sq1', 'SQ_ACTIVE_INST_ANY') / wc:7.3f} "
          f"{hit / (hit + miss):8.3f} {fetched:11.0f} ({alg:.0f}) {mean('sq2', 'SQ_INSTS_VALU') / mf:10.2f} {mean('sq2', 'SQ_INSTS_LDS') / mf:9.2f} "
          f"{mean('sq2', 'SQ_VALU_MFMA_COEXEC_CYCLES') / mean('sq1', 'SQ_VALU_MFMA_BUSY_CYCLES'):7.3f}")
