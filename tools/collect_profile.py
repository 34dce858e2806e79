#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/ into the tracked profiles/<tag>/ set: the two bench lines,
the rocprofv3 kernel statistics of the same commands, the headline kernel's average over all launches and over the
timed ones (from the kernel trace), and the counter summaries (tools/pmc_summary.py).
usage: tools/collect_profile.py r02"""
import csv
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
names = {"f32": "sgemm8192", "f64": "dgemm8192"}
timed = {"_note": "rocprofv3 --kernel-trace of `python3 bench.py --dtype <dt> --no-cpu --no-extras` (10 warm-up + 20 timed "
                  "launches of the headline kernel); averages in ms over all 30 launches (what "
                  f"{tag}_{{s,d}}gemm8192_nn_kernel_stats.csv reports) and over the 20 timed ones (what bench.py's "
                  "roofline.achieved is derived from, there with HIP events)."}
for dt, name in names.items():
    shutil.copy(os.path.join(src, f"bench_{name}.json"), os.path.join(dst, f"{tag}_bench_{name}.json"))
    shutil.copy(os.path.join(src, f"trace_{dt}", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_{name}_nn_kernel_stats.csv"))
    line = json.loads(open(os.path.join(src, f"bench_{name}.json")).read().strip().splitlines()[-1])
    rows = [r for r in csv.DictReader(open(os.path.join(src, f"trace_{dt}", "trace_kernel_trace.csv")))
            if "gemm_mfma_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    steps = line["steps"]
    timed[f"{dt}_nn_8192"] = {"kernel": rows[0]["Kernel_Name"].split("(")[0].replace("void sbamd::", ""),
                              "launches": len(dur), f"avg_ms_all_{len(dur)}": round(sum(dur) / len(dur), 4),
                              f"avg_ms_timed_{steps}": round(sum(dur[-steps:]) / steps, 4),
                              "bench_event_ms": round(line["ms_per_step"], 4)}
shutil.copy(os.path.join(src, "trace_extras", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_all_kernels_stats.csv"))
json.dump(timed, open(os.path.join(dst, f"{tag}_kernel_trace_timed_steps.json"), "w"), indent=1)
traffic = {"_note": "HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes of `bench.py --steps 3 "
                    f"--warmup 1` (tools/pmc_pass.sh, summaries in {tag}_{{s,d}}gemm8192_nn_pmc.txt); FETCH_SIZE is reported in KiB and "
                    "counts 128-byte requests as 64 bytes on gfx950, so it is multiplied by 1024 and by 2; WRITE_SIZE by 1024. "
                    "Infinity-Cache hits are included in FETCH_SIZE."}
for dt, name in names.items():
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"),
                          os.path.join(root, "gpurun_out", f"pmc_{tag}_{dt}nn")], capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_{name}_nn_pmc.txt"), "w").write(out)
    found = {w[0]: float(w[1]) for w in (l.split() for l in out.splitlines()) if len(w) > 1 and w[0] in ("FETCH_SIZE", "WRITE_SIZE")}
    traffic[f"{dt}_nn_8192"] = int(found["FETCH_SIZE"] + found["WRITE_SIZE"])
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(timed, indent=1))
