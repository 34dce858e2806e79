#!/usr/bin/env python3
"""Print VGPR / AGPR / occupancy per kernel of one .hip file (cross-compiled, no GPU needed).
usage: tools/kernel_resources.py sycl_blas_amd/csrc/gemm_f32.hip [name-filter]"""
import re
import subprocess
import sys

out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                      "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"sbamd::|void |\(sbamd::GemmArgs<\w+>\)", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|Occupancy \[waves/SIMD\]|TotalSGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        print(f"{name:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} "
              f"occ={r.get('Occupancy [waves/SIMD]')} spill={r.get('VGPRs Spill')}")
