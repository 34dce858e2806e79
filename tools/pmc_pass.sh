#!/bin/bash
# Runs the rocprofv3 counter passes for one bench configuration on the GPU box.
# Each pass is its own short run (counters are not split over passes by the tool).
# usage: tools/pmc_pass.sh <tag> <bench args...>     (writes gpurun_out/pmc_<tag>/)
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$tag
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$O" -o "$name" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras $ARGS > "$O/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$O/$name.log"; return 1; }
}
ARGS="$*"
pass fetch FETCH_SIZE && pass write WRITE_SIZE && \
pass sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES TCC_HIT_sum TCC_MISS_sum
ls "$O"
