#!/bin/bash
# Counter passes for the 16-bit direct-to-LDS ring kernel (bf16 -> f32, 8192^3, uniform [-1, 1) operands),
# one layout per run of tools/bin/bf16_lab, each pass its own run.
# usage: tools/pmc_bf16.sh [layouts="nn nt tn tt"] [n=8192]     -> gpurun_out/pmc_bf16/<layout>_<pass>/
set -u
layouts=${1:-"nn nt tn tt"}; n=${2:-8192}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bf16
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
pass() {
  tr=$1; name=$2; shift 2
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$O/${tr}_$name" -o out -- "$R/tools/bin/bf16_lab" $n 3 $tr > "$O/${tr}_$name.log" 2>&1 || { echo "pass $tr $name failed"; tail -5 "$O/${tr}_$name.log"; return 1; }
}
for tr in $layouts; do
  pass $tr sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
  pass $tr sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE || exit 1
  pass $tr fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
  echo "layout $tr done"
done
ls "$O"
