#!/bin/bash
# Counter passes over every tile family at one shape (the tuner launches each family under its own
# kernel name).  usage: tools/pmc_tiles.sh <float|double> [M K N]   -> gpurun_out/pmc_tiles_<dtype>/
set -u
dt=${1:-float}; M=${2:-4096}; K=${3:-4096}; N=${4:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_tiles_$dt
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$O" -o "$name" -- "$R/benchmark/tune_gemm" $M $K $N 1 3 --trans=nn --dtype=$dt > "$O/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$O/$name.log"; return 1; }
}
pass sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE && \
pass tcc TCC_HIT_sum TCC_MISS_sum && pass fetch FETCH_SIZE
ls "$O"
