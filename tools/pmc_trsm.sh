#!/bin/bash
# Counter pass over one 8192 x 8192 STRSM (benchmark/bench_blas3): MFMA-busy cycles, LDS conflicts.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_trsm
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$O" -o sq -- "$R/benchmark/bench_blas3" trsm --csv-param="$R/benchmark/config_csv/trsm_8192.csv" --dtype=float --benchmark_min_time=0.02 > "$O/sq.log" 2>&1 || { echo "pass failed"; tail -5 "$O/sq.log"; exit 1; }
ls "$O"
