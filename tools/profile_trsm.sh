#!/bin/bash
# Kernel-level profile of one 8192 x 8192 triangular solve (fp32 and fp64).
set -e
out=gpurun_out/trsm_prof
mkdir -p $out
export TMPDIR=/tmp
for dt in float double; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$dt -o run -- benchmark/bench_blas3 trsm \
    --csv-param=benchmark/config_csv/trsm_8192.csv --dtype=$dt --benchmark_min_time=0.05 > $out/$dt.log 2>&1
  f=$(find $out/$dt -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${dt}_kernel_stats.csv
done
echo done
